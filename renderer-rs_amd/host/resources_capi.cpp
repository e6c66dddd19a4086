// resources_capi.cpp -- C ABI (include/miresources.h) over image_decode.hpp.  Built with g++ into libmiresources.so.
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/miresources.h"
#include "image_decode.hpp"

namespace {
thread_local std::string g_error;
int32_t fail(int32_t code, const std::string& msg) { g_error = msg; return code; }
int32_t hand_over(mirhi::resources::ImageData&& img, mires_image* out) {
    out->rgba = (uint8_t*)malloc(img.rgba.size() ? img.rgba.size() : 1);
    if (!out->rgba) return fail(MIRES_ERR_DECODE, "out of host memory");
    memcpy(out->rgba, img.rgba.data(), img.rgba.size());
    out->width = img.width; out->height = img.height; out->source_channels = img.source_channels; out->reserved = 0;
    return MIRES_OK;
}
}  // namespace

extern "C" int32_t mires_image_decode(const uint8_t* bytes, uint64_t len, mires_image* out) {
    if (!bytes || !out) return fail(MIRES_ERR_ARGUMENT, "null argument");
    memset(out, 0, sizeof *out);
    try { return hand_over(mirhi::resources::decode_image(bytes, (size_t)len), out); }
    catch (const std::exception& e) { return fail(MIRES_ERR_DECODE, e.what()); }
}
extern "C" int32_t mires_image_load(const char* path, mires_image* out) {
    if (!path || !out) return fail(MIRES_ERR_ARGUMENT, "null argument");
    memset(out, 0, sizeof *out);
    try { return hand_over(mirhi::resources::load_image(path), out); }
    catch (const mirhi::resources::ImageError& e) {
        const std::string m = e.what();
        return fail(m.rfind("File not found", 0) == 0 ? MIRES_ERR_IO : MIRES_ERR_DECODE, m);
    }
    catch (const std::exception& e) { return fail(MIRES_ERR_DECODE, e.what()); }
}
extern "C" void mires_image_free(mires_image* img) {
    if (!img) return;
    free(img->rgba);
    memset(img, 0, sizeof *img);
}
extern "C" const char* mires_last_error_message(void) { return g_error.c_str(); }
