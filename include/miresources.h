/* miresources.h -- C ABI of the asset-decode helpers on the CALLER side of the hot path.
 *
 * Host-only (no GPU, no HIP): bytes of a PNG / JPEG (sequential or progressive) file -> RGBA8 rows ready for mirhi_image_upload
 * (include/mirhi.h).  Replaces what the reference obtains from its `image` / `gltf` dependencies:
 *   - `gltf::import(path)` returns `Vec<gltf::image::Data>` (pixels, format, width, height), which
 *     crates/resources/src/model.rs:120 binds to `_images` and drops;
 *   - assets/textures/<name>/<name>_{Color,NormalGL,Roughness,AmbientOcclusion,Metalness}.jpg and <name>.png are the
 *     files a texture loader would open (crates/rhi/src/texture.rs:1-5 is a stub).
 * Implementation: renderer-rs_amd/host/image_decode.hpp (header-only C++), wrapped by host/resources_capi.cpp into
 * renderer-rs_amd/libmiresources.so.  Thread-safe; the error string is thread-local. */
#ifndef MIRESOURCES_H
#define MIRESOURCES_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    uint32_t width, height;
    uint32_t source_channels;   /* 1 grey, 2 grey+alpha, 3 rgb, 4 rgba -- what the file stored before expansion */
    uint32_t reserved;
    uint8_t* rgba;              /* width*height*4 bytes, top row first; owned by the library until mires_image_free */
} mires_image;

#define MIRES_OK 0
#define MIRES_ERR_DECODE 1      /* malformed or unsupported file; see mires_last_error_message */
#define MIRES_ERR_IO 2          /* file not found / unreadable */
#define MIRES_ERR_ARGUMENT 3    /* null pointer */

/* format is sniffed from the signature (PNG, JPEG) like image::load_from_memory */
int32_t mires_image_decode(const uint8_t* bytes, uint64_t len, mires_image* out);
int32_t mires_image_load(const char* path, mires_image* out);
void mires_image_free(mires_image* img);            /* null-safe; zeroes the struct */
const char* mires_last_error_message(void);

#ifdef __cplusplus
}
#endif
#endif
