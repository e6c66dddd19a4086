// Mutation fuzzer for renderer-rs_amd/host/image_decode.hpp, meant for a sanitizer build on the CPU:
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -o /tmp/fuzz_image tools/fuzz_image_decode.cpp
//   /tmp/fuzz_image <iterations> file.png file.jpg ...
// Add -DMIRHI_IMAGE_FUZZ_SKIP_CHECKS to disable the PNG CRC / Adler-32 checks so mutants reach inflate and the unfilter.
// Every mutant must end in an image or an ImageError; the sanitizers catch anything else.
#include <cstdio>
#include <cstdlib>
#include "../renderer-rs_amd/host/image_decode.hpp"

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s iterations files...\n", argv[0]); return 2; }
    const long iters = atol(argv[1]);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (int a = 2; a < argc; a++) {
        std::ifstream f(argv[a], std::ios::binary);
        std::vector<uint8_t> base((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        if (base.empty()) { fprintf(stderr, "cannot read %s\n", argv[a]); return 2; }
        long ok = 0, refused = 0;
        for (long i = 0; i < iters; i++) {
            std::vector<uint8_t> m = base;
            const int kind = (int)(rnd() % 4);
            if (kind == 0) m.resize(rnd() % m.size());                                        // truncate
            else if (kind == 1) for (int k = 0; k < 1 + (int)(rnd() % 8); k++) m[rnd() % m.size()] ^= (uint8_t)(1u << (rnd() % 8));
            else if (kind == 2) for (int k = 0; k < 1 + (int)(rnd() % 4); k++) m[rnd() % m.size()] = (uint8_t)rnd();
            else { size_t at = rnd() % m.size(), n = rnd() % 64; for (size_t k = 0; k < n && at + k < m.size(); k++) m[at + k] = (uint8_t)(rnd() % 3 ? 0xFF : rnd()); }
            try { auto img = mirhi::resources::decode_image(m.data(), m.size()); ok += img.rgba.size() == (size_t)img.width * img.height * 4; }
            catch (const mirhi::resources::ImageError&) { refused++; }
        }
        printf("%s: %ld mutants decoded, %ld refused\n", argv[a], ok, refused);
    }
    return 0;
}
