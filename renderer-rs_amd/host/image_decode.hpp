// image_decode.hpp -- PNG and JPEG (sequential and progressive Huffman) -> RGBA8, the texture-decode half of SURVEY.md section 8f rank 1.
//
// The reference pulls `image 0.25.9` (+ png 0.18, zune-jpeg 0.5.8) through `gltf::import` and then DISCARDS the decoded
// images (crates/resources/src/model.rs:120); its own texture types are stubs (crates/rhi/src/{image,sampler,texture}.rs
// :1-5) while every model shader samples `Texture2D` slots (shaders/hlsl/pixel/model_full.hlsl:44-46, model_pbr.hlsl).
// This header is what stands between `assets/textures/*.{png,jpg}` / a glTF's `images[]` and mirhi_image_upload:
//
//   decode_png   all colour types (0,2,3,4,6), bit depths 1..16, tRNS, Adam7; chunk CRCs and the zlib Adler-32 are checked.
//                16-bit samples narrow with the `image` crate's rule (c + 128) / 257.  Bit-exact by construction.
//   decode_jpeg  sequential (SOF0, SOF1; one interleaved scan or several) and progressive (SOF2: spectral selection and
//                successive approximation, end-of-band runs, refinement scans) Huffman files, 8-bit, 1 or 3 components,
//                any sampling factors, restart intervals.  Coefficients are accumulated over the scans, then every block
//                goes through the 13-bit fixed-point Loeffler-Ligtenberg-Moschytz inverse DCT, the 16-bit fixed-point
//                YCbCr->RGB step of the IJG decoder family and the triangle ("fancy") chroma upsampling for 2x1 and
//                2x2 -- so 4:4:4, grey and 4:2:x files decode to the same bytes as libjpeg-turbo (checked against Pillow
//                in tests/test_image_decode_cpu.py).  Arithmetic coding, lossless / hierarchical modes, 12-bit and CMYK
//                files are refused with a message, never decoded approximately.
//
// Header-only, no dependency (own inflate).  Errors throw ImageError; nothing here touches the GPU.
#ifndef MIRHI_IMAGE_DECODE_HPP
#define MIRHI_IMAGE_DECODE_HPP

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace mirhi {
namespace resources {

struct ImageError : std::runtime_error { using std::runtime_error::runtime_error; };

struct ImageData {
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> rgba;          // width*height*4, row-major, top row first
    uint32_t source_channels = 0;       // 1 grey, 2 grey+alpha, 3 rgb, 4 rgba (before expansion)
};

namespace detail {

// ------------------------------------------------------------------ inflate (RFC 1951) + zlib wrapper (RFC 1950)
struct BitReader {
    const uint8_t* p; size_t n, pos = 0; uint64_t acc = 0; int cnt = 0;
    BitReader(const uint8_t* d, size_t len) : p(d), n(len) {}
    inline void refill() { while (cnt <= 56 && pos < n) { acc |= (uint64_t)p[pos++] << cnt; cnt += 8; } }
    inline uint32_t peek(int bits) { if (cnt < bits) refill(); return (uint32_t)(acc & ((1ull << bits) - 1)); }
    inline void drop(int bits) { if (cnt < bits) throw ImageError("inflate: unexpected end of stream"); acc >>= bits; cnt -= bits; }
    inline uint32_t take(int bits) { if (!bits) return 0; uint32_t v = peek(bits); drop(bits); return v; }
    void align_byte() { int r = cnt & 7; acc >>= r; cnt -= r; }
};

struct Huff {                           // canonical code, LSB-first lookup: 10-bit direct table + bit-serial fallback
    static constexpr int FAST = 10;
    uint16_t fast[1 << FAST];           // (len << 12) | symbol, 0 = not a short code
    uint16_t count[16]; uint16_t symbol[320]; int max_len = 0;
    void build(const uint8_t* lens, int n) {
        memset(count, 0, sizeof count); memset(fast, 0, sizeof fast); max_len = 0;
        for (int i = 0; i < n; i++) { count[lens[i]]++; if (lens[i] > max_len) max_len = lens[i]; }
        count[0] = 0;
        int left = 1;
        for (int l = 1; l < 16; l++) { left <<= 1; left -= count[l]; if (left < 0) throw ImageError("inflate: over-subscribed Huffman code"); }
        uint16_t offs[16]; offs[1] = 0;
        for (int l = 1; l < 15; l++) offs[l + 1] = offs[l] + count[l];
        for (int i = 0; i < n; i++) if (lens[i]) symbol[offs[lens[i]]++] = (uint16_t)i;
        // direct table: canonical codes are MSB-first, the stream is LSB-first -> reverse
        uint32_t code = 0; int idx = 0;
        for (int l = 1; l <= FAST && l < 16; l++) {
            for (int k = 0; k < count[l]; k++, code++, idx++) {
                uint32_t rev = 0;
                for (int b = 0; b < l; b++) if (code & (1u << b)) rev |= 1u << (l - 1 - b);
                for (uint32_t fill = rev; fill < (1u << FAST); fill += 1u << l) fast[fill] = (uint16_t)((l << 12) | symbol[idx]);
            }
            code <<= 1;
        }
    }
    inline int decode(BitReader& br) const {
        uint32_t bits = br.peek(FAST);
        uint16_t e = fast[bits];
        if (e) { br.drop(e >> 12); return e & 0xFFF; }
        // long or invalid code: canonical walk
        int code = 0, first = 0, index = 0;
        for (int l = 1; l <= 15; l++) {
            code |= (int)br.take(1);
            int c = count[l];
            if (code - c < first) return symbol[index + (code - first)];
            index += c; first += c; first <<= 1; code <<= 1;
        }
        throw ImageError("inflate: invalid Huffman code");
    }
};

// Returns false if the stream holds more than `expected` bytes: output stops there (the surplus is ignored, as libpng and
// the `png` crate do) and the caller skips the checksum, which covers the whole stream.
inline bool inflate(const uint8_t* src, size_t len, std::vector<uint8_t>& out, size_t expected) {
    static const uint16_t LBASE[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
    static const uint8_t LEXT[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
    static const uint16_t DBASE[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
    static const uint8_t DEXT[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
    static const uint8_t ORDER[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};
    BitReader br(src, len);
    // deflate expands at most 1032:1, so a short stream cannot justify a large reservation whatever the header claims; and
    // nothing is produced beyond what the caller expects (a 60-byte PNG must not drive multi-GiB allocations)
    out.clear(); out.reserve(expected < len * 1032 + 64 ? expected : len * 1032 + 64);
    auto room = [&](size_t more) { return out.size() + more <= expected; };
    static thread_local Huff lit, dist;
    bool last = false;
    while (!last) {
        last = br.take(1) != 0;
        const uint32_t type = br.take(2);
        if (type == 0) {
            br.align_byte();
            const uint32_t n = br.take(16), nn = br.take(16);
            if ((n ^ nn) != 0xFFFF) throw ImageError("inflate: stored block length check failed");
            for (uint32_t i = 0; i < n; i++) { if (!room(1)) return false; out.push_back((uint8_t)br.take(8)); }
            continue;
        }
        if (type == 3) throw ImageError("inflate: reserved block type");
        uint8_t lens[320];
        if (type == 1) {
            int i = 0;
            for (; i < 144; i++) lens[i] = 8;
            for (; i < 256; i++) lens[i] = 9;
            for (; i < 280; i++) lens[i] = 7;
            for (; i < 288; i++) lens[i] = 8;
            lit.build(lens, 288);
            for (i = 0; i < 30; i++) lens[i] = 5;
            dist.build(lens, 30);
        } else {
            const int nlen = (int)br.take(5) + 257, ndist = (int)br.take(5) + 1, ncode = (int)br.take(4) + 4;
            if (nlen > 286 || ndist > 30) throw ImageError("inflate: bad code counts");
            uint8_t cl[19] = {0};
            for (int i = 0; i < ncode; i++) cl[ORDER[i]] = (uint8_t)br.take(3);
            Huff clh; clh.build(cl, 19);
            int i = 0;
            while (i < nlen + ndist) {
                int sym = clh.decode(br);
                if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
                int rep; uint8_t val = 0;
                if (sym == 16) { if (i == 0) throw ImageError("inflate: repeat with no previous length"); val = lens[i - 1]; rep = 3 + (int)br.take(2); }
                else if (sym == 17) rep = 3 + (int)br.take(3);
                else rep = 11 + (int)br.take(7);
                if (i + rep > nlen + ndist) throw ImageError("inflate: code length repeat overruns");
                while (rep--) lens[i++] = val;
            }
            if (lens[256] == 0) throw ImageError("inflate: no end-of-block code");
            lit.build(lens, nlen);
            dist.build(lens + nlen, ndist);
        }
        for (;;) {
            int sym = lit.decode(br);
            if (sym < 256) { if (!room(1)) return false; out.push_back((uint8_t)sym); continue; }
            if (sym == 256) break;
            sym -= 257;
            if (sym >= 29) throw ImageError("inflate: invalid length symbol");
            const size_t length = LBASE[sym] + br.take(LEXT[sym]);
            const int ds = dist.decode(br);
            if (ds >= 30) throw ImageError("inflate: invalid distance symbol");
            const size_t d = DBASE[ds] + br.take(DEXT[ds]);
            if (d > out.size()) throw ImageError("inflate: distance reaches before the start of the output");
            const size_t start = out.size() - d;
            const size_t fits = room(length) ? length : expected - out.size();
            out.resize(out.size() + fits);
            uint8_t* o = out.data();
            for (size_t k = 0; k < fits; k++) o[start + d + k] = o[start + k];
            if (fits < length) return false;
        }
    }
    return true;
}

inline uint32_t adler32(const uint8_t* p, size_t n) {
    uint32_t a = 1, b = 0;
    while (n) {
        size_t k = n < 5552 ? n : 5552; n -= k;
        while (k--) { a += *p++; b += a; }
        a %= 65521; b %= 65521;
    }
    return (b << 16) | a;
}

inline void zlib_decompress(const uint8_t* src, size_t len, std::vector<uint8_t>& out, size_t expected) {
    if (len < 6) throw ImageError("zlib: stream too short");
    if ((src[0] & 0x0F) != 8 || ((src[0] << 8 | src[1]) % 31) != 0) throw ImageError("zlib: bad header");
    if (src[1] & 0x20) throw ImageError("zlib: preset dictionary not allowed in PNG");
    if (!inflate(src + 2, len - 6, out, expected)) return;
    const uint32_t want = (uint32_t)src[len - 4] << 24 | (uint32_t)src[len - 3] << 16 | (uint32_t)src[len - 2] << 8 | src[len - 1];
#ifndef MIRHI_IMAGE_FUZZ_SKIP_CHECKS
    if (adler32(out.data(), out.size()) != want) throw ImageError("zlib: Adler-32 mismatch");
#else
    (void)want;
#endif
}

inline uint32_t crc32(const uint8_t* p, size_t n) {
    static uint32_t table[256]; static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; table[i] = c; }
        ready = true;
    }
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) c = table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

inline uint32_t be32(const uint8_t* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }
inline uint32_t be16(const uint8_t* p) { return (uint32_t)p[0] << 8 | p[1]; }

// ------------------------------------------------------------------ PNG
inline uint8_t paeth(int a, int b, int c) {
    const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
    return (uint8_t)((pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c));
}

// reverses the scanline filters of one (sub)image in place; `data` = h rows of (1 + rowbytes)
inline void png_unfilter(uint8_t* data, size_t h, size_t rowbytes, size_t bpp) {
    std::vector<uint8_t> zero(rowbytes, 0);
    const uint8_t* prev = zero.data();
    for (size_t y = 0; y < h; y++) {
        uint8_t* row = data + y * (rowbytes + 1);
        const uint8_t f = row[0];
        uint8_t* cur = row + 1;
        switch (f) {
        case 0: break;
        case 1: for (size_t i = bpp; i < rowbytes; i++) cur[i] = (uint8_t)(cur[i] + cur[i - bpp]); break;
        case 2: for (size_t i = 0; i < rowbytes; i++) cur[i] = (uint8_t)(cur[i] + prev[i]); break;
        case 3:
            for (size_t i = 0; i < rowbytes; i++) cur[i] = (uint8_t)(cur[i] + (((i >= bpp ? cur[i - bpp] : 0) + prev[i]) >> 1));
            break;
        case 4:
            for (size_t i = 0; i < rowbytes; i++)
                cur[i] = (uint8_t)(cur[i] + paeth(i >= bpp ? cur[i - bpp] : 0, prev[i], i >= bpp ? prev[i - bpp] : 0));
            break;
        default: throw ImageError("PNG: unknown filter type " + std::to_string(f));
        }
        prev = cur;
    }
}

struct PngInfo {
    uint32_t w = 0, h = 0; int depth = 0, color = 0, interlace = 0, channels = 0;
    std::vector<uint8_t> plte, trns;
};

inline uint32_t png_sample(const uint8_t* row, size_t index, int depth) {
    switch (depth) {
    case 8: return row[index];
    case 16: return be16(row + 2 * index);
    default: { const size_t bit = index * depth; return (row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1); }
    }
}
inline uint8_t narrow16(uint32_t c) { return (uint8_t)((c + 128) / 257); }        // image crate: u16 -> u8 with rounding

// writes one decoded pixel row (already unfiltered) into the RGBA image at (x0 + i*dx, y)
inline void png_store_row(const PngInfo& I, const uint8_t* row, uint32_t count, uint8_t* rgba, uint32_t y, uint32_t x0, uint32_t dx) {
    const int d = I.depth;
    const uint32_t maxv = (1u << d) - 1;
    for (uint32_t i = 0; i < count; i++) {
        uint8_t* o = rgba + ((size_t)y * I.w + x0 + (size_t)i * dx) * 4;
        switch (I.color) {
        case 0: {
            const uint32_t v = png_sample(row, i, d);
            const uint8_t g = d == 16 ? narrow16(v) : (uint8_t)(v * 255u / maxv);
            o[0] = o[1] = o[2] = g;
            o[3] = (I.trns.size() >= 2 && v == (be16(I.trns.data()) & maxv)) ? 0 : 255;
            break; }
        case 2: {
            const uint32_t r = png_sample(row, 3 * (size_t)i, d), g = png_sample(row, 3 * (size_t)i + 1, d), b = png_sample(row, 3 * (size_t)i + 2, d);
            if (d == 16) { o[0] = narrow16(r); o[1] = narrow16(g); o[2] = narrow16(b); } else { o[0] = (uint8_t)r; o[1] = (uint8_t)g; o[2] = (uint8_t)b; }
            o[3] = (I.trns.size() >= 6 && r == (be16(I.trns.data()) & maxv) && g == (be16(I.trns.data() + 2) & maxv) && b == (be16(I.trns.data() + 4) & maxv)) ? 0 : 255;
            break; }
        case 3: {
            const uint32_t v = png_sample(row, i, d);
            if (3 * (size_t)v + 2 >= I.plte.size()) throw ImageError("PNG: palette index " + std::to_string(v) + " out of range");
            o[0] = I.plte[3 * v]; o[1] = I.plte[3 * v + 1]; o[2] = I.plte[3 * v + 2];
            o[3] = v < I.trns.size() ? I.trns[v] : 255;
            break; }
        case 4: {
            const uint32_t v = png_sample(row, 2 * (size_t)i, d), a = png_sample(row, 2 * (size_t)i + 1, d);
            o[0] = o[1] = o[2] = d == 16 ? narrow16(v) : (uint8_t)v; o[3] = d == 16 ? narrow16(a) : (uint8_t)a;
            break; }
        default: {
            for (int c = 0; c < 4; c++) { const uint32_t v = png_sample(row, 4 * (size_t)i + c, d); o[c] = d == 16 ? narrow16(v) : (uint8_t)v; }
            break; }
        }
    }
}

}  // namespace detail

inline bool is_png(const uint8_t* p, size_t n) { static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A}; return n >= 8 && memcmp(p, sig, 8) == 0; }
inline bool is_jpeg(const uint8_t* p, size_t n) { return n >= 3 && p[0] == 0xFF && p[1] == 0xD8 && p[2] == 0xFF; }

inline ImageData decode_png(const uint8_t* p, size_t n) {
    using namespace detail;
    if (!is_png(p, n)) throw ImageError("PNG: bad signature");
    PngInfo I; std::vector<uint8_t> idat; bool have_ihdr = false, have_iend = false;
    size_t pos = 8;
    while (pos + 12 <= n && !have_iend) {
        const uint32_t len = be32(p + pos);
        if ((size_t)len > n - pos - 12) throw ImageError("PNG: chunk runs past the end of the file");
        const uint8_t* type = p + pos + 4; const uint8_t* body = p + pos + 8;
#ifndef MIRHI_IMAGE_FUZZ_SKIP_CHECKS     /* sanitizer fuzz builds reach the inflate / unfilter code behind the checksums */
        if (crc32(type, (size_t)len + 4) != be32(body + len)) throw ImageError(std::string("PNG: CRC mismatch in chunk ") + std::string((const char*)type, 4));
#endif
        const std::string t((const char*)type, 4);
        if (!have_ihdr && t != "IHDR") throw ImageError("PNG: first chunk is not IHDR");
        if (t == "IHDR") {
            if (len != 13) throw ImageError("PNG: bad IHDR length");
            I.w = be32(body); I.h = be32(body + 4); I.depth = body[8]; I.color = body[9]; I.interlace = body[12];
            if (I.w == 0 || I.h == 0) throw ImageError("PNG: zero dimension");
            if ((uint64_t)I.w * I.h > (1ull << 28)) throw ImageError("PNG: image larger than 2^28 pixels");
            if (body[10] != 0 || body[11] != 0 || I.interlace > 1) throw ImageError("PNG: unknown compression, filter or interlace method");
            static const int CH[7] = {1, 0, 3, 1, 2, 0, 4};
            if (I.color > 6 || CH[I.color] == 0) throw ImageError("PNG: invalid colour type " + std::to_string(I.color));
            I.channels = CH[I.color];
            const int d = I.depth;
            const bool ok = (I.color == 0 && (d == 1 || d == 2 || d == 4 || d == 8 || d == 16)) || (I.color == 3 && (d == 1 || d == 2 || d == 4 || d == 8)) ||
                            ((I.color == 2 || I.color == 4 || I.color == 6) && (d == 8 || d == 16));
            if (!ok) throw ImageError("PNG: bit depth " + std::to_string(d) + " not allowed for colour type " + std::to_string(I.color));
            have_ihdr = true;
        } else if (t == "PLTE") {
            if (len % 3 || len > 768) throw ImageError("PNG: bad PLTE length");
            I.plte.assign(body, body + len);
        } else if (t == "tRNS") {
            I.trns.assign(body, body + len);
        } else if (t == "IDAT") {
            idat.insert(idat.end(), body, body + len);
        } else if (t == "IEND") {
            have_iend = true;
        } else if (!(type[0] & 0x20)) {
            throw ImageError("PNG: unknown critical chunk " + t);
        }
        pos += (size_t)len + 12;
    }
    if (!have_ihdr || idat.empty()) throw ImageError("PNG: missing IHDR or IDAT");
    if (I.color == 3 && I.plte.empty()) throw ImageError("PNG: palette image without PLTE");
    if (I.color == 4 || I.color == 6) I.trns.clear();

    const size_t bits_pp = (size_t)I.channels * I.depth;
    const size_t bpp = bits_pp >= 8 ? bits_pp / 8 : 1;
    auto rowbytes = [&](uint32_t w) { return ((size_t)w * bits_pp + 7) / 8; };
    static const int X0[7] = {0, 4, 0, 2, 0, 1, 0}, Y0[7] = {0, 0, 4, 0, 2, 0, 1}, DX[7] = {8, 8, 4, 4, 2, 2, 1}, DY[7] = {8, 8, 8, 4, 4, 2, 2};
    size_t expected = 0;
    if (!I.interlace) expected = (size_t)I.h * (rowbytes(I.w) + 1);
    else for (int k = 0; k < 7; k++) {
        const uint32_t pw = (I.w + DX[k] - 1 - X0[k]) / DX[k], ph = (I.h + DY[k] - 1 - Y0[k]) / DY[k];
        if (pw && ph) expected += (size_t)ph * (rowbytes(pw) + 1);
    }
    std::vector<uint8_t> raw;
    zlib_decompress(idat.data(), idat.size(), raw, expected);
    if (raw.size() < expected) throw ImageError("PNG: image data too short (" + std::to_string(raw.size()) + " of " + std::to_string(expected) + " bytes)");

    ImageData out; out.width = I.w; out.height = I.h; out.source_channels = I.color == 3 ? (I.trns.empty() ? 3 : 4) : (uint32_t)I.channels + ((I.color == 0 || I.color == 2) && !I.trns.empty() ? 1 : 0);
    out.rgba.resize((size_t)I.w * I.h * 4);
    if (!I.interlace) {
        const size_t rb = rowbytes(I.w);
        png_unfilter(raw.data(), I.h, rb, bpp);
        for (uint32_t y = 0; y < I.h; y++) png_store_row(I, raw.data() + (size_t)y * (rb + 1) + 1, I.w, out.rgba.data(), y, 0, 1);
    } else {
        size_t off = 0;
        for (int k = 0; k < 7; k++) {
            const uint32_t pw = (I.w + DX[k] - 1 - X0[k]) / DX[k], ph = (I.h + DY[k] - 1 - Y0[k]) / DY[k];
            if (!pw || !ph) continue;
            const size_t rb = rowbytes(pw);
            png_unfilter(raw.data() + off, ph, rb, bpp);
            for (uint32_t y = 0; y < ph; y++) png_store_row(I, raw.data() + off + (size_t)y * (rb + 1) + 1, pw, out.rgba.data(), Y0[k] + y * DY[k], X0[k], DX[k]);
            off += (size_t)ph * (rb + 1);
        }
    }
    return out;
}

// ------------------------------------------------------------------ JPEG (ITU T.81 baseline / extended sequential, Huffman)
namespace detail {

struct JHuff {
    uint8_t bits[17] = {0}; uint8_t vals[256] = {0};
    uint16_t look[512];                // 9-bit lookahead: (len << 8) | value, 0 = longer code
    int32_t maxcode[18]; int32_t valptr[17]; bool present = false;
    void build() {
        uint8_t size[257]; uint16_t code[257]; int k = 0;
        for (int l = 1; l <= 16; l++) for (int i = 0; i < bits[l]; i++) size[k++] = (uint8_t)l;
        size[k] = 0; const int total = k;
        uint32_t c = 0; int si = size[0]; k = 0;
        while (size[k]) {
            while (size[k] == si) { code[k++] = (uint16_t)c; c++; }
            if (c > (1u << si)) throw ImageError("JPEG: invalid Huffman table");
            c <<= 1; si++;
        }
        int p = 0;
        for (int l = 1; l <= 16; l++) {
            if (bits[l]) { valptr[l] = p - (int)code[p]; p += bits[l]; maxcode[l] = code[p - 1]; } else maxcode[l] = -1;
        }
        maxcode[17] = 0x7FFFFFFF;
        memset(look, 0, sizeof look);
        for (int i = 0; i < total; i++) {
            if (size[i] > 9) continue;
            const int shift = 9 - size[i];
            for (int f = 0; f < (1 << shift); f++) look[(code[i] << shift) | f] = (uint16_t)((size[i] << 8) | vals[i]);
        }
        present = true;
    }
};

struct JBits {
    const uint8_t* p; size_t n, pos; uint32_t acc = 0; int cnt = 0; bool hit_marker = false;
    JBits(const uint8_t* d, size_t len, size_t start) : p(d), n(len), pos(start) {}
    inline void fill() {
        while (cnt <= 24) {
            uint32_t b = 0;
            if (!hit_marker && pos < n) {
                b = p[pos];
                if (b == 0xFF) {
                    const uint8_t nx = pos + 1 < n ? p[pos + 1] : 0xD9;
                    if (nx == 0) pos += 2;
                    else { hit_marker = true; b = 0; }        // leave pos at the marker; feed zeros (T.81 F.2.2.5)
                } else pos++;
            }
            acc |= b << (24 - cnt); cnt += 8;
        }
    }
    inline uint32_t peek(int k) { if (cnt < k) fill(); return acc >> (32 - k); }
    inline void drop(int k) { acc <<= k; cnt -= k; }
    inline int receive_extend(int s) {
        if (!s) return 0;
        if (cnt < s) fill();
        const int v = (int)(acc >> (32 - s)); drop(s);
        return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
    }
    inline int bits(int k) { if (cnt < k) fill(); const int v = (int)(acc >> (32 - k)); drop(k); return v; }   // 1 <= k <= 16
    inline int decode(const JHuff& h) {
        if (cnt < 16) fill();
        const uint16_t e = h.look[acc >> 23];
        if (e) { drop(e >> 8); return e & 0xFF; }
        int l = 10; int32_t code = (int32_t)(acc >> 22);
        while (code > h.maxcode[l]) { l++; if (l > 16) throw ImageError("JPEG: corrupt Huffman code"); code = (int32_t)(acc >> (32 - l)); }
        drop(l);
        return h.vals[(code + h.valptr[l]) & 0xFF];
    }
    void reset() { acc = 0; cnt = 0; hit_marker = false; }
};

static const uint8_t ZIGZAG[64] = {0,1,8,16,9,2,3,10,17,24,32,25,18,11,4,5,12,19,26,33,40,48,41,34,27,20,13,6,7,14,21,28,35,42,49,56,57,50,43,36,29,22,15,23,30,37,44,51,58,59,52,45,38,31,39,46,53,60,61,54,47,55,62,63};

inline uint8_t clamp255(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
// coefficient * quantiser; an 8-bit file stays far inside +-2^17, so the clamp only ever touches corrupt data
inline int32_t dequant(int coef, uint16_t q) { const int64_t v = (int64_t)coef * q; return (int32_t)(v > (1 << 17) ? (1 << 17) : (v < -(1 << 17) ? -(1 << 17) : v)); }

// 8x8 inverse DCT, 13-bit fixed-point LL&M (the "slow integer" method of the IJG family); in = dequantised coefficients.
// Carried in 64 bits: a well-formed file never leaves the 32-bit range (same results), a corrupt one cannot overflow.
inline uint8_t clamp255(int64_t v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
inline void idct8x8(const int32_t* in, uint8_t* out, size_t stride) {
    using I = int64_t;
    constexpr int CB = 13, P1 = 2;
    constexpr I F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633, F1_501 = 12299,
                      F1_847 = 15137, F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
    I ws[64];
    for (int c = 0; c < 8; c++) {
        const int32_t* s = in + c; I* w = ws + c;
        if (!(s[8] | s[16] | s[24] | s[32] | s[40] | s[48] | s[56])) {
            const I dc = s[0] * (1 << P1);
            for (int r = 0; r < 8; r++) w[8 * r] = dc;
            continue;
        }
        I z2 = s[16], z3 = s[48];
        I z1 = (z2 + z3) * F0_541;
        I t2 = z1 + z3 * (-F1_847), t3 = z1 + z2 * F0_765;
        z2 = s[0]; z3 = s[32];
        I t0 = (z2 + z3) * (1 << CB), t1 = (z2 - z3) * (1 << CB);
        const I t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
        t0 = s[56]; t1 = s[40]; t2 = s[24]; t3 = s[8];
        z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2; I z4 = t1 + t3;
        const I z5 = (z3 + z4) * F1_175;
        t0 *= F0_298; t1 *= F2_053; t2 *= F3_072; t3 *= F1_501;
        z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
        z3 += z5; z4 += z5;
        t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
        constexpr int SH = CB - P1; constexpr I RND = 1 << (SH - 1);
        w[0] = (t10 + t3 + RND) >> SH; w[56] = (t10 - t3 + RND) >> SH;
        w[8] = (t11 + t2 + RND) >> SH; w[48] = (t11 - t2 + RND) >> SH;
        w[16] = (t12 + t1 + RND) >> SH; w[40] = (t12 - t1 + RND) >> SH;
        w[24] = (t13 + t0 + RND) >> SH; w[32] = (t13 - t0 + RND) >> SH;
    }
    for (int r = 0; r < 8; r++) {
        const I* w = ws + 8 * r; uint8_t* o = out + r * stride;
        constexpr int SH = CB + P1 + 3; constexpr I RND = 1 << (SH - 1);
        if (!(w[1] | w[2] | w[3] | w[4] | w[5] | w[6] | w[7])) {
            const uint8_t dc = clamp255(((w[0] + (1 << (P1 + 2))) >> (P1 + 3)) + 128);
            for (int c = 0; c < 8; c++) o[c] = dc;
            continue;
        }
        I z2 = w[2], z3 = w[6];
        I z1 = (z2 + z3) * F0_541;
        I t2 = z1 + z3 * (-F1_847), t3 = z1 + z2 * F0_765;
        I t0 = (w[0] + w[4]) * (1 << CB), t1 = (w[0] - w[4]) * (1 << CB);
        const I t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
        t0 = w[7]; t1 = w[5]; t2 = w[3]; t3 = w[1];
        z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2; I z4 = t1 + t3;
        const I z5 = (z3 + z4) * F1_175;
        t0 *= F0_298; t1 *= F2_053; t2 *= F3_072; t3 *= F1_501;
        z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
        z3 += z5; z4 += z5;
        t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
        o[0] = clamp255(((t10 + t3 + RND) >> SH) + 128); o[7] = clamp255(((t10 - t3 + RND) >> SH) + 128);
        o[1] = clamp255(((t11 + t2 + RND) >> SH) + 128); o[6] = clamp255(((t11 - t2 + RND) >> SH) + 128);
        o[2] = clamp255(((t12 + t1 + RND) >> SH) + 128); o[5] = clamp255(((t12 - t1 + RND) >> SH) + 128);
        o[3] = clamp255(((t13 + t0 + RND) >> SH) + 128); o[4] = clamp255(((t13 - t0 + RND) >> SH) + 128);
    }
}

struct JComp {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, pred = 0;
    uint32_t bw = 0, bh = 0;            // blocks per row / column (padded to whole MCUs)
    uint32_t cw = 0, chh = 0;           // real (downsampled) sample dimensions
    std::vector<uint8_t> plane;         // bw*8 x bh*8
    std::vector<int16_t> coef;          // bw*bh blocks of 64 coefficients in natural order, accumulated over the scans
    uint16_t q[64]; bool latched = false;   // the component's quantiser (natural order), fixed at its first scan
};

// triangle-filter 2x horizontal upsampling of one row of `n` samples into 2n
inline void upsample_h2(const uint8_t* in, uint32_t n, uint8_t* out) {
    if (n == 1) { out[0] = out[1] = in[0]; return; }
    out[0] = in[0]; out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
    for (uint32_t i = 1; i + 1 < n; i++) {
        const int v = in[i] * 3;
        out[2 * i] = (uint8_t)((v + in[i - 1] + 1) >> 2); out[2 * i + 1] = (uint8_t)((v + in[i + 1] + 2) >> 2);
    }
    out[2 * n - 2] = (uint8_t)((in[n - 1] * 3 + in[n - 2] + 1) >> 2); out[2 * n - 1] = in[n - 1];
}
// triangle-filter 2x2 upsampling: `near` is the row itself, `far` the neighbour row above (even output row) or below (odd)
inline void upsample_h2v2_row(const uint8_t* near, const uint8_t* far, uint32_t n, uint8_t* out) {
    if (n == 1) { const int s = near[0] * 3 + far[0]; out[0] = (uint8_t)((s * 4 + 8) >> 4); out[1] = (uint8_t)((s * 4 + 7) >> 4); return; }
    int cur = near[0] * 3 + far[0], next = near[1] * 3 + far[1], last;
    out[0] = (uint8_t)((cur * 4 + 8) >> 4); out[1] = (uint8_t)((cur * 3 + next + 7) >> 4);
    last = cur; cur = next;
    for (uint32_t i = 1; i + 1 < n; i++) {
        next = near[i + 1] * 3 + far[i + 1];
        out[2 * i] = (uint8_t)((cur * 3 + last + 8) >> 4); out[2 * i + 1] = (uint8_t)((cur * 3 + next + 7) >> 4);
        last = cur; cur = next;
    }
    out[2 * n - 2] = (uint8_t)((cur * 3 + last + 8) >> 4); out[2 * n - 1] = (uint8_t)((cur * 4 + 7) >> 4);
}

}  // namespace detail

inline ImageData decode_jpeg(const uint8_t* p, size_t n) {
    using namespace detail;
    if (!is_jpeg(p, n)) throw ImageError("JPEG: bad signature");
    uint16_t qt[4][64]; bool have_qt[4] = {false, false, false, false};
    static thread_local JHuff dc[4], ac[4];
    for (int i = 0; i < 4; i++) { dc[i].present = false; ac[i].present = false; }
    std::vector<JComp> comps; uint32_t W = 0, H = 0; int hmax = 1, vmax = 1; uint32_t restart = 0;
    int adobe_transform = -1; bool have_sof = false, jfif = false, progressive = false, saw_scan = false, saw_eoi = false;
    uint32_t mcux = 0, mcuy = 0;
    size_t pos = 2;
    while (!saw_eoi) {
        if (pos + 2 > n) { if (saw_scan) break; throw ImageError("JPEG: no scan data before the end of the file"); }   // (a missing EOI is tolerated, as libjpeg does)
        if (p[pos] != 0xFF) throw ImageError("JPEG: expected a marker at byte " + std::to_string(pos));
        while (pos < n && p[pos] == 0xFF) pos++;            // fill bytes
        if (pos >= n) { if (saw_scan) break; throw ImageError("JPEG: truncated marker"); }
        const uint8_t m = p[pos++];
        if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
        if (m == 0xD9) { saw_eoi = true; break; }
        if (pos + 2 > n) throw ImageError("JPEG: truncated segment");
        const size_t len = be16(p + pos);
        if (len < 2 || pos + len > n) throw ImageError("JPEG: segment runs past the end of the file");
        const uint8_t* s = p + pos + 2; const size_t sl = len - 2;
        if (m == 0xDB) {                                     // DQT
            size_t i = 0;
            while (i < sl) {
                const int pq = s[i] >> 4, tq = s[i] & 15; i++;
                if (tq > 3 || pq > 1) throw ImageError("JPEG: bad quantisation table header");
                if (i + (pq ? 128 : 64) > sl) throw ImageError("JPEG: truncated quantisation table");
                for (int k = 0; k < 64; k++) { qt[tq][ZIGZAG[k]] = pq ? (uint16_t)be16(s + i + 2 * k) : s[i + k]; }
                i += pq ? 128 : 64; have_qt[tq] = true;
            }
        } else if (m == 0xC4) {                              // DHT
            size_t i = 0;
            while (i < sl) {
                if (i + 17 > sl) throw ImageError("JPEG: truncated Huffman table");
                const int tc = s[i] >> 4, th = s[i] & 15; i++;
                if (tc > 1 || th > 3) throw ImageError("JPEG: bad Huffman table header");
                JHuff& h = tc ? ac[th] : dc[th];
                int total = 0; h.bits[0] = 0;
                for (int l = 1; l <= 16; l++) { h.bits[l] = s[i + l - 1]; total += h.bits[l]; }
                i += 16;
                if (total > 256 || i + total > sl) throw ImageError("JPEG: truncated Huffman table");
                memset(h.vals, 0, sizeof h.vals); memcpy(h.vals, s + i, total); i += total;
                h.build();
            }
        } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {    // SOF0 / SOF1 (sequential) / SOF2 (progressive)
            if (have_sof) throw ImageError("JPEG: more than one frame header");
            if (sl < 6) throw ImageError("JPEG: truncated frame header");
            if (s[0] != 8) throw ImageError("JPEG: " + std::to_string(s[0]) + "-bit samples are not supported (8-bit only)");
            H = be16(s + 1); W = be16(s + 3); const int nc = s[5];
            if (!W || !H) throw ImageError("JPEG: zero dimension");
            if (nc != 1 && nc != 3) throw ImageError("JPEG: " + std::to_string(nc) + "-component images are not supported (grey or YCbCr/RGB only)");
            if (sl < (size_t)6 + 3 * nc) throw ImageError("JPEG: truncated frame header");
            comps.resize(nc);
            for (int c = 0; c < nc; c++) {
                comps[c].id = s[6 + 3 * c]; comps[c].h = s[7 + 3 * c] >> 4; comps[c].v = s[7 + 3 * c] & 15; comps[c].tq = s[8 + 3 * c];
                if (comps[c].h < 1 || comps[c].h > 4 || comps[c].v < 1 || comps[c].v > 4 || comps[c].tq > 3) throw ImageError("JPEG: bad component parameters");
                if (comps[c].h > hmax) hmax = comps[c].h;
                if (comps[c].v > vmax) vmax = comps[c].v;
            }
            if (nc == 1) { comps[0].h = comps[0].v = 1; hmax = vmax = 1; }     // a single-component image is never interleaved
            progressive = m == 0xC2;
            mcux = (W + 8 * hmax - 1) / (8 * hmax); mcuy = (H + 8 * vmax - 1) / (8 * vmax);
            if ((uint64_t)mcux * mcuy * hmax * vmax * 64 * comps.size() > (1ull << 30)) throw ImageError("JPEG: image too large");
            for (auto& c : comps) {
                c.bw = mcux * c.h; c.bh = mcuy * c.v;
                c.cw = (uint32_t)(((uint64_t)W * c.h + hmax - 1) / hmax); c.chh = (uint32_t)(((uint64_t)H * c.v + vmax - 1) / vmax);
                c.coef.assign((size_t)c.bw * c.bh * 64, 0);
            }
            have_sof = true;
        } else if (m == 0xC3 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) throw ImageError("JPEG: lossless / hierarchical / arithmetic-coded files are not supported");
        else if (m == 0xCC) throw ImageError("JPEG: arithmetic coding is not supported");
        else if (m == 0xDD) { if (sl < 2) throw ImageError("JPEG: truncated DRI"); restart = be16(s); }
        else if (m == 0xE0) { if (sl >= 5 && memcmp(s, "JFIF\0", 5) == 0) jfif = true; }
        else if (m == 0xEE) { if (sl >= 12 && memcmp(s, "Adobe", 5) == 0) adobe_transform = s[11]; }
        else if (m == 0xDA) {                                // SOS: one scan (the only one of a baseline file; one of many otherwise)
            if (!have_sof) throw ImageError("JPEG: scan before frame header");
            const int ns = sl ? s[0] : 0;
            if (ns < 1 || ns > (int)comps.size()) throw ImageError("JPEG: bad component count in scan header");
            if (sl < (size_t)1 + 2 * ns + 3) throw ImageError("JPEG: truncated scan header");
            JComp* sc[3];
            for (int k = 0; k < ns; k++) {
                const int cid = s[1 + 2 * k]; sc[k] = nullptr;
                for (auto& c : comps) if (c.id == cid) { c.td = s[2 + 2 * k] >> 4; c.ta = s[2 + 2 * k] & 15; sc[k] = &c; }
                if (!sc[k]) throw ImageError("JPEG: scan names an unknown component");
                for (int j = 0; j < k; j++) if (sc[j] == sc[k]) throw ImageError("JPEG: scan names a component twice");
                if (sc[k]->td > 3 || sc[k]->ta > 3) throw ImageError("JPEG: bad Huffman table selector");
                if (!sc[k]->latched) {                       // the quantiser in force at a component's first scan is the one it keeps
                    if (!have_qt[sc[k]->tq]) throw ImageError("JPEG: component uses an undefined quantisation table");
                    memcpy(sc[k]->q, qt[sc[k]->tq], sizeof sc[k]->q); sc[k]->latched = true;
                }
            }
            const int Ss = s[1 + 2 * ns], Se = s[2 + 2 * ns], Ah = s[3 + 2 * ns] >> 4, Al = s[3 + 2 * ns] & 15;
            if (!progressive) { if (Ss != 0 || Se != 63 || Ah != 0 || Al != 0) throw ImageError("JPEG: bad spectral selection in a sequential scan"); }
            else if (Ss > Se || Se > 63 || (Ss == 0 && Se != 0) || (Ss > 0 && ns != 1) || Al > 13 || Ah > 13) throw ImageError("JPEG: bad progressive scan parameters");
            const bool need_dc = Ss == 0 && Ah == 0, need_ac = Se > 0;
            for (int k = 0; k < ns; k++) {
                if (need_dc && !dc[sc[k]->td].present) throw ImageError("JPEG: scan uses an undefined DC Huffman table");
                if (need_ac && !ac[sc[k]->ta].present) throw ImageError("JPEG: scan uses an undefined AC Huffman table");
                sc[k]->pred = 0;
            }
            JBits br(p, n, pos + len);
            // a scan of one component is not interleaved: its blocks run in raster order over the component's own size
            const uint32_t bx_n = ns == 1 ? (sc[0]->cw + 7) / 8 : mcux, by_n = ns == 1 ? (sc[0]->chh + 7) / 8 : mcuy;
            uint32_t until_restart = restart, eobrun = 0; int next_rst = 0;
            for (uint32_t my = 0; my < by_n; my++) for (uint32_t mx = 0; mx < bx_n; mx++) {
                if (restart && until_restart == 0) {
                    br.reset();                              // byte-align, expect RSTn
                    size_t q = br.pos;
                    while (q + 1 < n && !(p[q] == 0xFF && p[q + 1] >= 0xD0 && p[q + 1] <= 0xD7)) {
                        if (p[q] == 0xFF && p[q + 1] != 0 && p[q + 1] != 0xFF) throw ImageError("JPEG: expected a restart marker");
                        q++;
                    }
                    if (q + 1 >= n) throw ImageError("JPEG: missing restart marker");
                    if ((p[q + 1] & 7) != next_rst) throw ImageError("JPEG: restart markers out of order");
                    next_rst = (next_rst + 1) & 7;
                    br.pos = q + 2;
                    for (int k = 0; k < ns; k++) sc[k]->pred = 0;
                    eobrun = 0;
                    until_restart = restart;
                }
                for (int k = 0; k < ns; k++) {
                    JComp& c = *sc[k];
                    const int nh = ns == 1 ? 1 : c.h, nv = ns == 1 ? 1 : c.v;
                    for (int by = 0; by < nv; by++) for (int bx = 0; bx < nh; bx++) {
                        int16_t* blk = c.coef.data() + ((size_t)(my * nv + by) * c.bw + (size_t)(mx * nh + bx)) * 64;
                        if (!progressive) {
                            const int t = br.decode(dc[c.td]);
                            if (t > 11) throw ImageError("JPEG: bad DC difference size");
                            c.pred += br.receive_extend(t);
                            if (c.pred > 32767 || c.pred < -32768) throw ImageError("JPEG: DC predictor out of range");
                            blk[0] = (int16_t)c.pred;
                            for (int kk = 1; kk < 64;) {
                                const int rs = br.decode(ac[c.ta]); const int r = rs >> 4, sz = rs & 15;
                                if (sz == 0) { if (r == 15) { kk += 16; continue; } break; }
                                kk += r;
                                if (kk > 63) throw ImageError("JPEG: AC coefficient index out of range");
                                blk[ZIGZAG[kk]] = (int16_t)br.receive_extend(sz);
                                kk++;
                            }
                        } else if (Ss == 0) {
                            if (Ah == 0) {                   // DC, first pass
                                const int t = br.decode(dc[c.td]);
                                if (t > 11) throw ImageError("JPEG: bad DC difference size");
                                c.pred += br.receive_extend(t);
                                if (c.pred > 32767 || c.pred < -32768) throw ImageError("JPEG: DC predictor out of range");
                                blk[0] = (int16_t)((uint16_t)c.pred << Al);
                            } else if (br.bits(1)) blk[0] = (int16_t)(blk[0] | (1 << Al));       // DC, refinement bit
                        } else if (Ah == 0) {                // AC band, first pass (T.81 G.1.2.2)
                            if (eobrun) { eobrun--; continue; }
                            for (int kk = Ss; kk <= Se;) {
                                const int rs = br.decode(ac[c.ta]); const int r = rs >> 4, sz = rs & 15;
                                if (sz == 0) {
                                    if (r < 15) { eobrun = (1u << r) - 1u + (r ? (uint32_t)br.bits(r) : 0u); break; }
                                    kk += 16; continue;
                                }
                                kk += r;
                                if (kk > Se) throw ImageError("JPEG: AC coefficient index out of range");
                                blk[ZIGZAG[kk]] = (int16_t)(br.receive_extend(sz) * (1 << Al));
                                kk++;
                            }
                        } else {                             // AC band, refinement (T.81 G.1.2.3)
                            const int p1 = 1 << Al, m1 = -(1 << Al);
                            int kk = Ss;
                            if (!eobrun) {
                                for (; kk <= Se; kk++) {
                                    const int rs = br.decode(ac[c.ta]); int r = rs >> 4; const int sz = rs & 15; int val = 0;
                                    if (sz) {
                                        if (sz != 1) throw ImageError("JPEG: corrupt refinement scan");
                                        val = br.bits(1) ? p1 : m1;
                                    } else if (r != 15) {
                                        eobrun = 1u << r;
                                        if (r) eobrun += (uint32_t)br.bits(r);
                                        break;
                                    }
                                    // skip r still-zero coefficients, handing correction bits to the nonzero ones on the way
                                    for (; kk <= Se; kk++) {
                                        int16_t& cf = blk[ZIGZAG[kk]];
                                        if (cf != 0) {
                                            if (br.bits(1) && (cf & p1) == 0) cf = (int16_t)(cf + (cf >= 0 ? p1 : m1));
                                        } else if (--r < 0) break;
                                    }
                                    if (val && kk <= Se) blk[ZIGZAG[kk]] = (int16_t)val;
                                }
                            }
                            if (eobrun) {                    // the rest of the band: correction bits only
                                for (; kk <= Se; kk++) {
                                    int16_t& cf = blk[ZIGZAG[kk]];
                                    if (cf != 0 && br.bits(1) && (cf & p1) == 0) cf = (int16_t)(cf + (cf >= 0 ? p1 : m1));
                                }
                                eobrun--;
                            }
                        }
                    }
                }
                if (restart) until_restart--;
            }
            // the next marker: the reader stopped at it or just before it (padding bits); restart markers belong to the scan
            size_t q = br.pos;
            while (q + 1 < n && !(p[q] == 0xFF && p[q + 1] != 0x00 && p[q + 1] != 0xFF && !(p[q + 1] >= 0xD0 && p[q + 1] <= 0xD7))) q++;
            pos = q + 1 < n ? q : n;
            saw_scan = true;
            if (!progressive && ns == (int)comps.size()) { /* the usual single interleaved scan: nothing else can follow but EOI */ }
            continue;
        }
        pos += len;
    }
    if (!saw_scan) throw ImageError("JPEG: no scan in the file");
    for (auto& c : comps) if (!c.latched) throw ImageError("JPEG: a component has no scan");
    // dequantise + inverse DCT, block by block
    {
        int32_t blk[64];
        for (auto& c : comps) {
            c.plane.assign((size_t)c.bw * 8 * c.bh * 8, 0);
            const size_t stride = (size_t)c.bw * 8;
            for (uint32_t by = 0; by < c.bh; by++) for (uint32_t bx = 0; bx < c.bw; bx++) {
                const int16_t* src = c.coef.data() + ((size_t)by * c.bw + bx) * 64;
                for (int k = 0; k < 64; k++) blk[k] = dequant(src[k], c.q[k]);
                idct8x8(blk, c.plane.data() + (size_t)by * 8 * stride + (size_t)bx * 8, stride);
            }
            std::vector<int16_t>().swap(c.coef);
        }
    }

    ImageData out; out.width = W; out.height = H; out.source_channels = (uint32_t)comps.size();
    out.rgba.resize((size_t)W * H * 4);
    if (comps.size() == 1) {
        const JComp& c = comps[0]; const size_t stride = (size_t)c.bw * 8;
        for (uint32_t y = 0; y < H; y++) for (uint32_t x = 0; x < W; x++) {
            const uint8_t g = c.plane[y * stride + x]; uint8_t* o = &out.rgba[((size_t)y * W + x) * 4];
            o[0] = o[1] = o[2] = g; o[3] = 255;
        }
        return out;
    }
    // bring every component to full resolution, one output row at a time
    // colour space as the IJG decoders infer it: JFIF => YCbCr; Adobe marker => its transform flag; else by component ids
    const bool ycc = jfif ? true : (adobe_transform >= 0 ? adobe_transform != 0 : !(comps[0].id == 'R' && comps[1].id == 'G' && comps[2].id == 'B'));
    std::vector<uint8_t> rows[3];
    for (int c = 0; c < 3; c++) rows[c].resize((size_t)comps[c].bw * 8 * (hmax / comps[c].h ? hmax / comps[c].h : 1) + 16);
    for (uint32_t y = 0; y < H; y++) {
        const uint8_t* src[3];
        for (int ci = 0; ci < 3; ci++) {
            const JComp& c = comps[ci]; const size_t stride = (size_t)c.bw * 8;
            const int hs = hmax / c.h, vs = vmax / c.v;
            if (hmax % c.h || vmax % c.v) throw ImageError("JPEG: fractional sampling ratios are not supported");
            if (hs == 1 && vs == 1) { src[ci] = c.plane.data() + y * stride; continue; }
            uint8_t* dst = rows[ci].data();
            // the triangle filter needs more than two source columns; narrower components replicate (as libjpeg does)
            if (hs == 2 && vs == 1 && c.cw > 2) upsample_h2(c.plane.data() + y * stride, c.cw, dst);
            else if (hs == 2 && vs == 2 && c.cw > 2) {
                const uint32_t cy = y >> 1;
                const uint32_t fy = (y & 1) ? (cy + 1 < c.chh ? cy + 1 : cy) : (cy ? cy - 1 : 0);
                upsample_h2v2_row(c.plane.data() + cy * stride, c.plane.data() + fy * stride, c.cw, dst);
            } else {                                              // any other ratio: sample replication
                const uint8_t* r = c.plane.data() + (size_t)(y / vs) * stride;
                for (uint32_t x = 0; x < W; x++) dst[x] = r[x / hs];
            }
            src[ci] = dst;
        }
        uint8_t* o = &out.rgba[(size_t)y * W * 4];
        if (ycc) {
            for (uint32_t x = 0; x < W; x++, o += 4) {
                const int Y = src[0][x], cb = src[1][x] - 128, cr = src[2][x] - 128;
                // 16-bit fixed point: 1.40200, 0.34414, 0.71414, 1.77200 scaled by 65536 and rounded
                o[0] = clamp255(Y + ((91881 * cr + 32768) >> 16));
                o[1] = clamp255(Y + ((-22554 * cb - 46802 * cr + 32768) >> 16));
                o[2] = clamp255(Y + ((116130 * cb + 32768) >> 16));
                o[3] = 255;
            }
        } else {
            for (uint32_t x = 0; x < W; x++, o += 4) { o[0] = src[0][x]; o[1] = src[1][x]; o[2] = src[2][x]; o[3] = 255; }
        }
    }
    return out;
}

// sniffs the container (the reference goes through image::load_from_memory's format guess)
inline ImageData decode_image(const uint8_t* p, size_t n) {
    if (is_png(p, n)) return decode_png(p, n);
    if (is_jpeg(p, n)) return decode_jpeg(p, n);
    throw ImageError("Unsupported image format (PNG and JPEG are decoded)");
}

inline ImageData load_image(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw ImageError("File not found: " + path);
    std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    try { return decode_image(bytes.data(), bytes.size()); }
    catch (const ImageError& e) { throw ImageError(path + ": " + e.what()); }
}

}  // namespace resources
}  // namespace mirhi
#endif
