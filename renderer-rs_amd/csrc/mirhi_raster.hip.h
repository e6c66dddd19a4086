// mirhi_raster.hip.h -- raster_kernel: tile records, coverage, depth key, resolve (rows a6, a7, a9)
// Part of the single device translation unit mirhi_kernels.hip (included inside namespace mirhi).
#ifndef MIRHI_RASTER_HIP_H
#define MIRHI_RASTER_HIP_H

#pragma clang fp contract(off)

#ifndef MIRHI_STAGE_PRIO
#define MIRHI_STAGE_PRIO 0
#endif

// ------------------------------------------------------------------------------------------------
// raster kernel
// ------------------------------------------------------------------------------------------------
struct PixelState { uint32_t zk[4], idk[4]; };   // (depth key, id key) per owned pixel; lexicographic minimum wins
struct RecRegs { uint4 w0, w1, w2, w3; };

__device__ __forceinline__ RecRegs load_rec(const uint4* lds_rec, uint32_t j) {
    RecRegs r;
    r.w0 = lds_rec[j * 4u + 0]; r.w1 = lds_rec[j * 4u + 1]; r.w2 = lds_rec[j * 4u + 2]; r.w3 = lds_rec[j * 4u + 3];
    return r;
}

// d = a * b + c with 24-bit signed a, b (full-rate integer multiply-add)
__device__ __forceinline__ int32_t mad24(int32_t a, int32_t b, int32_t c) {
    int32_t d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// 48-bit product of two signed 24-bit values
__device__ __forceinline__ int64_t mul24x24(int32_t a, int32_t b) {
    const uint32_t lo = (uint32_t)__mul24(a, b);     // low 32 bits of the product (operands fit 24 bits)
    const int32_t hi = __mulhi(a, b);                // high 32 bits of the 64-bit product
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | lo);
}

// A triangle in TILE-RELATIVE coordinates (1/256 px from the tile's origin) -> the tile record the pixel loops work on;
// false if no 8x8 block of the tile can be touched.
//   w0 = { Q0, Q1, Q2, A0 }   Q_i = floor((E_i(tile origin pixel centre) + bias_i) / 256), clamped to +-2^30
//   w1 = { A1, A2, B0, B1 }   A_i = Ya - Yb, B_i = Xb - Xa in 1/256 px (|.| < 2^23, fits v_mad_i32_i24)
//   w2 = { B2, dxt, dyt, z0 } (tile origin pixel centre) - (snapped vertex 0), in pixels (exact), vertex-0 depth
//   w3 = { zx, zy, idk, mask } mask bits 0..15 = 8x8 blocks the triangle may touch, bit 31 = pixel box applies
// box: the triangle's inclusive pixel box clamped to the tile, bx0 | bx1 << 8 | by0 << 16 | by1 << 24 (bx0 > bx1 or by0 > by1: it
// misses the tile); boxed: 0x80000000 if a scissor cut the box (the pixel loops then test it per pixel)
struct TileTri { int32_t X[3], Y[3]; uint32_t box, z0, zx, zy, idk, boxed; };
__device__ __forceinline__ uint32_t clamp_box(int32_t bx0, int32_t bx1, int32_t by0, int32_t by1) {
    // an empty intersection keeps lo > hi after the clamps: lo in [0, 32], hi in [-1, 31]
    bx0 = bx0 < 0 ? 0 : (bx0 > TILE ? TILE : bx0); by0 = by0 < 0 ? 0 : (by0 > TILE ? TILE : by0);
    bx1 = bx1 > TILE - 1 ? TILE - 1 : (bx1 < -1 ? -1 : bx1); by1 = by1 > TILE - 1 ? TILE - 1 : (by1 < -1 ? -1 : by1);
    return (uint32_t)bx0 | (((uint32_t)bx1 & 0xFFu) << 8) | ((uint32_t)by0 << 16) | ((uint32_t)by1 << 24);
}

// The arithmetic of a tile record, ONE copy for both record sources (bin records relative to their tile, big-list / ordered records in
// absolute screen coordinates): X / Y the snapped vertices and (Ptx, Pty) the tile origin's pixel centre in the same 1/256-pixel frame,
// (bx0..by1) the inclusive pixel box relative to the tile (already clamped; lo > hi: it misses the tile), (fox, foy) the tile origin in
// pixels of that frame.  Forced inline: every caller keeps its own register allocation (the TRIANGLE-only variants live on 64 VGPRs).
__device__ __forceinline__ bool tile_rec_core(uint4 out[4], const int32_t X[3], const int32_t Y[3], int32_t Ptx, int32_t Pty,
                                              int32_t bx0, int32_t bx1, int32_t by0, int32_t by1, float fox, float foy,
                                              uint32_t z0, uint32_t zx, uint32_t zy, uint32_t idk, uint32_t boxed) {
    int32_t A[3], B[3], Q[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int a = i, b = (i + 1) % 3;
        const int32_t dx = X[b] - X[a], dy = Y[b] - Y[a];
        A[i] = -dy; B[i] = dx;
        const bool topleft = (dy < 0) || (dy == 0 && dx > 0);              // top-left fill rule
        // E_i + bias at the tile origin; every factor fits 24 bits (|X|,|Y| < 2^22 and |tile origin| < 2^21 in absolute terms)
        const int64_t e0 = mul24x24(A[i], Ptx - X[a]) + mul24x24(B[i], Pty - Y[a]) + (topleft ? 0 : -1);
        int64_t q = e0 >> 8;                                 // floor(E/256): E = 256*(q + A*ix + B*iy) + r, 0 <= r < 256
        q = q > (1 << 30) ? (1 << 30) : (q < -(1 << 30) ? -(1 << 30) : q);
        Q[i] = (int32_t)q;
    }
    // conservative 8x8 block mask: a block is dropped if it misses the pixel box or lies outside one edge.
    // Per edge the value at the most-inside pixel of block (bx,by) is c_i + 8*(A_i*bx + B_i*by).  Straight-line code:
    // three adds, one OR of the three edge values and one funnel shift that appends the sign bit (set = outside) per
    // block -- no compares, no branches, nothing on the scalar unit.
    int32_t c[3], a8[3], b8[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        c[i] = Q[i] + (A[i] >= 0 ? A[i] * (BLOCK - 1) : 0) + (B[i] >= 0 ? B[i] * (BLOCK - 1) : 0);
        a8[i] = A[i] * BLOCK; b8[i] = B[i] * BLOCK;
    }
    uint32_t outside = 0;                               // after the loop: bit (15 - (by*4+bx)) set <=> block outside an edge
#pragma unroll
    for (int by = 0; by < 4; by++) {
        int32_t v0 = c[0], v1 = c[1], v2 = c[2];
#pragma unroll
        for (int bx = 0; bx < 4; bx++) {
            outside = __builtin_amdgcn_alignbit(outside, (uint32_t)(v0 | v1 | v2), 31);   // (outside << 1) | sign
            v0 += a8[0]; v1 += a8[1]; v2 += a8[2];
        }
        c[0] += b8[0]; c[1] += b8[1]; c[2] += b8[2];
    }
    // pixel box -> block box -> mask of the blocks inside it (4 column bits replicated per row, row bits spread to nibbles)
    const uint32_t cols = ((2u << ((uint32_t)bx1 >> 3)) - 1u) & ~((1u << ((uint32_t)bx0 >> 3)) - 1u);          // bits bx0b..bx1b
    const uint32_t rows = ((2u << ((uint32_t)by1 >> 3)) - 1u) & ~((1u << ((uint32_t)by0 >> 3)) - 1u);
    const uint32_t rowsel = ((rows & 1u) * 0xFu) | ((rows & 2u) * 0x78u) | ((rows & 4u) * 0x3C0u) | ((rows & 8u) * 0x1E00u);
    const uint32_t inside = __builtin_bitreverse32(~outside) >> 16;         // bit (by*4+bx) set <=> not outside any edge
    const uint32_t mask = (bx0 <= bx1 && by0 <= by1) ? (inside & (cols * 0x1111u) & rowsel) : 0u;
    const float inv256 = 1.0f / 256.0f;
    const float dxt = (fox + 0.5f) - (float)X[0] * inv256;                  // exact: multiples of 2^-8 below 2^15
    const float dyt = (foy + 0.5f) - (float)Y[0] * inv256;
    out[0] = make_uint4((uint32_t)Q[0], (uint32_t)Q[1], (uint32_t)Q[2], (uint32_t)A[0]);
    out[1] = make_uint4((uint32_t)A[1], (uint32_t)A[2], (uint32_t)B[0], (uint32_t)B[1]);
    out[2] = make_uint4((uint32_t)B[2], __float_as_uint(dxt), __float_as_uint(dyt), z0);
    out[3] = make_uint4(zx, zy, idk, mask | boxed);
    return mask != 0;
}

// (opx, opy): the tile's origin in the pixel frame T.X / T.Y are given in -- (0, 0) for a bin record, which is relative to its
// tile; the tile's position in the target for a big-list record, whose coordinates are absolute.  Wave-uniform.
__device__ __forceinline__ bool make_tile_rec(uint4 out[4], uint32_t& box, const TileTri& T, int32_t opx = 0, int32_t opy = 0) {
    const int32_t bx0 = (int32_t)(T.box & 0xFFu), bx1 = (int32_t)(int8_t)((T.box >> 8) & 0xFFu);
    const int32_t by0 = (int32_t)((T.box >> 16) & 0xFFu), by1 = (int32_t)T.box >> 24;
    box = T.box;
    return tile_rec_core(out, T.X, T.Y, 256 * opx + 128, 256 * opy + 128, bx0, bx1, by0, by1, (float)opx, (float)opy, T.z0, T.zx, T.zy, T.idk, T.boxed);
}

// BinRec (already relative to its tile) -> TileTri; the pixel box is recomputed from the vertices (no scissor cuts a binned triangle)
__device__ __forceinline__ void tile_tri_from_bin(TileTri& T, const uint4 w0, const uint4 w1) {
    const int32_t ox = (int32_t)(w0.x << 16) >> 16, oy = (int32_t)w0.x >> 16;
    const uint32_t x0 = w0.y & 0xFFFFu, x1 = w0.z & 0xFFFFu, x2 = w0.w & 0xFFFFu, y0 = w0.y >> 16, y1 = w0.z >> 16, y2 = w0.w >> 16;
    T.X[0] = ox + (int32_t)x0; T.X[1] = ox + (int32_t)x1; T.X[2] = ox + (int32_t)x2;
    T.Y[0] = oy + (int32_t)y0; T.Y[1] = oy + (int32_t)y1; T.Y[2] = oy + (int32_t)y2;
    const int32_t xmax = ox + (int32_t)max(x0, max(x1, x2)), ymax = oy + (int32_t)max(y0, max(y1, y2));
    T.box = clamp_box((ox + 127) >> 8, (xmax - 128) >> 8, (oy + 127) >> 8, (ymax - 128) >> 8);     // as setup_triangle, in tile-relative pixels
    T.z0 = w1.x; T.zx = w1.y; T.zy = w1.z; T.idk = w1.w; T.boxed = 0u;
}
// TriRec (big list) -> TileTri for tile (tx, ty), coordinates left absolute (make_tile_rec takes the tile's origin); false if its
// pixel box misses the tile
__device__ __forceinline__ bool tile_tri_from_big(TileTri& T, const uint4 w0, const uint4 w1, const uint4 w2, int32_t tx, int32_t ty) {
    const int32_t opx = tx * TILE, opy = ty * TILE;
    const int32_t bx0 = (int32_t)(w2.z & 0x7FFFu) - opx, bx1 = (int32_t)((w2.z >> 16) & 0x7FFFu) - opx;
    const int32_t by0 = (int32_t)(w2.w & 0xFFFFu) - opy, by1 = (int32_t)(w2.w >> 16) - opy;
    if (bx1 < 0 || bx0 > TILE - 1 || by1 < 0 || by0 > TILE - 1) return false;
    T.box = clamp_box(bx0, bx1, by0, by1);
    T.X[0] = (int32_t)w0.x; T.Y[0] = (int32_t)w0.y; T.X[1] = (int32_t)w0.z; T.Y[1] = (int32_t)w0.w;
    T.X[2] = (int32_t)w1.x; T.Y[2] = (int32_t)w1.y;
    T.z0 = w1.z; T.zx = w1.w; T.zy = w2.x; T.idk = w2.y; T.boxed = w2.z & 0x80000000u;
    return true;
}
// TriRec (absolute screen coordinates: big list, ordered segments) -> tile record for tile (tx, ty); false if no 8x8 block of the
// tile can be touched.  The same arithmetic (tile_rec_core) with the tile's origin added back.
__device__ __forceinline__ bool make_tile_rec(uint4 out[4], uint32_t& box, const uint4 w0, const uint4 w1, const uint4 w2,
                                              int32_t tx, int32_t ty) {
    const int32_t X[3] = {(int32_t)w0.x, (int32_t)w0.z, (int32_t)w1.x}, Y[3] = {(int32_t)w0.y, (int32_t)w0.w, (int32_t)w1.y};
    const int32_t ox = tx * TILE, oy = ty * TILE;
    int32_t bx0 = (int32_t)(w2.z & 0x7FFFu) - ox, bx1 = (int32_t)((w2.z >> 16) & 0x7FFFu) - ox;
    int32_t by0 = (int32_t)(w2.w & 0xFFFFu) - oy, by1 = (int32_t)(w2.w >> 16) - oy;
    bx0 = bx0 < 0 ? 0 : bx0; by0 = by0 < 0 ? 0 : by0;
    bx1 = bx1 > TILE - 1 ? TILE - 1 : bx1; by1 = by1 > TILE - 1 ? TILE - 1 : by1;
    box = (uint32_t)bx0 | ((uint32_t)bx1 << 8) | ((uint32_t)by0 << 16) | ((uint32_t)by1 << 24);
    return tile_rec_core(out, X, Y, 256 * ox + 128, 256 * oy + 128, bx0, bx1, by0, by1, (float)ox, (float)oy, w1.z, w1.w, w2.x, w2.y, w2.z & 0x80000000u);
}

// The colour target is written once and never read back by this kernel: streaming ("nt") stores keep its 8 - 33 MB from piling up as
// dirty lines in the L2s, which the end-of-kernel release would have to write back before the frame's fence can signal.
// A streaming store that leaves a 64-byte line half written is paid twice at the memory (C4's raster kernel: 33 -> 60 MB written when the resolve
// stored one 4-byte pixel per lane and 8x8 block, i.e. 32-byte row segments): the resolve pairs the two side-by-side blocks of a wave and stores
// 8 bytes per lane, 64-byte row segments (store_pair); a lone 32-byte segment (the wide variant with one block per wave, edge tiles) stays a plain store.
#ifndef MIRHI_PLAIN_TARGET_STORES
__device__ __forceinline__ void store_target(uint32_t* p, uint32_t v) { __builtin_nontemporal_store(v, p); }
typedef unsigned long long target_pair_t __attribute__((aligned(4)));
__device__ __forceinline__ void store_target2(uint32_t* p, uint32_t lo, uint32_t hi) { __builtin_nontemporal_store(((unsigned long long)hi << 32) | lo, reinterpret_cast<target_pair_t*>(p)); }
#else
__device__ __forceinline__ void store_target(uint32_t* p, uint32_t v) { *p = v; }
__device__ __forceinline__ void store_target2(uint32_t* p, uint32_t lo, uint32_t hi) { p[0] = lo; p[1] = hi; }
#endif
// The two horizontally adjacent 8x8 blocks of a wave (left block's packed colours in `left`, right block's in `right`, lane = (x = lane & 7, y = lane >> 3) of its
// block) as ONE store of 8 bytes per lane: lanes x < 4 take pixels 2x, 2x + 1 of the left block, lanes x >= 4 pixels 2(x - 4), 2(x - 4) + 1 of the right one --
// every row of the 16-pixel strip goes out as one 64-byte segment.  row_left: this lane's row start + the LEFT block's first pixel, in pixels.
__device__ __forceinline__ void store_pair(uint32_t* target, uint32_t row_left, uint32_t left, uint32_t right, uint32_t lane) {
    const uint32_t x = lane & 7u;
    const int src = (int)((lane & ~7u) | ((x & 3u) << 1));
    const uint32_t la = (uint32_t)__shfl((int)left, src), lb = (uint32_t)__shfl((int)left, src + 1);
    const uint32_t ra = (uint32_t)__shfl((int)right, src), rb = (uint32_t)__shfl((int)right, src + 1);
    const bool lhs = x < 4u;
    store_target2(target + row_left + (lhs ? 2u * x : 8u + 2u * (x - 4u)), lhs ? la : ra, lhs ? lb : rb);
}

// coverage + depth resolve of one record against the 4 blocks (8x8 px each) this wave owns.
// KEYED = 0: depth key is the raw float bits (LESS / LESS_OR_EQUAL); 1: generic (zflip / zmask applied);
//         2: predicate against the scope's initial depth (see DESIGN.md "Depth key").
// d = (a << 3) + b in one instruction (8 = BLOCK: the step of an edge function from one 8x8 block to the next)
__device__ __forceinline__ int32_t step8(int32_t a, int32_t b) {
    int32_t d;
    asm("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// (a & mask) | (b & ~mask) in one instruction; the mask is wave-uniform
__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) {
    uint32_t d;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "s"(mask), "v"(a), "v"(b));
    return d;
}

// NB: 8x8 blocks this wave owns -- 4 (a 16x16 quadrant, four waves per tile) or 1 (the wide mesh variant: sixteen waves per tile)
template <int KEYED, bool BOXED, int NB = 4>
__device__ __forceinline__ void raster_record(const RecRegs& r, uint32_t box, int32_t ix0, int32_t iy0, float fix0,
                                              float fiy0, ParamsRef P, PixelState& st, uint32_t qbit0) {
    const int32_t A0 = (int32_t)r.w0.w, A1 = (int32_t)r.w1.x, A2 = (int32_t)r.w1.y;
    const int32_t B0 = (int32_t)r.w1.z, B1 = (int32_t)r.w1.w, B2 = (int32_t)r.w2.x;
    const float z0 = __uint_as_float(r.w2.w), zx = __uint_as_float(r.w3.x), zy = __uint_as_float(r.w3.y);
    const uint32_t idk = r.w3.z;
    const uint32_t m = __builtin_amdgcn_readfirstlane(r.w3.w);
    const int32_t s0 = mad24(B0, iy0, mad24(A0, ix0, (int32_t)r.w0.x));
    const int32_t s1 = mad24(B1, iy0, mad24(A1, ix0, (int32_t)r.w0.y));
    const int32_t s2 = mad24(B2, iy0, mad24(A2, ix0, (int32_t)r.w0.z));
    // pixel centre minus vertex 0, exact in binary32 (see make_tile_rec), for the two columns / rows of blocks
    const float dx0 = fix0 + __uint_as_float(r.w2.y), dy0 = fiy0 + __uint_as_float(r.w2.z);
#pragma unroll
    for (int b = 0; b < NB; b++) {
        const int bx = b & 1, by = b >> 1;
        if (!(m & (qbit0 << (by * 4 + bx)))) continue;
        // edge functions at this block: one shift-add per edge and step (v_lshl_add_u32), no shared shift results
        int32_t sgn;                                         // covered <=> sign bit clear
        if (!bx && !by) sgn = s0 | s1 | s2;
        else if (bx && !by) sgn = step8(A0, s0) | step8(A1, s1) | step8(A2, s2);
        else if (!bx && by) sgn = step8(B0, s0) | step8(B1, s1) | step8(B2, s2);
        else sgn = step8(B0, step8(A0, s0)) | step8(B1, step8(A1, s1)) | step8(B2, step8(A2, s2));
        const float dx = dx0 + (float)(bx * BLOCK), dy = dy0 + (float)(by * BLOCK);
        const float z = __builtin_fmaf(dy, zy, __builtin_fmaf(dx, zx, z0));
        // clamp to [0,1]: v_med3_f32 returns min3 = 0 when z is NaN; the mask turns a -0 result into +0
        const uint32_t zc = __float_as_uint(__builtin_amdgcn_fmed3f(z, 0.0f, 1.0f));
        uint32_t zk;
        bool upd;
        if (KEYED == 0 && !BOXED) {
            // plain key: depth bits are <= 0x3F800000, so a lane outside the triangle can carry its miss in the key's
            // top bit (such a key never beats a stored one) -- no separate compare, no mask AND.  One bit-field insert
            // takes the low 31 bits from the depth and the top bit from the edge functions' OR.
            zk = bfi(0x7FFFFFFFu, zc, (uint32_t)sgn);      // (an updating lane has the bit clear: zk is stored unchanged)
            upd = (((uint64_t)zk << 32) | idk) < (((uint64_t)st.zk[b] << 32) | st.idk[b]);
        } else {
            zk = zc & 0x7FFFFFFFu;
            bool inside = sgn >= 0;
            if (BOXED) {
                const int32_t ix = ix0 + bx * BLOCK, iy = iy0 + by * BLOCK;
                inside = inside && ix >= (int32_t)(box & 0xFF) && ix <= (int32_t)((box >> 8) & 0xFF) &&
                         iy >= (int32_t)((box >> 16) & 0xFF) && iy <= (int32_t)(box >> 24);
            }
            if (KEYED == 2) {
                // predicate mode (depth test without write, EQUAL, ALWAYS with write): the fragment is tested against the
                // depth the scope started with (kept in st.zk), the latest passing primitive wins (idk = MAX - id)
                const uint32_t pred = P.pred;
                const bool lt = zk < st.zk[b], eq = zk == st.zk[b];
                const bool pass = (lt && (pred & 1u)) || (eq && (pred & 2u)) || (!lt && !eq && (pred & 4u));
                // Two fragments of ONE primitive at a pixel (clip pieces that overlap after the snap): only ALWAYS-with-write shows which came
                // last, and records reach a tile in no fixed order -- the nearer one is kept, as the oracle does (a7 there).
                upd = inside && pass && (idk < st.idk[b] || ((pred & 8u) && idk == st.idk[b] && zk < st.zk[b]));
                if (!(pred & 8u)) zk = st.zk[b];           // only ALWAYS-with-write replaces the depth (by the winner's)
            } else {
                if (KEYED) zk = (zk ^ P.zflip) & P.zmask;
                upd = inside && (((uint64_t)zk << 32) | idk) < (((uint64_t)st.zk[b] << 32) | st.idk[b]);
            }
        }
        st.zk[b] = upd ? zk : st.zk[b];
        st.idk[b] = upd ? idk : st.idk[b];
    }
}

// bit 30 of a tile record's mask word: a record of an alpha-masked draw that is resolved pixel-parallel (see raster_record_masked)
#define MIRHI_REC_MASKED 0x40000000u

// One record of an alpha-masked Cook-Torrance draw against the four blocks of this wave, pixel-parallel (records whose pixel box is
// too large for the triangle-parallel walk of raster_small_masked): coverage as raster_record, then the fragment program's alpha
// (pbr_base_alpha: the shading path's bits) on the covered lanes, and only fragments it keeps compete by their depth key.  The
// triangle's clip positions and texture coordinates are fetched once per record (a uniform address).
template <int KEYED, int NB = 4>
__device__ __forceinline__ void raster_record_masked(const RecRegs& r, int32_t ix0, int32_t iy0, float fix0, float fiy0, ParamsRef P,
                                                     PixelState& st, uint32_t qbit0, uint32_t tx, uint32_t ty) {
    const int32_t A0 = (int32_t)r.w0.w, A1 = (int32_t)r.w1.x, A2 = (int32_t)r.w1.y;
    const int32_t B0 = (int32_t)r.w1.z, B1 = (int32_t)r.w1.w, B2 = (int32_t)r.w2.x;
    const float z0 = __uint_as_float(r.w2.w), zx = __uint_as_float(r.w3.x), zy = __uint_as_float(r.w3.y);
    const uint32_t idk = __builtin_amdgcn_readfirstlane(r.w3.z);
    const uint32_t m = __builtin_amdgcn_readfirstlane(r.w3.w);
    const uint32_t prim = P.idflip ? (MAX_PRIM_ID - idk) : idk;
    DrawRef D = const_draws(P.draws)[P.num_draws > 1 ? find_draw(P, prim) : 0u];
    uint32_t vin[3];
    fetch_triangle_indices(D, prim - D.prim_base, vin);
    f4 c[3]; float uvk[3][2];
#pragma unroll
    for (uint32_t k = 0; k < 3; k++) {
        const uint4 w0 = reinterpret_cast<const uint4*>(D.vs_out)[vin[k]];
        const uint4 w2 = (reinterpret_cast<const uint4*>(D.vs_attr) + (size_t)vin[k] * (D.vs_words - 1u))[1];
        c[k] = {__uint_as_float(w0.x), __uint_as_float(w0.y), __uint_as_float(w0.z), __uint_as_float(w0.w)};
        uvk[k][0] = __uint_as_float(w2.z); uvk[k][1] = __uint_as_float(w2.w);
    }
    const float cutoff = ldcf(cb(D.material), 44);
    const int32_t s0 = mad24(B0, iy0, mad24(A0, ix0, (int32_t)r.w0.x));
    const int32_t s1 = mad24(B1, iy0, mad24(A1, ix0, (int32_t)r.w0.y));
    const int32_t s2 = mad24(B2, iy0, mad24(A2, ix0, (int32_t)r.w0.z));
    const float dx0 = fix0 + __uint_as_float(r.w2.y), dy0 = fiy0 + __uint_as_float(r.w2.z);
    const float pxc0 = (float)(tx * TILE + (uint32_t)ix0) + 0.5f, pyc0 = (float)(ty * TILE + (uint32_t)iy0) + 0.5f;
#pragma unroll 1
    for (int b = 0; b < NB; b++) {
        const int bx = b & 1, by = b >> 1;
        if (!(m & (qbit0 << (by * 4 + bx)))) continue;
        const int32_t S0 = s0 + (A0 * bx + B0 * by) * BLOCK, S1 = s1 + (A1 * bx + B1 * by) * BLOCK, S2 = s2 + (A2 * bx + B2 * by) * BLOCK;
        bool inside = (S0 | S1 | S2) >= 0;
        if (__ballot(inside) == 0ull) continue;
        if (inside) inside = !(pbr_base_alpha(D, c, uvk, pxc0 + (float)(bx * BLOCK), pyc0 + (float)(by * BLOCK)) < cutoff);
        const float dx = dx0 + (float)(bx * BLOCK), dy = dy0 + (float)(by * BLOCK);
        const float z = __builtin_fmaf(dy, zy, __builtin_fmaf(dx, zx, z0));
        uint32_t zk = __float_as_uint(__builtin_amdgcn_fmed3f(z, 0.0f, 1.0f)) & 0x7FFFFFFFu;
        if (KEYED == 1) zk = (zk ^ P.zflip) & P.zmask;
        const uint32_t zo = b == 0 ? st.zk[0] : (b == 1 ? st.zk[1] : (b == 2 ? st.zk[2] : st.zk[3]));
        const uint32_t io = b == 0 ? st.idk[0] : (b == 1 ? st.idk[1] : (b == 2 ? st.idk[2] : st.idk[3]));
        const bool upd = inside && (((uint64_t)zk << 32) | idk) < (((uint64_t)zo << 32) | io);
        if (b == 0) { st.zk[0] = upd ? zk : zo; st.idk[0] = upd ? idk : io; }
        else if (b == 1) { st.zk[1] = upd ? zk : zo; st.idk[1] = upd ? idk : io; }
        else if (b == 2) { st.zk[2] = upd ? zk : zo; st.idk[2] = upd ? idk : io; }
        else { st.zk[3] = upd ? zk : zo; st.idk[3] = upd ? idk : io; }
    }
}

// all records of an LDS chunk: per 64 records one ballot builds the bitmap of records that touch this
// wave's quadrant; LDS latency of the broadcast record reads is hidden by the other waves of the SIMD
template <int KEYED, int TP, bool MASKED = false, int NB = 4>
__device__ __forceinline__ void raster_chunk(const uint4* lds_rec, const uint32_t* lds_box, uint32_t n, uint32_t qmask,
                                             int32_t ix0, int32_t iy0, float fix0, float fiy0, ParamsRef P,
                                             PixelState& st, uint32_t qbit0, uint32_t lane, uint32_t tx = 0u, uint32_t ty = 0u) {
    for (uint32_t g = 0; g < n; g += 64u) {
        const uint32_t j = g + lane;
        const uint32_t mymask = j < n ? lds_rec[j * 4u + 3u].w : 0u;
        const bool rel = (mymask & qmask) != 0u, boxed = !TP && (mymask & 0x80000000u) != 0u;
        const bool mrec = MASKED && (mymask & MIRHI_REC_MASKED) != 0u;
        if (MASKED) {        // (alpha-masked scopes only: no such record exists elsewhere)
            uint64_t mbits = __ballot(rel && mrec);
            while (mbits) {
                const uint32_t cur_j = g + (uint32_t)(__ffsll((long long)mbits) - 1);
                mbits &= mbits - 1;
                const RecRegs cur = load_rec(lds_rec, cur_j);
                raster_record_masked<KEYED, NB>(cur, ix0, iy0, fix0, fiy0, P, st, qbit0, tx, ty);
            }
        }
        uint64_t bits = __ballot(rel && !boxed && !mrec);
        while (bits) {
            const uint32_t bit = (uint32_t)(__ffsll((long long)bits) - 1);
            // clear the bit with one scalar instruction (the compiler's `bits &= bits - 1` is add, addc, and: the scalar unit issues
            // at the vector unit's rate on this part, and this loop runs once per record and wave)
            asm("s_bitset0_b64 %0, %1" : "+s"(bits) : "s"(bit));
            const RecRegs cur = load_rec(lds_rec, g + bit);
            raster_record<KEYED, false, NB>(cur, 0u, ix0, iy0, fix0, fiy0, P, st, qbit0);
        }
        if (!TP) {   // without the triangle-parallel path, scissor-cut triangles (rare) take the per-pixel box test here
            uint64_t bbits = __ballot(rel && boxed);
            while (bbits) {
                const uint32_t cur_j = g + (uint32_t)(__ffsll((long long)bbits) - 1);
                bbits &= bbits - 1;
                const RecRegs cur = load_rec(lds_rec, cur_j);
                raster_record<KEYED, true, NB>(cur, lds_box[cur_j], ix0, iy0, fix0, fiy0, P, st, qbit0);
            }
        }
    }
}

// Triangle-parallel resolve of ONE small record by the lane that built it: walks the record's pixel box inside the
// tile and merges covered pixels into the tile's LDS key array with 64-bit ds_min.  For tiles holding many small
// triangles this keeps all 64 lanes busy on different triangles, where the pixel-parallel loop above would spend a
// full wave iteration per triangle with a handful of lanes covered.  Same integers, same depth FMAs, same keys.
template <int KEYED>
__device__ __forceinline__ void raster_small(const uint4 rec[4], uint32_t box, unsigned long long* lds_key, ParamsRef P) {
    const int32_t A0 = (int32_t)rec[0].w, A1 = (int32_t)rec[1].x, A2 = (int32_t)rec[1].y;
    const int32_t B0 = (int32_t)rec[1].z, B1 = (int32_t)rec[1].w, B2 = (int32_t)rec[2].x;
    const float dxt = __uint_as_float(rec[2].y), dyt = __uint_as_float(rec[2].z), z0 = __uint_as_float(rec[2].w);
    const float zx = __uint_as_float(rec[3].x), zy = __uint_as_float(rec[3].y);
    const uint32_t idk = rec[3].z;
    const int32_t bx0 = (int32_t)(box & 0xFF), bx1 = (int32_t)((box >> 8) & 0xFF);
    const int32_t by0 = (int32_t)((box >> 16) & 0xFF), by1 = (int32_t)(box >> 24);
    int32_t r0 = mad24(B0, by0, mad24(A0, bx0, (int32_t)rec[0].x));
    int32_t r1 = mad24(B1, by0, mad24(A1, bx0, (int32_t)rec[0].y));
    int32_t r2 = mad24(B2, by0, mad24(A2, bx0, (int32_t)rec[0].z));
    // Two pixels per step: the two coverage tests / depth evaluations are independent instruction chains, which is what a
    // wave that sits alone on its SIMD (a mesh's hot tiles while most of the chip is idle) needs to keep issuing.
    const int32_t A0x2 = A0 + A0, A1x2 = A1 + A1, A2x2 = A2 + A2;
    for (int32_t iy = by0; iy <= by1; iy++) {
        int32_t s0 = r0, s1 = r1, s2 = r2;
        const float dy = (float)iy + dyt;
        for (int32_t ix = bx0; ix <= bx1; ix += 2) {
            const int32_t t0 = s0 + A0, t1 = s1 + A1, t2 = s2 + A2;
            const bool in_a = (s0 | s1 | s2) >= 0;
            const bool in_b = (t0 | t1 | t2) >= 0 && ix < bx1;
            if (in_a) {
                const float dx = (float)ix + dxt;
                const float z = __builtin_fmaf(dy, zy, __builtin_fmaf(dx, zx, z0));
                uint32_t zk = __float_as_uint(__builtin_amdgcn_fmed3f(z, 0.0f, 1.0f)) & 0x7FFFFFFFu;
                if (KEYED == 1) zk = (zk ^ P.zflip) & P.zmask;
                atomicMin(&lds_key[iy * TILE + ix], ((unsigned long long)zk << 32) | idk);
            }
            if (in_b) {
                const float dx = (float)(ix + 1) + dxt;
                const float z = __builtin_fmaf(dy, zy, __builtin_fmaf(dx, zx, z0));
                uint32_t zk = __float_as_uint(__builtin_amdgcn_fmed3f(z, 0.0f, 1.0f)) & 0x7FFFFFFFu;
                if (KEYED == 1) zk = (zk ^ P.zflip) & P.zmask;
                atomicMin(&lds_key[iy * TILE + ix + 1], ((unsigned long long)zk << 32) | idk);
            }
            s0 += A0x2; s1 += A1x2; s2 += A2x2;
        }
        r0 += B0; r1 += B1; r2 += B2;
    }
}

// The same walk for a record of an alpha-masked Cook-Torrance draw (PassParams::alpha_scope): the fragment program ends a fragment
// whose base-colour alpha is below the material's cutoff BEFORE the depth write (`discard`, pixel/model_pbr.hlsl:176-179), so the
// alpha is evaluated here, per covered pixel, in front of the key minimum -- with the shading path's own barycentrics, (u, v),
// footprint and lookup (pbr_base_alpha), i.e. the bits the oracle's fragment loop compares.  Visibility stays order-independent:
// a kept fragment competes by its depth key like any other.  D is wave-uniform (the caller's waterfall over the wave's draws).
template <int KEYED>
__device__ __forceinline__ void raster_small_masked(DrawRef D, uint32_t prim, const uint4 rec[4], uint32_t box, unsigned long long* lds_key,
                                                    ParamsRef P, uint32_t tx, uint32_t ty) {
    uint32_t vin[3];
    fetch_triangle_indices(D, prim - D.prim_base, vin);
    f4 c[3]; float uvk[3][2];
#pragma unroll
    for (uint32_t k = 0; k < 3; k++) {
        const uint4 w0 = reinterpret_cast<const uint4*>(D.vs_out)[vin[k]];
        const uint4 w2 = (reinterpret_cast<const uint4*>(D.vs_attr) + (size_t)vin[k] * (D.vs_words - 1u))[1];
        c[k] = {__uint_as_float(w0.x), __uint_as_float(w0.y), __uint_as_float(w0.z), __uint_as_float(w0.w)};
        uvk[k][0] = __uint_as_float(w2.z); uvk[k][1] = __uint_as_float(w2.w);
    }
    const float cutoff = ldcf(cb(D.material), 44);
    const int32_t A0 = (int32_t)rec[0].w, A1 = (int32_t)rec[1].x, A2 = (int32_t)rec[1].y;
    const int32_t B0 = (int32_t)rec[1].z, B1 = (int32_t)rec[1].w, B2 = (int32_t)rec[2].x;
    const float dxt = __uint_as_float(rec[2].y), dyt = __uint_as_float(rec[2].z), z0 = __uint_as_float(rec[2].w);
    const float zx = __uint_as_float(rec[3].x), zy = __uint_as_float(rec[3].y);
    const uint32_t idk = rec[3].z;
    const int32_t bx0 = (int32_t)(box & 0xFF), bx1 = (int32_t)((box >> 8) & 0xFF);
    const int32_t by0 = (int32_t)((box >> 16) & 0xFF), by1 = (int32_t)(box >> 24);
    int32_t r0 = mad24(B0, by0, mad24(A0, bx0, (int32_t)rec[0].x));
    int32_t r1 = mad24(B1, by0, mad24(A1, bx0, (int32_t)rec[0].y));
    int32_t r2 = mad24(B2, by0, mad24(A2, bx0, (int32_t)rec[0].z));
    const float px_base = (float)(tx * TILE) + 0.5f, py_base = (float)(ty * TILE) + 0.5f;      // (exact: integers below 2^14 plus a half)
#pragma unroll 1
    for (int32_t iy = by0; iy <= by1; iy++) {
        int32_t s0 = r0, s1 = r1, s2 = r2;
        const float dy = (float)iy + dyt;
#pragma unroll 1
        for (int32_t ix = bx0; ix <= bx1; ix++) {
            if ((s0 | s1 | s2) >= 0) {
                const float alpha = pbr_base_alpha(D, c, uvk, px_base + (float)ix, py_base + (float)iy);
                if (!(alpha < cutoff)) {
                    const float dx = (float)ix + dxt;
                    const float z = __builtin_fmaf(dy, zy, __builtin_fmaf(dx, zx, z0));
                    uint32_t zk = __float_as_uint(__builtin_amdgcn_fmed3f(z, 0.0f, 1.0f)) & 0x7FFFFFFFu;
                    if (KEYED == 1) zk = (zk ^ P.zflip) & P.zmask;
                    atomicMin(&lds_key[iy * TILE + ix], ((unsigned long long)zk << 32) | idk);
                }
            }
            s0 += A0; s1 += A1; s2 += A2;
        }
        r0 += B0; r1 += B1; r2 += B2;
    }
}

__device__ __forceinline__ void init_key(ParamsRef P, uint32_t px, uint32_t py, bool valid, uint32_t& zk,
                                         uint32_t& idk, uint32_t& zorig) {
    zk = P.init_zk; idk = P.init_idk; zorig = P.clear_depth_bits;
    if (P.depth_load && P.depth && valid) {
        const uint32_t bits = __float_as_uint(P.depth[(size_t)py * P.width + px]);
        zorig = bits;
        if (P.zmask) {
            const uint32_t t = bits ^ P.zflip;
            if (!P.strict) { zk = t; idk = NO_PRIM; }
            else if (t == 0u) { zk = 0u; idk = 0u; }
            else { zk = t - 1u; idk = NO_PRIM; }
        }
    }
}

// Stages up to RASTER_THREADS triangle records of `list` (bin or big list) into LDS as tile records
// (one record per lane, wave ballot + prefix popcount compaction), then resolves them.
// TEAMS > 1 (mesh variants): the workgroup is TEAMS sets of four waves; team t stages and rasters chunks t, t + TEAMS, ...
// of the list into its own staging area and its own register keys (tid = lane index within the team).  Every team runs
// the same number of passes -- the barriers are workgroup-wide -- a team whose chunk lies beyond the list stages nothing.
// seg (two-team variant, bin list only): the bin is eight per-XCD lists of at most sub_cap records; seg[k] = records in the
// sub-bins before k, so that flat index i lives in sub-bin #{k >= 1 : i >= seg[k]} at offset i - seg[that].  nullptr: a plain list.
// bins: `list` is the bin pool and flat index i of the tile's bin lives in pool page pages[...] (LDS copy of the tile's page-table
// row; the first page of a single-list bin is page `tile` itself); otherwise `list` is a plain TriRec array (the big list).
template <int KEYED, int TP, int CHUNK, int TEAMS, bool BINS, bool MASKED = false, int WPT = 4>
__device__ __forceinline__ void raster_list(const uint4* __restrict__ list, uint32_t n_total, uint32_t tile, uint32_t fixed_recs, const uint32_t* pages, const uint32_t* seg, uint4* lds_rec, uint32_t* lds_box,
                                            uint32_t* lds_count, uint32_t& flip, uint32_t team, uint32_t nteams, unsigned long long* lds_key, uint32_t tx, uint32_t ty,
                                            uint32_t qmask, int32_t ix0, int32_t iy0, float fix0, float fiy0,
                                            ParamsRef P, PixelState& st, uint32_t qbit0, uint32_t tid,
                                            uint32_t lane) {
    // Most bins hold fewer than 64 records, so one wave builds all tile records of a tile.  Which wave does it rotates
    // with the tile: wave k of every workgroup sits on the same SIMD, and a fixed choice would load that SIMD alone.
    // (fq, the wave's place in the staging order, is wave-uniform: a wave none of whose lanes has a record to build -- three of the four
    // when the bin holds fewer than 64 -- branches over the whole staging step on the scalar unit instead of walking it with no lane active)
    const uint32_t fq = (uint32_t)__builtin_amdgcn_readfirstlane((int)(((tid >> 6) + ((tx + ty) & (uint32_t)(WPT - 1))) & (uint32_t)(WPT - 1)));
    const uint32_t ftid = fq * 64u + lane;
    for (uint32_t base0 = 0; base0 < n_total; base0 += (uint32_t)CHUNK * (TEAMS > 1 ? nteams : 1u)) {
        const uint32_t base = base0 + team * (uint32_t)CHUNK;
        // Two staging counters used alternately: the one of this pass was zeroed during the previous pass (or at kernel
        // entry), the other one is re-armed here, behind the barrier that every wave reaches only after it has read that
        // counter as the previous pass's record count.  (A single counter zeroed in front of the barrier could be cleared
        // under a wave that an instruction-cache miss held up between the previous barrier and that read.)
        __syncthreads();
        uint32_t* const cnt = lds_count + flip;
        if (tid == 0) lds_count[flip ^ 1u] = 0;
        // opaque copies: what make_tile_rec derives from the tile coordinates is rebuilt per chunk (a few instructions)
        // instead of being hoisted out of the loops into VGPRs that then spill
        uint32_t txl = tx, tyl = ty;
        asm volatile("" : "+s"(txl), "+s"(tyl));
        const uint32_t i = base + ftid;
        bool hit = false;
        uint4 rec[4]; uint32_t box = 0;
        const bool wave_stages = fq * 64u < (uint32_t)CHUNK && base + fq * 64u < n_total;      // (scalar)
        if (wave_stages) {
        if (MIRHI_STAGE_PRIO) __builtin_amdgcn_s_setprio(MIRHI_STAGE_PRIO);      // (the tile's other waves wait at the barrier for this one)
        if (ftid < (uint32_t)CHUNK && i < n_total) {
            TileTri T;
            if (BINS) {
                // flat index -> (list, slot) -> pool page; both words of the record are requested together
                uint32_t k = 0, j = i;
                if ((TEAMS > 1 || WPT > 4) && seg) {
#pragma unroll
                    for (uint32_t q = 1; q < 8u; q++) k += i >= seg[q] ? 1u : 0u;
                    j = i - seg[k];
                }
                const uint32_t page = j < fixed_recs ? (tile * fixed_recs + j) >> BIN_PAGE_LOG2 : pages[k * 8u + (j >> BIN_PAGE_LOG2)];
                hit = page < PAGE_NONE;            // (a page the exhausted pool could not supply: its records are in the big list)
                if (hit) {
                    const uint32_t ri = (page * (uint32_t)BIN_PAGE_RECS + (j & (BIN_PAGE_RECS - 1u))) * 2u;      // (pools stay far below 2^32 words)
                    const uint4 w0 = list[ri], w1 = list[ri + 1u];
                    tile_tri_from_bin(T, w0, w1);
                }
            } else {
                // all three words are requested together: one memory round trip, not two
                const uint4 w0 = list[(size_t)i * 3u], w1 = list[(size_t)i * 3u + 1u], w2 = list[(size_t)i * 3u + 2u];
                const int32_t tpx0 = (int32_t)txl * TILE, tpy0 = (int32_t)tyl * TILE;
                const int32_t minx = (int32_t)(w2.z & 0x7FFFu), maxx = (int32_t)((w2.z >> 16) & 0x7FFFu);
                const int32_t miny = (int32_t)(w2.w & 0xFFFFu), maxy = (int32_t)(w2.w >> 16);
                hit = !(maxx < tpx0 || minx > tpx0 + TILE - 1 || maxy < tpy0 || miny > tpy0 + TILE - 1);
                if (hit) hit = make_tile_rec(rec, box, w0, w1, w2, (int32_t)txl, (int32_t)tyl);
            }
            if (BINS && hit) hit = make_tile_rec(rec, box, T);
        }
        bool small = false, boxed = false;
        if (hit) {
            // small (and all scissor-cut) records are resolved right here, triangle-parallel; the rest is staged
            const uint32_t bw = ((box >> 8) & 0xFF) - (box & 0xFF) + 1u, bh = (box >> 24) - ((box >> 16) & 0xFF) + 1u;
            small = bw * bh <= P.tp_max_area;
            boxed = (rec[3].w & 0x80000000u) != 0u;
        }
        // Alpha-masked scope (MASKED variants: PROGS = 4 with the triangle-parallel path): records of draws whose fragments are
        // alpha-tested one by one leave here, whatever their size.  Waterfall over the draws present in the wave, so the draw
        // descriptor (viewport, texture, material) stays wave-uniform.
        if (MASKED && P.alpha_scope) {
            uint32_t prim = 0u, mydraw = 0xFFFFFFFFu;
            if (hit) {
                prim = P.idflip ? (MAX_PRIM_ID - rec[3].z) : rec[3].z;
                mydraw = P.num_draws > 1 ? find_draw(P, prim) : 0u;
            }
            uint64_t todo = __ballot(hit);
            while (todo) {
                const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)mydraw, __ffsll((long long)todo) - 1);
                const bool mine = hit && mydraw == d;
                DrawRef D = const_draws(P.draws)[__builtin_amdgcn_readfirstlane((int)d)];
                if (draw_needs_alpha_test(D)) {                      // (wave-uniform)
                    // small pixel boxes (and scissor-cut records, which need the box): the lane that built the record walks it;
                    // larger ones are staged with the masked bit and resolved pixel-parallel by raster_record_masked
                    const uint32_t mbw = ((box >> 8) & 0xFF) - (box & 0xFF) + 1u, mbh = (box >> 24) - ((box >> 16) & 0xFF) + 1u;
                    const bool walk = mbw * mbh <= 64u || (rec[3].w & 0x80000000u) != 0u;
                    if (mine && walk) {
                        raster_small_masked<KEYED>(D, prim, rec, box, lds_key, P, txl, tyl);
                        hit = false;
                    }
                    if (mine && !walk) { rec[3].w |= MIRHI_REC_MASKED; small = false; }
                }
                todo &= ~__ballot(mine);
            }
        }
        // Triangle-parallel only pays when the wave holds enough small records to keep its lanes busy (meshes of small
        // triangles); a few stragglers in a sparse tile would serialise their pixel loops while 3 waves wait.
        if (TP) {
            const bool wave_tp = __popcll(__ballot(hit && small)) >= TP_MIN_LANES;
            if (hit && (boxed || (small && wave_tp))) {
                raster_small<KEYED>(rec, box, lds_key, P);
                hit = false;
            }
        }
        const uint64_t ball = __ballot(hit);
        uint32_t wbase = 0;
        if (lane == 0 && ball) wbase = atomicAdd(cnt, (uint32_t)__popcll(ball));
        wbase = __builtin_amdgcn_readfirstlane(wbase);
        if (hit) {
            const uint32_t slot = wbase + (uint32_t)__popcll(ball & ((1ull << lane) - 1ull));
            lds_rec[slot * 4u + 0] = rec[0]; lds_rec[slot * 4u + 1] = rec[1];
            lds_rec[slot * 4u + 2] = rec[2]; lds_rec[slot * 4u + 3] = rec[3];
            if (!TP) lds_box[slot] = box;
        }
        if (MIRHI_STAGE_PRIO) __builtin_amdgcn_s_setprio(0);
        }   // wave_stages
        __syncthreads();
        const uint32_t n = *cnt;
        flip ^= 1u;
        if (base0 == 0) { STAMP(5); STAGE_END(2u); }
        if (n) raster_chunk<KEYED, TP, MASKED, 16 / WPT>(lds_rec, lds_box, n, qmask, ix0, iy0, fix0, fiy0, P, st, qbit0, lane, txl, tyl);
    }
}

// PROGS: bit 0 = pass contains TRIANGLE-program draws, bit 1 = MODEL / MODEL_FULL draws; 4 = any mix that
// includes MODEL_PBR draws or mip-mapped textures (its own variant so that the Cook-Torrance and trilinear code costs the
// other variants no registers)
// TP: 1 = the triangle-parallel path (LDS key array) is compiled in; the host enables it for scopes with many
//     triangles per tile, sparse scopes use the leaner pixel-parallel-only variant
// TEAMS = 2 (mesh variants of scopes whose triangles sit in a small part of the frame, PassParams::raster_teams): two sets
// of four waves per tile.  A real mesh is bound by its fullest tiles -- one workgroup per tile is one serial chain of
// staging passes while most of the chip idles -- and the teams split that chain: alternate chunks of the bin, keys merged
// through LDS at the end, then each team shades half of the tile's pixels.  No traffic through memory and no extra
// workgroups, unlike sharing a tile between workgroups ("measured and not adopted").  On frames that already
// fill the chip the wider workgroups only cost occupancy (C4 raster 109 -> 156 us, C5 202 -> 272 us), hence the host's choice.
// MASKEDV: the variant of an alpha-masked scope (PassParams::alpha_scope; PROGS = 4 with TP): carries raster_small_masked and
// raster_record_masked -- its own variant so that the scopes without masked materials keep their registers (the two paths cost
// the PROGS = 4 kernels 48-78 scalar registers spilled to vector lanes and 2 % of the dancer asset's frame).
// WIDE (mesh variants, one team): SIXTEEN waves per tile, one 8x8 block each, 1024 records staged per pass.  A mesh that covers a part of
// the frame (the 70k-triangle sphere: 419 of 2040 tiles hold everything) leaves three quarters of the SIMDs without a wave while every
// busy tile's four waves walk their four blocks one after the other -- 12.5 us of shading per wave, tools/stamps.py c3 -- at the issue rate
// of a wave that has its SIMD to itself.  Splitting the tile's PIXELS over four times the waves needs no merge and no atomics (unlike the
// two-team variant, which splits the records): same staging, same triangle-parallel walk, each wave visits the records that touch its
// block and shades its 64 pixels.  Chosen by the host from the number of busy tiles the previous frame reported (PassParams::raster_wide).
template <int PROGS, int KEYED, int TP, int TEAMS = 1, bool MASKEDV = false, int WPT = 4>
#ifndef MIRHI_PROGS2_WAVES
#define MIRHI_PROGS2_WAVES 5
#endif
__device__ __forceinline__ void raster_body(const PassParams* __restrict__ params, const RasterHead& H) {
    ParamsRef P = *(ParamsPtr)(uintptr_t)params;
    constexpr bool WIDE = WPT > 4;                      // WPT: waves per tile (and team): 4, or 8 / 16 in the wide variants
    static_assert(WPT == 4 || WPT == 8 || WPT == 16, "waves per tile");
    static_assert(!WIDE || (TEAMS == 1 && TP != 0 && PROGS >= 2), "the wide variants exist for one-team mesh scopes with the triangle-parallel path");
    constexpr int NB = 16 / WPT;                        // 8x8 blocks per wave
    constexpr uint32_t TT = (uint32_t)WPT * 64u;        // lanes per team
    // mesh variants stage with all four waves: their small records are resolved while staging (triangle-parallel), so a
    // hot tile's serial chain is one pass per CHUNK records; the sparse variants keep 192 (LDS per workgroup bounds
    // their 7-8 workgroups per CU)
    constexpr int CHUNK = WIDE ? (int)TT : ((TP && PROGS >= 2) ? RASTER_THREADS : RASTER_CHUNK);
    __shared__ uint4 lds_rec[TEAMS][CHUNK * 4];
    __shared__ unsigned long long lds_key[TP ? TILE * TILE : 1];   // depth keys written by the triangle-parallel path
    __shared__ uint32_t lds_box[TP ? 1 : RASTER_CHUNK];
    __shared__ uint32_t lds_count[TEAMS * 2];
    const uint32_t tid = threadIdx.x & (TT - 1u), lane = tid & 63u;       // tid: index within the team
    const uint32_t team = TEAMS > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 8) : 0u;
    const uint32_t q = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave index within the team: uniform, keep it in SGPRs
    // (one contiguous band of tiles per XCD measured 20-30 % slower on unevenly covered frames: runs stay interleaved)
    // Plain order: a 2-D grid, (blockIdx.x, blockIdx.y) = (tile column, tile row of the band): no division.
    // P.xcd_swizzle > 1 (1-D grid): workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 shares an XCD, each XCD
    // has its own L2); runs of G consecutive tiles go to the same XCD so neighbouring tiles hit the same L2.
    uint32_t tx = blockIdx.x, tyr = blockIdx.y;
    if (gridDim.y == 1u && P.xcd_swizzle > 1u) {
        const uint32_t G = P.xcd_swizzle, ntiles = gridDim.x;
        uint32_t t = blockIdx.x;
        if (ntiles % (8u * G) == 0u) {
            const uint32_t xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
            t = ((j / G) * 8u + xcd) * G + (j % G);
        }
        tx = t % H.tiles_x; tyr = t / H.tiles_x;
    }
    const uint32_t tile = tyr * H.tiles_x + tx, ty = H.tile_row_begin + tyr * H.tile_row_step;      // (grid row = owned row: PassParams::tile_row_step)
    // four waves: wave q owns the 16x16 quadrant q (blocks b: bx = b & 1, by = b >> 1 from its first); eight: the 16x8 strip
    // (q & 1, q >> 1), two blocks side by side; sixteen: block q (bx = q & 3, by = q >> 2)
    const int32_t ix0 = WPT == 16 ? (int32_t)((q & 3u) * 8u + (lane & 7u)) : (int32_t)((q & 1u) * 16u + (lane & 7u));
    const int32_t iy0 = WPT == 16 ? (int32_t)((q >> 2) * 8u + (lane >> 3)) : (WPT == 8 ? (int32_t)((q >> 1) * 8u + (lane >> 3)) : (int32_t)((q >> 1) * 16u + (lane >> 3)));
    const float fix0 = (float)ix0, fiy0 = (float)iy0;
    // the record's block mask has bit by * 4 + bx for block (bx, by) of the tile: the wave's first block, and all of its blocks
    const uint32_t qbit0 = WPT == 16 ? (1u << q) : (WPT == 8 ? (1u << ((q >> 1) * 4u + (q & 1u) * 2u)) : (1u << ((q >> 1) * 8u + (q & 1u) * 2u)));
    const uint32_t qmask = WPT == 16 ? qbit0 : (WPT == 8 ? qbit0 * 0x3u : qbit0 * 0x33u);

    STAMP(0);
    // both counters are fetched up front so their latencies overlap
    // (the head of the parameters comes by value: the counter loads depend on the kernarg load alone, not on a second hop)
    __shared__ uint32_t lds_seg[(TEAMS > 1 || WIDE) ? 9 : 1];     // two-team and wide variants: records in the per-XCD sub-bins before k; [8] = all
    const bool xcd_bins = (TEAMS > 1 || WIDE) && H.count_stride != 0u;
    uint32_t count_raw = H.bin_count[tile * BIN_COUNT_STRIDE];
    const uint32_t nbig_raw = *H.big_count;
    if (TEAMS > 1 || WIDE) {
        if (xcd_bins) {
            if (threadIdx.x == 0) {
                uint32_t c[8];
#pragma unroll
                for (uint32_t k = 0; k < 8u; k++) c[k] = H.bin_count[(k * H.count_stride + tile) * BIN_COUNT_STRIDE];
                uint32_t acc = 0;
#pragma unroll
                for (uint32_t k = 0; k < 8u; k++) { lds_seg[k] = acc; acc += c[k] < H.sub_cap ? c[k] : H.sub_cap; }
                lds_seg[8] = acc;
            }
            __syncthreads();
            count_raw = lds_seg[8];
        }
    }
    const uint32_t count = count_raw < H.bin_cap ? count_raw : H.bin_cap;
    const uint32_t nbig = nbig_raw < H.big_cap ? nbig_raw : H.big_cap;
    // The tile's row of the page table is needed only beyond the tile's fixed pages (there are none with per-XCD lists).
    __shared__ uint32_t lds_pages[BIN_TABLE_ROW];
    const bool need_pages = count > H.fixed_recs;
    if (need_pages && threadIdx.x < (uint32_t)BIN_TABLE_ROW) lds_pages[threadIdx.x] = P.bin_table[tile * (uint32_t)BIN_TABLE_ROW + threadIdx.x];
    // (Letting the second team leave tiles whose lists fit one staging pass was measured: the scopes that get this variant
    // leave most of the chip idle anyway, and those tiles then lose the split resolve: dancer 43 -> 46 us, 49 -> 57 us textured.)
    constexpr bool solo = TEAMS == 1;
    constexpr uint32_t nteams = (uint32_t)TEAMS;
    // The big-list counters are re-armed right away (no workgroup reads the other parity's counter, and the next scope
    // that uses this workspace is ordered behind this kernel), so nbig_raw need not stay live across the raster loops.
    // The tile's own bin counter is re-armed after the bin pass: every wave of this workgroup reads it above, and the
    // barriers of that pass order those reads before the store.
    if (tid == 0 && team == 0 && tile == 0) {
        *P.big_count_next = 0;                              // the next scope on this workspace appends to the other counter
        P.status[1] = nbig_raw;
        uint32_t pages_used = 0;                            // dynamic bin pages of this scope; no raster workgroup reads the counters,
        for (uint32_t x = 0; x < 8u; x++) {                 // and the next geometry kernel on this workspace is ordered behind this kernel
            const uint32_t c = P.pool_next[x * (uint32_t)POOL_COUNTER_STRIDE];
            pages_used += c < P.pool_dyn_pages ? c : P.pool_dyn_pages;
            P.pool_next[x * (uint32_t)POOL_COUNTER_STRIDE] = 0;
        }
        P.status[2] = pages_used;
        uint32_t busy = 0;                                  // busy tiles of the previous scope on this workspace (its counters are final)
        for (uint32_t x = 0; x < 8u; x++) { busy += P.active_prev[x * 32u]; P.active_prev[x * 32u] = 0; }
        P.status[3] = busy | 0x80000000u;
    }

    if (count == 0u && nbig == 0u) {
        // Empty tile (most tiles of a frame that shows one mesh): nothing to raster, nothing to shade -- every pixel takes what
        // the general resolve below gives a pixel no primitive reached: the clear colour / NO_PRIM unless the colour is
        // loaded, the clear depth if depth is stored and was not loaded (a loaded depth would be written back unchanged).
        // A fifth of the general path's instructions, and the workgroup's slots are free again a few microseconds sooner.
        if (TEAMS > 1 && team != 0u) return;
        const bool write_color = !P.color_load, write_depth = P.depth && P.depth_store && !P.depth_load;
        if (write_color || write_depth) {
            const uint32_t px0 = tx * TILE + (uint32_t)ix0, py0 = ty * TILE + (uint32_t)iy0;
#pragma unroll
            for (int b = 0; b < NB; b++) {
                const uint32_t px = px0 + (uint32_t)(b & 1) * BLOCK, py = py0 + (uint32_t)(b >> 1) * BLOCK;
                if (px >= P.width || py >= P.height) continue;
                const size_t pix = (size_t)py * P.width + px;
                if (write_color) {
                    if (P.color_format == 2) reinterpret_cast<float4*>(P.color)[pix] = make_float4(P.clear_color[0], P.clear_color[1], P.clear_color[2], P.clear_color[3]);
                    else reinterpret_cast<uint32_t*>(P.color)[pix] = P.clear_packed;         // (plain: four 32-byte segments per lane pair up in the L2)
                    if (P.prim_out) P.prim_out[pix] = NO_PRIM;
                }
                if (write_depth) P.depth[pix] = __uint_as_float(P.clear_depth_bits);
            }
        }
        STAMP(4);
        return;
    }

    if (tid == 0) { lds_count[2u * team] = 0; lds_count[2u * team + 1u] = 0; }      // this team's staging counters (ordered by raster_list's first barrier)
    if (PROGS >= 2 && tid == 0 && team == 0) __hip_atomic_fetch_add(&P.active[(tile & 7u) * 32u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (a busy tile; mesh scopes only)
    uint32_t flip = 0;
    if (TP) for (uint32_t e = tid + team * TT; e < TILE * TILE; e += TT * nteams) lds_key[e] = ~0ull;
    PixelState st;
#pragma unroll
    for (int b = 0; b < 4; b++) { st.zk[b] = P.init_zk; st.idk[b] = P.init_idk; }
    if (TEAMS > 1 && team != 0u) {
        // The other teams start from the WORST key, one that loses every merge: team 0 alone carries the pixel's real
        // starting state, and with a loaded depth that state may be worse than init_zk (a GREATER segment behind a LESS
        // one: the "cleared" key would beat every loaded key and resolve to a primitive that does not exist).  Plain keys
        // keep the top bit clear for the miss-in-the-key trick of raster_record.
#pragma unroll
        for (int b = 0; b < 4; b++) { st.zk[b] = KEYED == 0 ? 0x7FFFFFFFu : 0xFFFFFFFFu; st.idk[b] = 0xFFFFFFFFu; }
    }
    if (P.depth_load && P.depth && team == 0) { // second scope on a kept depth buffer: keys start from the stored depth (one team's keys)
        const uint32_t px0 = tx * TILE + (uint32_t)ix0, py0 = ty * TILE + (uint32_t)iy0;
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const uint32_t px = px0 + (uint32_t)(b & 1) * BLOCK, py = py0 + (uint32_t)(b >> 1) * BLOCK;
            uint32_t zo;
            init_key(P, px, py, px < P.width && py < P.height, st.zk[b], st.idk[b], zo);
        }
    }

    STAMP(1);
    STAGE_END(1u);
    // the tile's bin, then the list every tile tests (large / clipped / spilled triangles): one copy of the code
    // the tile's bin (32-byte records in pool pages), then the list every tile tests (48-byte records: large / clipped / scissor-cut /
    // spilled triangles).  Two instantiations of raster_list: one body fed from either source has to hold both record forms in
    // registers on the way to the tile record, which the TRIANGLE-only variants (64 VGPRs) pay with ~20 spills.
    constexpr bool MASKED = MASKEDV && PROGS == 4 && TP != 0;
    if (count) raster_list<KEYED, TP, CHUNK, TEAMS, true, MASKED, WPT>(reinterpret_cast<const uint4*>(H.bin_pool), count, tile, H.fixed_recs, lds_pages, xcd_bins ? lds_seg : nullptr, lds_rec[team], lds_box, lds_count + 2u * team, flip, team, nteams, lds_key, tx, ty, qmask, ix0, iy0, fix0, fiy0, P, st,
                                                         qbit0, tid, lane);
    STAMP(2);
    if (count && tid == 0 && team == 0) {                        // ready for the next scope that uses this workspace
        H.bin_count[tile * BIN_COUNT_STRIDE] = 0;
        if (xcd_bins) for (uint32_t k = 1; k < 8u; k++) H.bin_count[(k * H.count_stride + tile) * BIN_COUNT_STRIDE] = 0;
    }
    // (the row was copied to LDS before the first barrier of the bin pass; the next geometry kernel comes behind this kernel)
    if (need_pages && threadIdx.x < (uint32_t)BIN_TABLE_ROW) launder_params((ParamsPtr)(uintptr_t)params)->bin_table[tile * (uint32_t)BIN_TABLE_ROW + threadIdx.x] = PAGE_EMPTY;
    // parameters of this phase are (re)read here, see launder_params
    if (nbig) raster_list<KEYED, TP, CHUNK, TEAMS, false, MASKED, WPT>(reinterpret_cast<const uint4*>(launder_params((ParamsPtr)(uintptr_t)params)->big_recs), nbig, tile, 0u, lds_pages, nullptr, lds_rec[team], lds_box, lds_count + 2u * team, flip, team, nteams, lds_key, tx, ty, qmask, ix0, iy0, fix0, fiy0, P, st,
                                                         qbit0, tid, lane);

    STAMP(3);
    STAGE_END(3u);
    if (TP || TEAMS > 1) __syncthreads();     // every triangle-parallel ds_min of this tile has landed; staging areas are free
    // ---- resolve: shade the winning primitive of each pixel, store once ---------------------------
    // merge the pixel-parallel (registers) and triangle-parallel (LDS) results: smaller key wins
    if (TP) {
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const unsigned long long kreg = ((unsigned long long)st.zk[b] << 32) | st.idk[b];
            const unsigned long long klds = lds_key[(iy0 + (b >> 1) * BLOCK) * TILE + ix0 + (b & 1) * BLOCK];
            const unsigned long long kmin = klds < kreg ? klds : kreg;
            st.zk[b] = (uint32_t)(kmin >> 32); st.idk[b] = (uint32_t)kmin;
        }
    }
    if (TEAMS > 1 && !solo) {
        // merge the teams' keys through their (now idle) staging areas: teams > 0 publish, team 0 takes the minimum and
        // publishes the result, everyone reads it back; then team t shades blocks [4t/TEAMS, 4(t+1)/TEAMS) of each lane
        unsigned long long* const mine = reinterpret_cast<unsigned long long*>(&lds_rec[team][0]);
        if (team != 0u) {
#pragma unroll
            for (int b = 0; b < 4; b++) mine[b * RASTER_THREADS + tid] = ((unsigned long long)st.zk[b] << 32) | st.idk[b];
        }
        __syncthreads();
        if (team == 0u) {
#pragma unroll
            for (int b = 0; b < 4; b++) {
                unsigned long long k = ((unsigned long long)st.zk[b] << 32) | st.idk[b];
#pragma unroll
                for (int t = 1; t < TEAMS; t++) {
                    const unsigned long long o = reinterpret_cast<const unsigned long long*>(&lds_rec[t][0])[b * RASTER_THREADS + tid];
                    k = o < k ? o : k;
                }
                mine[b * RASTER_THREADS + tid] = k;
                st.zk[b] = (uint32_t)(k >> 32); st.idk[b] = (uint32_t)k;
            }
        }
        __syncthreads();
        if (team != 0u) {
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const unsigned long long k = reinterpret_cast<const unsigned long long*>(&lds_rec[0][0])[b * RASTER_THREADS + tid];
                st.zk[b] = (uint32_t)(k >> 32); st.idk[b] = (uint32_t)k;
            }
        }
    }
    // The resolve reads its parameters through a laundered kernarg pointer: the scalar loads are issued here, not at
    // kernel entry, so their registers are not live across the raster loops (which otherwise spill SGPRs to VGPR lanes).
    const ParamsPtr R = launder_params((ParamsPtr)(uintptr_t)params);
    const uint32_t px0 = tx * TILE + (uint32_t)ix0, py0 = ty * TILE + (uint32_t)iy0;
    // flat colours of all four owned pixels are requested before the first one is used (four overlapping loads
    // instead of four dependent round trips in the loop below).  A pixel is covered iff its id key moved off the
    // initial one (no primitive carries NO_PRIM, and the "nothing can pass" state (0, 0) is never replaced).
    // Addressing is a uniform base plus a 32-bit byte offset per lane (tables and targets stay far below 4 GB).
    uint32_t flat4[4] = {0u, 0u, 0u, 0u};
    const uint8_t* flat_color = PROGS == 1 && !R->depth_load ? reinterpret_cast<const uint8_t*>(R->flat_color) : nullptr;
    const uint32_t init_idk = R->init_idk;
    if (flat_color) {
        if (R->idflip) {
#pragma unroll
            for (int b = 0; b < 4; b++)
                if (st.idk[b] != init_idk) flat4[b] = *reinterpret_cast<const uint32_t*>(flat_color + ((MAX_PRIM_ID - st.idk[b]) << 2));
        } else {
#pragma unroll
            for (int b = 0; b < 4; b++)
                if (st.idk[b] != init_idk) flat4[b] = *reinterpret_cast<const uint32_t*>(flat_color + (st.idk[b] << 2));
        }
    }
    // Fast exit for the headline shape of work: every covered pixel of this wave belongs to a flat-coloured
    // triangle whose packed colour the geometry kernel already produced, and only the 8-bit colour target is
    // written.  Same values as the general loop below, a fraction of its instructions.
    if (PROGS == 1 && R->resolve_flat_only) {            // (= flat_color && 8-bit target && no primitive-id image && no depth store, computed by the host)
        // a covered pixel without a flat colour has to be shaded: then the whole wave takes the general loop
        const bool need_shade = (st.idk[0] != init_idk && flat4[0] == 0u) || (st.idk[1] != init_idk && flat4[1] == 0u) ||
                                (st.idk[2] != init_idk && flat4[2] == 0u) || (st.idk[3] != init_idk && flat4[3] == 0u);
        if (__ballot(need_shade) == 0ull) {
            const uint32_t width = R->width, height = R->height, clear_packed = R->clear_packed;
            uint8_t* row0 = reinterpret_cast<uint8_t*>(R->color);
            uint8_t* row1 = row0 + (size_t)BLOCK * width * 4u;                 // the lower pair of blocks: uniform base
            const uint32_t off = (py0 * width + px0) * 4u;
            if ((tx + 1u) * TILE <= width && (ty + 1u) * TILE <= height && !R->color_load) {   // wave-uniform: interior tile
                store_target(reinterpret_cast<uint32_t*>(row0 + off), st.idk[0] != init_idk ? flat4[0] : clear_packed);
                store_target(reinterpret_cast<uint32_t*>(row0 + off + 4u * BLOCK), st.idk[1] != init_idk ? flat4[1] : clear_packed);
                store_target(reinterpret_cast<uint32_t*>(row1 + off), st.idk[2] != init_idk ? flat4[2] : clear_packed);
                store_target(reinterpret_cast<uint32_t*>(row1 + off + 4u * BLOCK), st.idk[3] != init_idk ? flat4[3] : clear_packed);
            } else {
                const uint32_t color_load = R->color_load;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const uint32_t px = px0 + (uint32_t)(b & 1) * BLOCK, py = py0 + (uint32_t)(b >> 1) * BLOCK;
                    const bool won = st.idk[b] != init_idk;
                    if (px < width && py < height && (won || !color_load))
                        store_target(reinterpret_cast<uint32_t*>((b >> 1 ? row1 : row0) + off + 4u * BLOCK * (uint32_t)(b & 1)), won ? flat4[b] : clear_packed);
                }
            }
            STAMP(4);
            return;
        }
    }
    uint32_t held = 0;                 // packed colour of the pair's left (even) block, until the right one is shaded
    bool held_all = false;             //   ... and every lane of the wave writes it (no pixel outside the target, none kept by LOAD)
#pragma unroll 1
    for (int b = 0; b < NB; b++) {
        if (TEAMS > 1 && !solo && (uint32_t)(b * TEAMS) / 4u != team) continue;      // another team shades this block
        const uint32_t px = px0 + (uint32_t)(b & 1) * BLOCK, py = py0 + (uint32_t)(b >> 1) * BLOCK;
        const bool inb = px < P.width && py < P.height;
        uint32_t izk, iidk, zorig;
        init_key(P, px, py, inb, izk, iidk, zorig);
        const uint32_t zkb = b == 0 ? st.zk[0] : (b == 1 ? st.zk[1] : (b == 2 ? st.zk[2] : st.zk[3]));
        const uint32_t idb = b == 0 ? st.idk[0] : (b == 1 ? st.idk[1] : (b == 2 ? st.idk[2] : st.idk[3]));
        const bool none = !inb || ((zkb == izk) && (idb == iidk));
        const size_t pix = (size_t)py * P.width + px;
        const uint32_t prim = none ? NO_PRIM : (P.idflip ? (MAX_PRIM_ID - idb) : idb);
        f4 col = {P.clear_color[0], P.clear_color[1], P.clear_color[2], P.clear_color[3]};
        // waterfall over the draws present in this wave: the draw descriptor stays wave-uniform (scalar loads)
        const uint32_t mydraw = none ? 0xFFFFFFFFu : (P.num_draws > 1 ? find_draw(P, prim) : 0u);
        uint32_t flat = 0;
        if (PROGS == 1 && P.flat_color && !none) {      // alpha is 255 whenever it is set
            if (P.depth_load) flat = P.flat_color[prim];
            else flat = b == 0 ? flat4[0] : (b == 1 ? flat4[1] : (b == 2 ? flat4[2] : flat4[3]));
        }
        uint64_t todo = __ballot(!none && flat == 0u);
        while (todo) {
            const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)mydraw, __ffsll((long long)todo) - 1);
            const bool mine = mydraw == d && flat == 0u;
            if (mine) {
                // readfirstlane again: inside this branch the compiler knows mydraw == d and would otherwise
                // substitute the per-lane value, turning every descriptor access into a vector load
                DrawRef D = const_draws(P.draws)[__builtin_amdgcn_readfirstlane(mydraw)];
                const uint32_t tri = prim - D.prim_base;
                const float pxc = (float)px + 0.5f, pyc = (float)py + 0.5f;
                if (PROGS == 1) col = shade_triangle_program(D, tri, pxc, pyc);
                else if (PROGS == 2) col = shade_model_program<false>(D, tri, pxc, pyc);
                else col = (D.program == 0) ? shade_triangle_program(D, tri, pxc, pyc) : shade_model_program<PROGS == 4>(D, tri, pxc, pyc);
            }
            todo &= ~__ballot(mine);
        }
        if (P.color_format != 2 && NB >= 2) {
            // 8-bit target: the wave's two side-by-side blocks leave as one 8-byte store per lane (store_pair) when every lane of both writes its pixel
            // (an interior tile without LOAD: the usual case); otherwise pixel by pixel, plain stores
            const bool wr = inb && !(none && P.color_load);
            const uint32_t packed = none ? P.clear_packed : (flat ? flat : pack_bgra8_srgb(col));
            const bool all = __ballot(wr) == ~0ull;
            if ((b & 1) == 0) { held = packed; held_all = all; if (!all && wr) reinterpret_cast<uint32_t*>(P.color)[pix] = packed; }
            else if (all && held_all) store_pair(reinterpret_cast<uint32_t*>(P.color), py * P.width + (px - (uint32_t)BLOCK) - (lane & 7u), held, packed, lane);
            else {
                if (held_all) reinterpret_cast<uint32_t*>(P.color)[pix - (size_t)BLOCK] = held;      // (the left block was complete, this one is not)
                if (wr) reinterpret_cast<uint32_t*>(P.color)[pix] = packed;
            }
        }
        if (!inb) continue;
        if (!(none && P.color_load)) {
            if (P.color_format == 2) reinterpret_cast<float4*>(P.color)[pix] = make_float4(col.x, col.y, col.z, col.w);
            else if (NB < 2) reinterpret_cast<uint32_t*>(P.color)[pix] = none ? P.clear_packed : (flat ? flat : pack_bgra8_srgb(col));      // (one block per wave: a lone 32-byte segment)
        }
        if (P.prim_out && !(none && P.color_load)) P.prim_out[pix] = prim;     // LOAD keeps what an earlier scope / segment wrote
        if (P.depth && P.depth_store) {
            const uint32_t zb = (none || !P.zmask) ? zorig : (zkb ^ P.zflip);
            P.depth[pix] = __uint_as_float(zb);
        }
    }
    STAMP(4);
}

#define MIRHI_RASTER_BOUNDS __launch_bounds__(RASTER_THREADS * TEAMS, (PROGS == 1 ? (TP ? 7 : 8) : (PROGS == 2 ? MIRHI_PROGS2_WAVES : 4)))
// one rendering scope: grid (tiles_x, tile rows of the band)
template <int PROGS, int KEYED, int TP, int TEAMS = 1, bool MASKEDV = false>
__global__ MIRHI_RASTER_BOUNDS void raster_kernel(const PassParams* __restrict__ params, const RasterHead H) {
    raster_body<PROGS, KEYED, TP, TEAMS, MASKEDV>(params, H);
}
// the wide mesh variant: sixteen waves per tile (raster_body, WIDE); one workgroup is a CU's four waves per SIMD
template <int PROGS, int KEYED, int WPT>
__global__ __launch_bounds__(WPT * 64, (PROGS == 2 && WPT == 16 ? 6 : (PROGS == 2 ? 5 : 4))) void raster_kernel_wide(const PassParams* __restrict__ params, const RasterHead H) {
    raster_body<PROGS, KEYED, 1, 1, false, WPT>(params, H);
}
// up to MAX_BATCH independent rendering scopes of equal shape (the frames of one mirhi_queue_submit): grid (tiles_x, tile rows, scopes).
// One launch instead of one per frame: the ramp-up and drain of a kernel (5 us of the 11 us an isolated 10k-triangle raster kernel
// takes) are paid once per batch, and the frames' tiles fill the chip back to back.  The per-scope arguments come by value in the
// kernarg segment, indexed by blockIdx.z (scalar loads).
template <int PROGS, int KEYED, int TP, int TEAMS = 1>
__global__ MIRHI_RASTER_BOUNDS void raster_kernel_batch(const RasterBatch B) {
    const uint32_t z = blockIdx.z;
    const RasterHead H = B.head[z];
    raster_body<PROGS, KEYED, TP, TEAMS>(B.params[z], H);
}

#endif  // MIRHI_RASTER_HIP_H
