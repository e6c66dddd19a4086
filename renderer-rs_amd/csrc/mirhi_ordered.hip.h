// mirhi_ordered.hip.h -- ordered_kernel: fragment-by-fragment resolve in primitive order (colour blending; depth states
// whose outcome depends on the order of all fragments).  ColorBlendAttachment pipeline.rs:478-531, BlendFactor :411-448,
// BlendOp :452-476; Vulkan 1.3 section 28.1 for the equations.
// Part of the single device translation unit mirhi_kernels.hip (included inside namespace mirhi).
#ifndef MIRHI_ORDERED_HIP_H
#define MIRHI_ORDERED_HIP_H

#pragma clang fp contract(off)

// Vulkan blend factor for RGB (x, y, z) and for alpha (w) from source s and destination d
__device__ __forceinline__ f4 blend_factor(uint32_t f, f4 s, f4 d) {
    const float omda = 1.0f - d.w, sat = s.w < omda ? s.w : omda;
    f4 r = {0.0f, 0.0f, 0.0f, 0.0f};                       // Zero, and the constant-colour factors the pipeline refuses
    if (f == 1u) r = {1.0f, 1.0f, 1.0f, 1.0f};
    else if (f == 2u) r = s;
    else if (f == 3u) r = {1.0f - s.x, 1.0f - s.y, 1.0f - s.z, 1.0f - s.w};
    else if (f == 4u) r = d;
    else if (f == 5u) r = {1.0f - d.x, 1.0f - d.y, 1.0f - d.z, 1.0f - d.w};
    else if (f == 6u) r = {s.w, s.w, s.w, s.w};
    else if (f == 7u) r = {1.0f - s.w, 1.0f - s.w, 1.0f - s.w, 1.0f - s.w};
    else if (f == 8u) r = {d.w, d.w, d.w, d.w};
    else if (f == 9u) r = {omda, omda, omda, omda};
    else if (f == 14u) r = {sat, sat, sat, 1.0f};
    return r;
}
__device__ __forceinline__ float blend_op(uint32_t op, float s, float sf, float d, float df) {
    float r = s * sf + d * df;
    if (op == 1u) r = s * sf - d * df;
    else if (op == 2u) r = d * df - s * sf;
    else if (op == 3u) r = s < d ? s : d;
    else if (op == 4u) r = s > d ? s : d;
    return r;
}
// dst <- blend(src, dst) under the segment's ColorBlendAttachment (P.blend[0] == 0: opaque overwrite under the write mask)
__device__ __forceinline__ f4 blend_pixel(ParamsRef P, f4 s, f4 d) {
    f4 r = s;
    if (P.blend[0]) {
        const f4 sc = blend_factor(P.blend[1], s, d), dc = blend_factor(P.blend[2], s, d);
        const float sa = blend_factor(P.blend[4], s, d).w, da = blend_factor(P.blend[5], s, d).w;
        r = {blend_op(P.blend[3], s.x, sc.x, d.x, dc.x), blend_op(P.blend[3], s.y, sc.y, d.y, dc.y),
             blend_op(P.blend[3], s.z, sc.z, d.z, dc.z), blend_op(P.blend[6], s.w, sa, d.w, da)};
    }
    const uint32_t wm = P.blend[0] ? P.blend[7] : 0xFu;
    return {(wm & 1u) ? r.x : d.x, (wm & 2u) ? r.y : d.y, (wm & 4u) ? r.z : d.z, (wm & 8u) ? r.w : d.w};
}
__device__ __forceinline__ bool depth_passes(uint32_t op, uint32_t z, uint32_t stored) {      // bits of clamped, non-negative floats
    const bool lt = z < stored, eq = z == stored;
    return (lt && (op & 1u)) || (eq && (op & 2u)) || (!lt && !eq && (op & 4u));     // the compare op's bits are {<, ==, >}
}

// Per owned pixel: stored depth bits, last writer, colour (6 words).  The state lives in LDS, [word][block][thread]: the
// block loop stays rolled (one copy of the fragment programs) and a rolled loop over register arrays would put them in
// scratch memory, which no kernel here may use.
#define ORD_STATE(word, b) lds_state[((word) * 4u + (uint32_t)(b)) * ORDERED_THREADS + tid]

// one tile record against the four blocks of this wave's quadrant, in place: coverage, depth test, fragment program, blend
template <int PROGS>
__device__ __forceinline__ void ordered_record(const RecRegs& r, uint32_t box, int32_t ix0, int32_t iy0, float fix0, float fiy0,
                                               uint32_t px0, uint32_t py0, ParamsRef P, uint32_t* lds_state, uint32_t tid, uint32_t qbit0) {
    const int32_t A0 = (int32_t)r.w0.w, A1 = (int32_t)r.w1.x, A2 = (int32_t)r.w1.y;
    const int32_t B0 = (int32_t)r.w1.z, B1 = (int32_t)r.w1.w, B2 = (int32_t)r.w2.x;
    const float z0 = __uint_as_float(r.w2.w), zx = __uint_as_float(r.w3.x), zy = __uint_as_float(r.w3.y);
    const uint32_t prim = __builtin_amdgcn_readfirstlane(r.w3.z);
    const uint32_t m = __builtin_amdgcn_readfirstlane(r.w3.w);
    const int32_t s0 = mad24(B0, iy0, mad24(A0, ix0, (int32_t)r.w0.x));
    const int32_t s1 = mad24(B1, iy0, mad24(A1, ix0, (int32_t)r.w0.y));
    const int32_t s2 = mad24(B2, iy0, mad24(A2, ix0, (int32_t)r.w0.z));
    const float dx0 = fix0 + __uint_as_float(r.w2.y), dy0 = fiy0 + __uint_as_float(r.w2.z);
    const bool boxed = (m & 0x80000000u) != 0u;
    DrawRef D = const_draws(P.draws)[P.num_draws > 1 ? find_draw(P, prim) : 0u];
    // every block of the record shades the same triangle: its vertex indices are fetched once (a uniform address), so a block's
    // fragment program starts at the vertex fetch, one dependent memory round trip later than the record, not two
    uint32_t vin[3];
    fetch_triangle_indices(D, prim - D.prim_base, vin);
#pragma unroll 1
    for (int b = 0; b < 4; b++) {
        const int bx = b & 1, by = b >> 1;
        if (!(m & (qbit0 << (by * 4 + bx)))) continue;
        const int32_t S0 = s0 + (A0 * bx + B0 * by) * BLOCK, S1 = s1 + (A1 * bx + B1 * by) * BLOCK, S2 = s2 + (A2 * bx + B2 * by) * BLOCK;
        bool inside = (S0 | S1 | S2) >= 0;
        if (boxed) {
            const int32_t ix = ix0 + bx * BLOCK, iy = iy0 + by * BLOCK;
            inside = inside && ix >= (int32_t)(box & 0xFF) && ix <= (int32_t)((box >> 8) & 0xFF) &&
                     iy >= (int32_t)((box >> 16) & 0xFF) && iy <= (int32_t)(box >> 24);
        }
        const float dx = dx0 + (float)(bx * BLOCK), dy = dy0 + (float)(by * BLOCK);
        const float z = __builtin_fmaf(dy, zy, __builtin_fmaf(dx, zx, z0));
        const uint32_t zk = __float_as_uint(__builtin_amdgcn_fmed3f(z, 0.0f, 1.0f)) & 0x7FFFFFFFu;
        const uint32_t dcur = ORD_STATE(0u, b);
        bool pass = inside && (!P.ord_depth_test || depth_passes(P.ord_depth_op, zk, dcur));
        // a second fragment of the primitive that owns the pixel (overlapping clip pieces) under Always with depth write: the nearer one is kept
        if (P.ord_depth_test && P.ord_depth_write && P.ord_depth_op == 7u && ORD_STATE(1u, b) == prim && !(zk < dcur)) pass = false;
        if (__ballot(pass) == 0ull) continue;
        const uint32_t px = px0 + (uint32_t)bx * BLOCK, py = py0 + (uint32_t)by * BLOCK;
        f4 src = {0.0f, 0.0f, 0.0f, 0.0f};
        if (pass) {
            const float pxc = (float)px + 0.5f, pyc = (float)py + 0.5f;
            if (PROGS == 1) src = shade_triangle_program(D, vin, pxc, pyc);
            else if (PROGS == 2) src = shade_model_program<false>(D, vin, pxc, pyc);
            else src = (D.program == 0) ? shade_triangle_program(D, vin, pxc, pyc) : shade_model_program<PROGS == 4>(D, vin, pxc, pyc);
        }
        // `if (baseColor.a < alphaCutoff) discard;` (pixel/model_pbr.hlsl:176-179; the program's alpha output is baseColor.a): a discarded
        // fragment writes neither colour nor depth nor its primitive id
        bool kept = pass;
        if (PROGS == 4 && D.program == 3 && src.w < ldcf(cb(D.material), 44)) kept = false;
        if (kept) {
            const f4 dcol = {__uint_as_float(ORD_STATE(2u, b)), __uint_as_float(ORD_STATE(3u, b)), __uint_as_float(ORD_STATE(4u, b)), __uint_as_float(ORD_STATE(5u, b))};
            const f4 out = blend_pixel(P, src, dcol);
            ORD_STATE(2u, b) = __float_as_uint(out.x); ORD_STATE(3u, b) = __float_as_uint(out.y);
            ORD_STATE(4u, b) = __float_as_uint(out.z); ORD_STATE(5u, b) = __float_as_uint(out.w);
            ORD_STATE(1u, b) = prim;
            if (P.ord_depth_test && P.ord_depth_write) ORD_STATE(0u, b) = zk;
        }
    }
}

// sRGB8 BGRA -> linear RGBA (exact table; the store side is pack_bgra8_srgb)
__device__ __forceinline__ f4 unpack_bgra8_srgb(uint32_t p) {
    return {g_srgb_lut[(p >> 16) & 0xFF], g_srgb_lut[(p >> 8) & 0xFF], g_srgb_lut[p & 0xFF], (float)(p >> 24) * (1.0f / 255.0f)};
}

// One 256-lane workgroup per 32x32 tile, wave q = quadrant q, as raster_kernel; the segment's triangles are walked in
// primitive order, 256 slots at a time: slots whose pixel box overlaps the tile are compacted IN ORDER into LDS as tile
// records (ballot + prefix inside a wave, wave totals through LDS), then every wave visits them one after the other.
// A slot marked ORDERED_MARKER stands for the pieces of a clipped triangle in the big list; a chunk that holds one is
// processed in sub-ranges so the pieces are visited at the triangle's position.
template <int PROGS>
__global__ __launch_bounds__(ORDERED_THREADS, 2) void ordered_kernel(const PassParams* __restrict__ params, const RasterHead H) {
    ParamsRef P = *(ParamsPtr)(uintptr_t)params;
    __shared__ uint4 lds_rec[ORDERED_THREADS * 4];
    __shared__ uint32_t lds_box[ORDERED_THREADS];
    __shared__ uint32_t lds_wave[4];
    __shared__ uint32_t lds_marker[ORDERED_THREADS / 32];     // bit per slot of the chunk: clipped-triangle marker
    __shared__ uint32_t lds_state[6 * 4 * ORDERED_THREADS];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t q = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t tx = blockIdx.x, tyr = blockIdx.y, ty = H.tile_row_begin + tyr * H.tile_row_step, tile = tyr * H.tiles_x + tx;
    const int32_t ix0 = (int32_t)((q & 1u) * 16u + (lane & 7u)), iy0 = (int32_t)((q >> 1) * 16u + (lane >> 3));
    const float fix0 = (float)ix0, fiy0 = (float)iy0;
    const uint32_t qbit0 = 1u << ((q >> 1) * 8u + (q & 1u) * 2u), qmask = qbit0 * 0x33u;
    const uint32_t px0 = tx * TILE + (uint32_t)ix0, py0 = ty * TILE + (uint32_t)iy0;
    const int32_t tpx0 = (int32_t)(tx * TILE), tpy0 = (int32_t)(ty * TILE);
    const uint32_t nbig_raw = *H.big_count;
    if (tid == 0 && tile == 0) { *P.big_count_next = 0; P.status[1] = nbig_raw; }   // (bins are not used by an ordered segment)

    // The pixel state is set up when the first record of the segment reaches this tile.  A tile no record reaches has nothing
    // to do at all if the segment keeps the colour (and depth) written so far -- the usual case for translucent geometry over
    // an opaque frame -- so it neither reads nor rewrites its 4 KB of the target.
    const bool may_skip = P.color_load && !(P.depth && P.depth_store && !P.depth_load);
    bool loaded = false;
    auto load_state = [&]() {
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const uint32_t px = px0 + (uint32_t)(b & 1) * BLOCK, py = py0 + (uint32_t)(b >> 1) * BLOCK;
            const bool inb = px < P.width && py < P.height;
            const size_t pix = (size_t)py * P.width + px;
            f4 c0 = {P.clear_color[0], P.clear_color[1], P.clear_color[2], P.clear_color[3]};
            if (P.color_load && inb) {
                if (P.color_format == 2) { const float4 c = reinterpret_cast<const float4*>(P.color)[pix]; c0 = {c.x, c.y, c.z, c.w}; }
                else c0 = unpack_bgra8_srgb(reinterpret_cast<const uint32_t*>(P.color)[pix]);
            }
            ORD_STATE(0u, b) = (P.depth_load && P.depth && inb) ? __float_as_uint(P.depth[pix]) : P.clear_depth_bits;
            ORD_STATE(1u, b) = NO_PRIM;
            ORD_STATE(2u, b) = __float_as_uint(c0.x); ORD_STATE(3u, b) = __float_as_uint(c0.y);
            ORD_STATE(4u, b) = __float_as_uint(c0.z); ORD_STATE(5u, b) = __float_as_uint(c0.w);
        }
        loaded = true;
    };
    if (!may_skip) load_state();
    const uint4* slots = reinterpret_cast<const uint4*>(P.ordered_recs);
    const uint4* pieces = reinterpret_cast<const uint4*>(P.big_recs);
    // the box word of the NEXT chunk's slot is requested while this chunk is handled: the scan over the segment's slots then
    // runs at the rate of its barriers, not of a memory round trip per chunk
    // (lanes beyond the segment's end read its last slot -- a valid address -- and ignore it)
    const uint32_t last = P.ordered_count ? P.ordered_count - 1u : 0u;
    uint4 w2_next = make_uint4(0u, 0u, 0u, 0u);
    if (P.ordered_count) w2_next = slots[(size_t)(tid < last ? tid : last) * 3u + 2u];
    for (uint32_t base = 0; base < P.ordered_count; base += ORDERED_THREADS) {
        // ---- classify this thread's slot ----
        const uint32_t i = base + tid;
        const uint4 w2 = w2_next;
        {
            const uint32_t nx = i + (uint32_t)ORDERED_THREADS;
            w2_next = slots[(size_t)(nx < last ? nx : last) * 3u + 2u];
        }
        const bool live = i < P.ordered_count;
        const bool marker = live && w2.w == ORDERED_MARKER;
        const int32_t minx = (int32_t)(w2.z & 0x7FFFu), maxx = (int32_t)((w2.z >> 16) & 0x7FFFu);
        const int32_t miny = (int32_t)(w2.w & 0xFFFFu), maxy = (int32_t)(w2.w >> 16);
        const bool overlap = live && !marker && !(maxx < minx) && !(maxx < tpx0 || minx > tpx0 + TILE - 1 || maxy < tpy0 || miny > tpy0 + TILE - 1);
        {
            const uint64_t mb = __ballot(marker);
            if (lane == 0) { lds_marker[q * 2u] = (uint32_t)mb; lds_marker[q * 2u + 1u] = (uint32_t)(mb >> 32); }
        }
        // Most chunks of a segment hold nothing for this tile (256 scattered 50-pixel triangles reach a given tile one time in five):
        // such a chunk costs its one 16-byte load per lane and this barrier, not the three staging barriers below.  (Nobody reads
        // lds_marker in that case, so the next chunk may overwrite it right away.)
        if (!__syncthreads_or((int)(overlap || marker))) continue;
        // ---- sub-ranges [lo, hi) of the chunk between markers ----
        uint32_t lo = 0;
        while (lo < (uint32_t)ORDERED_THREADS) {
            uint32_t hi = ORDERED_THREADS;           // first marker at or after lo (wave-uniform: read from LDS)
            for (uint32_t wv = lo >> 5; wv < (uint32_t)ORDERED_THREADS / 32u; wv++) {
                uint32_t bits = lds_marker[wv];
                if (wv == (lo >> 5)) bits &= ~0u << (lo & 31u);
                if (bits) { hi = wv * 32u + (uint32_t)__ffs((int)bits) - 1u; break; }
            }
            hi = __builtin_amdgcn_readfirstlane(hi);
            // records of the sub-range, compacted in slot order
            const bool take = overlap && tid >= lo && tid < hi;
            uint4 rec[4]; uint32_t box = 0;
            bool hit = false;
            if (take) {
                const uint4 w0 = slots[(size_t)i * 3u], w1 = slots[(size_t)i * 3u + 1u];
                hit = make_tile_rec(rec, box, w0, w1, w2, (int32_t)tx, (int32_t)ty);
            }
            const uint64_t ball = __ballot(hit);
            if (lane == 0) lds_wave[q] = (uint32_t)__popcll(ball);
            __syncthreads();
            uint32_t before = 0, n = 0;
#pragma unroll
            for (uint32_t wv = 0; wv < 4u; wv++) { const uint32_t c = lds_wave[wv]; before += wv < q ? c : 0u; n += c; }
            if (hit) {
                const uint32_t slot = before + (uint32_t)__popcll(ball & ((1ull << lane) - 1ull));
                lds_rec[slot * 4u + 0] = rec[0]; lds_rec[slot * 4u + 1] = rec[1]; lds_rec[slot * 4u + 2] = rec[2]; lds_rec[slot * 4u + 3] = rec[3];
                lds_box[slot] = box;
            }
            __syncthreads();
            if (n && !loaded) load_state();
            for (uint32_t j = 0; j < n; j++) {
                const uint32_t mj = lds_rec[j * 4u + 3u].w;
                if (mj & qmask) ordered_record<PROGS>(load_rec(lds_rec, j), lds_box[j], ix0, iy0, fix0, fiy0, px0, py0, P, lds_state, tid, qbit0);
            }
            __syncthreads();
            if (hi < (uint32_t)ORDERED_THREADS) {
                // the clipped triangle at slot hi: its pieces, in the big list, one per lane of wave 0 (at most 7)
                const uint4 mk = slots[(size_t)(base + hi) * 3u];
                const uint32_t first = __builtin_amdgcn_readfirstlane(mk.x), count = __builtin_amdgcn_readfirstlane(mk.y);
                bool phit = false;
                if (tid < count) {
                    const uint4 p0 = pieces[(size_t)(first + tid) * 3u], p1 = pieces[(size_t)(first + tid) * 3u + 1u], p2 = pieces[(size_t)(first + tid) * 3u + 2u];
                    const int32_t pminx = (int32_t)(p2.z & 0x7FFFu), pmaxx = (int32_t)((p2.z >> 16) & 0x7FFFu);
                    const int32_t pminy = (int32_t)(p2.w & 0xFFFFu), pmaxy = (int32_t)(p2.w >> 16);
                    if (!(pmaxx < tpx0 || pminx > tpx0 + TILE - 1 || pmaxy < tpy0 || pminy > tpy0 + TILE - 1))
                        phit = make_tile_rec(rec, box, p0, p1, p2, (int32_t)tx, (int32_t)ty);
                }
                const uint64_t pball = __ballot(phit);
                if (tid == 0) lds_wave[0] = (uint32_t)__popcll(pball);
                if (phit) {      // (count <= 7 < 64: all in wave 0)
                    const uint32_t slot = (uint32_t)__popcll(pball & ((1ull << lane) - 1ull));
                    lds_rec[slot * 4u + 0] = rec[0]; lds_rec[slot * 4u + 1] = rec[1]; lds_rec[slot * 4u + 2] = rec[2]; lds_rec[slot * 4u + 3] = rec[3];
                    lds_box[slot] = box;
                }
                __syncthreads();
                const uint32_t pn = lds_wave[0];
                if (pn && !loaded) load_state();
                for (uint32_t j = 0; j < pn; j++) {
                    const uint32_t mj = lds_rec[j * 4u + 3u].w;
                    if (mj & qmask) ordered_record<PROGS>(load_rec(lds_rec, j), lds_box[j], ix0, iy0, fix0, fiy0, px0, py0, P, lds_state, tid, qbit0);
                }
                __syncthreads();
            }
            lo = hi + 1u;
        }
    }
    // ---- store ----
    if (!loaded) return;
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const uint32_t px = px0 + (uint32_t)(b & 1) * BLOCK, py = py0 + (uint32_t)(b >> 1) * BLOCK;
        if (!(px < P.width && py < P.height)) continue;
        const size_t pix = (size_t)py * P.width + px;
        const uint32_t pw = ORD_STATE(1u, b);
        const bool touched = pw != NO_PRIM;
        const f4 c = {__uint_as_float(ORD_STATE(2u, b)), __uint_as_float(ORD_STATE(3u, b)), __uint_as_float(ORD_STATE(4u, b)), __uint_as_float(ORD_STATE(5u, b))};
        if (touched || !P.color_load) {
            if (P.color_format == 2) reinterpret_cast<float4*>(P.color)[pix] = make_float4(c.x, c.y, c.z, c.w);
            else reinterpret_cast<uint32_t*>(P.color)[pix] = pack_bgra8_srgb(c);
        }
        if (P.prim_out && (touched || !P.color_load)) P.prim_out[pix] = pw;
        if (P.depth && P.depth_store) P.depth[pix] = __uint_as_float(ORD_STATE(0u, b));
    }
}

#endif  // MIRHI_ORDERED_HIP_H
