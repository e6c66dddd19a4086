// mirhi_geometry.hip.h -- vertex_kernel + geometry_kernel: fetch, vertex shader position, clip, setup, pair-parallel binning
// Part of the single device translation unit mirhi_kernels.hip (included inside namespace mirhi).
#ifndef MIRHI_GEOMETRY_HIP_H
#define MIRHI_GEOMETRY_HIP_H

// Issue priority of the vertex / geometry kernels' waves (s_setprio): these kernels are a few latency-bound waves whose dependent chain is the
// frame's critical path; in a frame loop they share their SIMDs with other frames' raster waves, which are throughput work.
#ifndef MIRHI_GEOM_PRIO
#define MIRHI_GEOM_PRIO 0
#endif

// ------------------------------------------------------------------------------------------------
// a1/a2/a4: index fetch, vertex fetch, vertex-shader position
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t fetch_index(DrawRef D, uint32_t k) {
    if (D.index_type == 0) return D.first + k;
    uint32_t idx;
    if (D.index_type == 2) idx = reinterpret_cast<const uint16_t*>(D.ib)[D.first + k];
    else idx = reinterpret_cast<const uint32_t*>(D.ib)[D.first + k];
    return (uint32_t)((int32_t)idx + D.vertex_offset);
}

__device__ __forceinline__ f4 vs_position(DrawRef D, uint32_t vidx, f3* world) {
    const uint8_t* v = D.vb + (size_t)vidx * D.stride;
    f4 p = {ldf(v, 0), ldf(v, 4), ldf(v, 8), 1.0f};
    if (D.program == 0) {                                   // vertex/triangle.hlsl:19
        if (world) *world = {p.x, p.y, p.z};
        return p;
    }
    f4 w = mat4_mul(cf(D.object), p);                       // vertex/model.hlsl:44
    if (world) *world = {w.x, w.y, w.z};
    return mat4_mul(cf(D.camera) + 32, w);                  // :48 (viewProjection @128 B)
}

// ------------------------------------------------------------------------------------------------
// screen-space triangle record (TriRec, 48 B; big list), the bin record (BinRec, 32 B, relative to its tile) and the raster
// kernel's working form (64 B, LDS only)
// ------------------------------------------------------------------------------------------------
struct ScreenTri {
    int32_t X[3], Y[3];          // 1/256 px, orientation normalised (interior has E > 0)
    float z0, zx, zy;
    int32_t minx, maxx, miny, maxy;   // inclusive pixel bbox (scissor-clamped)
    uint32_t idk, boxed;
    int32_t vminx, vminy;             // smallest vertex coordinates, 1/256 px
    uint32_t compact;                 // the vertices lie within 65535 sub-pixels of (vminx, vminy) and no scissor cuts the box
};

__device__ __forceinline__ void store_tri(uint4* dst, const ScreenTri& t) {
    dst[0] = make_uint4((uint32_t)t.X[0], (uint32_t)t.Y[0], (uint32_t)t.X[1], (uint32_t)t.Y[1]);
    dst[1] = make_uint4((uint32_t)t.X[2], (uint32_t)t.Y[2], __float_as_uint(t.z0), __float_as_uint(t.zx));
    dst[2] = make_uint4(__float_as_uint(t.zy), t.idk, (uint32_t)t.minx | ((uint32_t)t.maxx << 16) | (t.boxed << 31),
                        (uint32_t)t.miny | ((uint32_t)t.maxy << 16));
}

// Tile-row split: the owned rows (PassParams::tile_row_begin / _step) inside the tile-row range [ty0, ty1], as owned-row indices [k0, k1]; k1 < k0: none.
__device__ __forceinline__ void owned_rows(ParamsRef P, int32_t ty0, int32_t ty1, int32_t& k0, int32_t& k1) {
    const int32_t first = (int32_t)P.tile_row_begin, count = (int32_t)P.tile_row_end - first, step = (int32_t)P.tile_row_step;
    const int32_t a = ty0 - first, b = ty1 - first;
    if (step <= 1) { k0 = max(a, 0); k1 = min(b, count - 1); return; }          // (wave-uniform: one contiguous band, or no split at all)
    k0 = a <= 0 ? 0 : (int32_t)(((uint32_t)a + (uint32_t)step - 1u) / (uint32_t)step);
    k1 = b < 0 ? -1 : min((int32_t)((uint32_t)b / (uint32_t)step), count - 1);
}

// ------------------------------------------------------------------------------------------------
// a5: clip-space triangle (all w > 0) -> snapped, culled, oriented screen triangle + depth plane
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool setup_triangle(ParamsRef P, DrawRef D, const f4 c[3], uint32_t prim,
                                               ScreenTri& t) {
    float z[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        if (!(c[i].w > 0.0f)) return false;
        const float iw = rcp_rn(c[i].w);                                  // IEEE 1/w (mirhi_exact.hip.h: verified for every binary32)
        const float xs = (c[i].x * iw) * D.hw + D.cx;                     // Vulkan viewport transform
        const float ys = (c[i].y * iw) * D.hh + D.cy;
        const float zs = (c[i].z * iw) * D.dscale + D.dmin;
        if (!(fabsf(xs) <= 16383.0f) || !(fabsf(ys) <= 16383.0f)) return false;
        t.X[i] = (int32_t)rintf(xs * 256.0f);                             // 8 sub-pixel bits, round-half-even
        t.Y[i] = (int32_t)rintf(ys * 256.0f);
        z[i] = zs;
    }
    const int64_t S = (int64_t)(t.X[1] - t.X[0]) * (int64_t)(t.Y[2] - t.Y[0]) -
                      (int64_t)(t.X[2] - t.X[0]) * (int64_t)(t.Y[1] - t.Y[0]);
    if (S == 0) return false;
    const bool front = (D.front_face == 0) ? (S < 0) : (S > 0);           // Vulkan: a = -S/2, CCW front <=> a > 0
    if (D.cull_mode == 3) return false;
    if (D.cull_mode == 2 && !front) return false;
    if (D.cull_mode == 1 && front) return false;
    if (S < 0) {
        int32_t ti = t.X[1]; t.X[1] = t.X[2]; t.X[2] = ti;
        ti = t.Y[1]; t.Y[1] = t.Y[2]; t.Y[2] = ti;
        float tz = z[1]; z[1] = z[2]; z[2] = tz;
    }
    const float inv256 = 1.0f / 256.0f;
    const float fx1 = (float)(t.X[1] - t.X[0]) * inv256, fy1 = (float)(t.Y[1] - t.Y[0]) * inv256;
    const float fx2 = (float)(t.X[2] - t.X[0]) * inv256, fy2 = (float)(t.Y[2] - t.Y[0]) * inv256;
    const float area = fx1 * fy2 - fx2 * fy1;
    const float dz1 = z[1] - z[0], dz2 = z[2] - z[0];
    t.zx = div_rn(dz1 * fy2 - dz2 * fy1, area);                           // IEEE quotients
    t.zy = div_rn(dz2 * fx1 - dz1 * fx2, area);
    t.z0 = z[0];
    const int32_t xmin = min(t.X[0], min(t.X[1], t.X[2])), xmax = max(t.X[0], max(t.X[1], t.X[2]));
    const int32_t ymin = min(t.Y[0], min(t.Y[1], t.Y[2])), ymax = max(t.Y[0], max(t.Y[1], t.Y[2]));
    int32_t px0 = (xmin + 127) >> 8, px1 = (xmax - 128) >> 8;
    int32_t py0 = (ymin + 127) >> 8, py1 = (ymax - 128) >> 8;
    const bool cut = D.scissor_partial && (px0 < D.sx0 || px1 > D.sx1 || py0 < D.sy0 || py1 > D.sy1);
    px0 = max(px0, D.sx0); px1 = min(px1, D.sx1); py0 = max(py0, D.sy0); py1 = min(py1, D.sy1);
    if (px0 > px1 || py0 > py1) return false;
    // a triangle that touches none of this device's tile rows is not rasterized here (tile-row split)
    {
        int32_t k0, k1;
        owned_rows(P, py0 >> TILE_LOG2, py1 >> TILE_LOG2, k0, k1);
        if (k1 < k0) return false;
    }
    t.minx = px0; t.maxx = px1; t.miny = py0; t.maxy = py1;
    t.boxed = cut ? 1u : 0u;
    t.vminx = xmin; t.vminy = ymin;
    t.compact = (!cut && xmax - xmin <= 65535 && ymax - ymin <= 65535) ? 1u : 0u;
    t.idk = P.idflip ? (MAX_PRIM_ID - prim) : prim;
    return true;
}

__device__ __forceinline__ void emit_big(ParamsRef P, const ScreenTri& t) {
    const uint32_t slot = atomicAdd(P.big_count, 1u);
    if (slot < P.big_cap) store_tri(reinterpret_cast<uint4*>(P.big_recs) + (size_t)slot * 3u, t);
    else __hip_atomic_fetch_or(P.status, STATUS_BIG_OVERFLOW, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// clip planes: near z>=0, far w-z>=0, guard band x,y within +-g*w (oracle: clip_polygon)
__device__ __forceinline__ float plane_dist(int plane, f4 c, float gx, float gy) {
    switch (plane) {
        case 1: return c.z;
        case 2: return c.w - c.z;
        case 4: return c.x + gx * c.w;
        case 8: return gx * c.w - c.x;
        case 16: return c.y + gy * c.w;
        default: return gy * c.w - c.y;
    }
}
__device__ __forceinline__ uint32_t outcode_clip(f4 c, float gx, float gy) {
    uint32_t oc = 0;
    if (c.z < 0.0f) oc |= 1;
    if (c.w - c.z < 0.0f) oc |= 2;
    if (c.x + gx * c.w < 0.0f) oc |= 4;
    if (gx * c.w - c.x < 0.0f) oc |= 8;
    if (c.y + gy * c.w < 0.0f) oc |= 16;
    if (gy * c.w - c.y < 0.0f) oc |= 32;
    return oc;
}
__device__ __forceinline__ uint32_t outcode_view(f4 c) {
    uint32_t oc = 0;
    if (c.x < -c.w) oc |= 1; if (c.x > c.w) oc |= 2;
    if (c.y < -c.w) oc |= 4; if (c.y > c.w) oc |= 8;
    if (c.z < 0.0f) oc |= 16; if (c.z > c.w) oc |= 32;
    return oc;
}

// Rare path: Sutherland-Hodgman in homogeneous space on a per-lane polygon in LDS (no scratch memory: a
// kernel that touches scratch pays ~5 us per launch on this part).  Every resulting fan triangle goes to
// the big list (the raster kernel builds its tile records), so this path needs no binning code.
constexpr int CLIP_MAX_VERTS = 10;     // 3 + one per plane (near, far, 4 guard-band planes) = 9
constexpr int CLIP_BATCH = 8;          // lanes clipping concurrently per wave (LDS polygon slots)

__device__ __forceinline__ void clip_and_emit(ParamsRef P, DrawRef D, f4 (*poly)[CLIP_MAX_VERTS],
                                              f4 c0, f4 c1, f4 c2, uint32_t any, uint32_t prim) {
    f4* in = poly[0]; f4* tmp = poly[1];
    in[0] = c0; in[1] = c1; in[2] = c2;
    int n = 3;
    for (int plane = 1; plane <= 32 && n >= 3; plane <<= 1) {
        if (!(any & plane)) continue;
        int m = 0;
        for (int i = 0; i < n; i++) {
            const f4 a = in[i], b = in[(i + 1) % n];
            const float da = plane_dist(plane, a, D.gx, D.gy), db = plane_dist(plane, b, D.gx, D.gy);
            const bool ina = da >= 0.0f, inb = db >= 0.0f;
            if (ina) tmp[m++] = a;
            if (ina != inb) {
                f4 p, q; float dp, dq;
                if (ina) { p = a; q = b; dp = da; dq = db; } else { p = b; q = a; dp = db; dq = da; }
                const float tt = div_rn(dp, dp - dq);
                tmp[m++] = {p.x + tt * (q.x - p.x), p.y + tt * (q.y - p.y), p.z + tt * (q.z - p.z), p.w + tt * (q.w - p.w)};
            }
        }
        n = m;
        f4* s = in; in = tmp; tmp = s;
    }
    if (!P.ordered_recs) {
        for (int i = 1; i + 1 < n; i++) {
            const f4 tri[3] = {in[0], in[i], in[i + 1]};
            ScreenTri t;
            if (setup_triangle(P, D, tri, prim, t)) emit_big(P, t);
        }
        return;
    }
    // ordered segment: the pieces must stay together at the triangle's position in primitive order.  They go to one
    // contiguous range of the big list (counted first, reserved with one atomic) and the triangle's ordered slot becomes a
    // marker {first piece, piece count}.
    uint32_t pieces = 0;
    for (int i = 1; i + 1 < n; i++) {
        const f4 tri[3] = {in[0], in[i], in[i + 1]};
        ScreenTri t;
        if (setup_triangle(P, D, tri, prim, t)) pieces++;
    }
    uint4* slot = reinterpret_cast<uint4*>(P.ordered_recs) + (size_t)(prim - P.ordered_first) * 3u;
    uint32_t first = 0;
    if (pieces) {
        first = atomicAdd(P.big_count, pieces);
        if (first + pieces > P.big_cap) { __hip_atomic_fetch_or(P.status, STATUS_BIG_OVERFLOW, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); pieces = 0; }
    }
    slot[0] = make_uint4(first, pieces, 0u, 0u);
    slot[1] = make_uint4(0u, 0u, 0u, 0u);
    slot[2] = make_uint4(0u, 0u, 1u, pieces ? ORDERED_MARKER : 0u);          // (bbox x = {min 1, max 0}: empty unless marked)
    uint32_t k = 0;
    for (int i = 1; i + 1 < n && pieces; i++) {
        const f4 tri[3] = {in[0], in[i], in[i + 1]};
        ScreenTri t;
        if (setup_triangle(P, D, tri, prim, t)) store_tri(reinterpret_cast<uint4*>(P.big_recs) + (size_t)(first + k++) * 3u, t);
    }
}

__device__ __forceinline__ uint32_t find_draw(ParamsRef P, uint32_t prim) {
    if (P.prim_draw) return P.prim_draw[prim - P.first_prim];
    uint32_t lo = 0, hi = P.num_draws;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (const_draws(P.draws)[mid].prim_base <= prim) lo = mid; else hi = mid;
    }
    return lo;
}

// ------------------------------------------------------------------------------------------------
// a4: vertex-shader pre-pass for the MODEL / MODEL_FULL programs (vertex/model.hlsl:39-68)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void vertex_body(const PassParams* __restrict__ params) {
    if (MIRHI_GEOM_PRIO) __builtin_amdgcn_s_setprio(MIRHI_GEOM_PRIO);
    ParamsRef P = *(ParamsPtr)(uintptr_t)params;
    const uint32_t slot0 = blockIdx.x * GEOM_THREADS;
    if (slot0 >= P.vs_total_slots) return;              // (batched launch: the grid is sized for the largest scope)
    const MIRHI_CONST VsJob* jobs = (const MIRHI_CONST VsJob*)(uintptr_t)P.vs_jobs;
    uint32_t lo = 0, hi = P.num_vs_jobs;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (jobs[mid].slot_base <= slot0) lo = mid; else hi = mid;
    }
    const MIRHI_CONST VsJob& J = jobs[lo];
    const uint32_t vidx = slot0 - J.slot_base + threadIdx.x;
    if (vidx >= J.count) return;
    const uint8_t* v = J.vb + (size_t)vidx * J.stride;
    const CFloatPtr model = cf(J.object);
    const f4 p = {ldf(v, 0), ldf(v, 4), ldf(v, 8), 1.0f};
    const f4 w = mat4_mul(model, p);                                         // vertex/model.hlsl:44
    const f4 c = mat4_mul(cf(J.camera) + 32, w);                             // :48
    const f3 n = {ldf(v, 12), ldf(v, 16), ldf(v, 20)};
    const f3 N = normalize3(mat3_mul(model + 16, n));                        // :51
    reinterpret_cast<uint4*>(J.out)[vidx] = make_uint4(__float_as_uint(c.x), __float_as_uint(c.y), __float_as_uint(c.z), __float_as_uint(c.w));
    // (out[1..]: the attribute stream behind the clip positions, words - 1 per vertex)
    uint4* out = reinterpret_cast<uint4*>(reinterpret_cast<uint8_t*>(J.out) + vs_attr_offset(J.count)) + (size_t)vidx * (J.words - 1u) - 1;
    out[1] = make_uint4(__float_as_uint(w.x), __float_as_uint(w.y), __float_as_uint(w.z), __float_as_uint(N.x));
    out[2] = make_uint4(__float_as_uint(N.y), __float_as_uint(N.z), ldu(v, 24), ldu(v, 28));
    if (J.words == 5) {
        const f3 t = {ldf(v, 32), ldf(v, 36), ldf(v, 40)};
        const float tw = ldf(v, 44);
        f3 T = normalize3(mat3_mul(model, t));                               // :52
        T = normalize3(sub3(T, scale3(N, dot3(T, N))));                      // :55 Gram-Schmidt
        const f3 B = scale3(cross3(N, T), tw);                               // :58
        out[3] = make_uint4(__float_as_uint(T.x), __float_as_uint(T.y), __float_as_uint(T.z), __float_as_uint(B.x));
        out[4] = make_uint4(__float_as_uint(B.y), __float_as_uint(B.z), 0u, 0u);
    }
}

__global__ __launch_bounds__(GEOM_THREADS) void vertex_kernel(const PassParams* __restrict__ params) { vertex_body(params); }
__global__ __launch_bounds__(GEOM_THREADS) void vertex_kernel_batch(const GeometryBatch B) { vertex_body(B.params[blockIdx.y]); }

__device__ __forceinline__ uint32_t pack_bgra8_srgb(f4 c);

// Pair-parallel binning: the wave's (triangle, bin) pairs -- one per triangle for fine meshes, about five for scattered
// 50-pixel triangles, sixteen at most -- are enumerated densely through LDS and dealt to the lanes 64 at a time, one
// record copy per pair.  Every lane stays busy; a lane walking its own <= 16 bins (the first design) issued 2-3x the
// instructions per wave, and sixteen lanes per triangle 10x (measured: C2 907 / 987 / 1089 Mtris/s for walk / wide / pairs).
constexpr uint32_t PAIR_MAX = GEOM_THREADS * MAX_BIN_SPAN * MAX_BIN_SPAN;

// Reserves one bin slot for every active lane with a SINGLE returning atomic instruction: lanes that target the same
// tile (mesh order: most of a wave) are grouped first -- pure ballot arithmetic, no memory -- and only a group's first
// lane adds, the whole group size at once; lanes left over after GROUP_ROUNDS groups, or after the first group too
// small to be worth a round (scattered input), add 1 for themselves in the same instruction.  Returns the value the
// lane's own add fetched (meaningful on reserving lanes) and in `who` the reserving lane | rank within its group << 8.
// One atomic per group IN SEPARATE ROUNDS (the first design) made the compiler wait for each result before the next
// add -- up to nine serial round trips per wave, 13 us of the dancer asset's 29 us geometry time.
// Per-XCD bins (PassParams::count_stride != 0, scopes whose triangles sit in few tiles): the counter a wave adds to is its
// XCD's own copy.  Measured (tools/microbench/atomic_contention.hip, 17k returning atomics on 232 addresses, the dancer's
// pattern): 14.4 us when all eight XCDs share the counters, 4.9 us with one copy per XCD, 2.4 us without any sharing --
// a contended line ping-pongs between the XCDs' L2s at ~190 ns per atomic.
__device__ __forceinline__ uint32_t xcd_of_wave() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u; }   // HW_REG_XCC_ID[3:0]

__device__ __forceinline__ uint32_t reserve_bin_slots(ParamsRef P, bool act, uint32_t tile, uint32_t lane, uint64_t lt, uint32_t& who, uint32_t xcd) {
    constexpr int GROUP_ROUNDS = 12, GROUP_MIN = 2;
    who = lane;
    uint32_t gsize = act ? 1u : 0u;
    uint64_t rem = __ballot(act);
    for (int round = 0; round < GROUP_ROUNDS && rem; round++) {
        const int leader = __ffsll((long long)rem) - 1;
        const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)tile, leader);
        const uint64_t grp = __ballot(act && tile == t0) & rem;
        const uint32_t n = (uint32_t)__popcll(grp);
        if (n < (uint32_t)GROUP_MIN) break;
        if ((grp >> lane) & 1ull) { who = (uint32_t)leader | ((uint32_t)__popcll(grp & lt) << 8); gsize = (int)lane == leader ? n : 0u; }
        rem &= ~grp;
    }
    uint32_t raw = 0;
    if (gsize) raw = atomicAdd(&P.bin_count[(xcd * P.count_stride + tile) * BIN_COUNT_STRIDE], gsize);
    return raw;
}

// ------------------------------------------------------------------------------------------------
// bin pages.  A list's slot s lives in its page s / 64 at offset s % 64.  Without per-XCD lists the first fixed_recs / 64 pages of
// a tile have a fixed place in the pool (as many as the scope's average density fills): the usual bin costs no allocation.  Any
// other page is taken from the pool by the ONE lane
// whose slot is the page's first (s % 64 == 0) and published in the tile's row of the page table; the lanes that drew the
// other slots of that page (this wave or another) read the entry until it is there.  Waiting is safe: the publishing lane has
// executed its reservation (its slot is smaller), and a wave publishes every page of a batch of reservations before it waits
// for anyone (bin_triangle_pairs), so no wave ever waits for a wave that waits for it.  The wait is bounded all the same.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t bin_table_entry(ParamsRef P, uint32_t tile, uint32_t xcd, uint32_t slot) {
    return tile * (uint32_t)BIN_TABLE_ROW + (P.count_stride ? xcd * 8u : 0u) + (slot >> BIN_PAGE_LOG2);
}
__device__ __forceinline__ uint32_t bin_page_alloc(ParamsRef P, uint32_t entry, uint32_t xcd) {
    // this XCD's share of the pool first, then the others' (rare: the shares are equal and the load is spread by the dispatcher)
    uint32_t page = PAGE_NONE;
    for (uint32_t d = 0; d < 8u && page == PAGE_NONE; d++) {
        const uint32_t x = (xcd + d) & 7u;
        const uint32_t id = atomicAdd(&P.pool_next[x * (uint32_t)POOL_COUNTER_STRIDE], 1u);
        if (id < P.pool_dyn_pages) page = P.pool_dyn_base + x * P.pool_dyn_pages + id;
    }
    // pool exhausted: the page's records take the big list; the host grows the pool for the next submit
    if (page == PAGE_NONE) __hip_atomic_fetch_or(P.status, STATUS_POOL_EXHAUSTED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&P.bin_table[entry], page, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return page;
}
__device__ __forceinline__ uint32_t bin_page_wait(ParamsRef P, uint32_t entry) {
    for (uint32_t spin = 0; spin < (1u << 22); spin++) {
        const uint32_t page = __hip_atomic_load(&P.bin_table[entry], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (page != PAGE_EMPTY) return page;
        __builtin_amdgcn_s_sleep(2);
    }
    __hip_atomic_fetch_or(P.status, STATUS_PAGE_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return PAGE_NONE;
}
// the triangle relative to tile (tx, ty): c0 = { vminx, vminy, x0|y0<<16, x1|y1<<16 }, c1 = { x2|y2<<16, z0, zx, zy }
__device__ __forceinline__ void store_bin_rec(ParamsRef P, uint32_t page, uint32_t slot, uint4 c0, uint4 c1, uint32_t idk, int32_t tx, int32_t ty) {
    const int32_t ox = (int32_t)c0.x - tx * (TILE * 256), oy = (int32_t)c0.y - ty * (TILE * 256);
    uint4* dst = reinterpret_cast<uint4*>(P.bin_pool) + ((size_t)page * BIN_PAGE_RECS + (slot & (BIN_PAGE_RECS - 1u))) * 2u;
    dst[0] = make_uint4(((uint32_t)ox & 0xFFFFu) | ((uint32_t)oy << 16), c0.z, c0.w, c1.x);
    dst[1] = make_uint4(c1.y, c1.z, c1.w, idk);
}

// overlap(): work of the caller that depends on nothing here, run exactly once right behind the first batch of returning atomics --
// a geometry wave is alone on its SIMD (10k triangles are 157 waves on 1024 SIMDs), so whatever it computes while its reservations
// are on their way through the fabric is free (the flat-colour packing: 1.1 us of a wave's 8.1 us, tools/stamps.py).
template <uint32_t BATCH, typename Overlap>
__device__ __forceinline__ void bin_triangle_pairs(ParamsRef P, bool valid, const ScreenTri& t, uint4 (*lds_tri)[3],
                                                   uint32_t* lds_meta, uint16_t* lds_owner, Overlap&& overlap) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t lt = (1ull << lane) - 1ull;
    const uint32_t hw_xcd = xcd_of_wave();                           // wave-uniform
    const uint32_t xcd = P.count_stride ? hw_xcd : 0u;               // the list this wave appends to (per-XCD bins)
    const uint32_t fixed_recs = P.fixed_recs;                        // slots below live in the tile's own fixed pages
    // ty0: the first OWNED row of the triangle's span as an owned-row index (bins and counters are indexed by it); the row itself is
    // row_first + (ty0 + j) * row_step for the j-th owned row of the span
    const int32_t row_first = (int32_t)P.tile_row_begin, row_step = max((int32_t)P.tile_row_step, 1);
    int32_t tx0 = 0, ty0 = 0, ntx = 0, nty = 0;
    bool spill = false;
    if (valid) {
        int32_t k1;
        tx0 = t.minx >> TILE_LOG2;
        owned_rows(P, t.miny >> TILE_LOG2, t.maxy >> TILE_LOG2, ty0, k1);
        ntx = (t.maxx >> TILE_LOG2) - tx0 + 1;
        nty = k1 - ty0 + 1;
        // a bin record is 16-bit relative to its tile: the triangle's smallest coordinates must be within reach of every tile of
        // its span (they always are unless the triangle hangs far out of the target or the band)
        const int32_t tw = TILE * 256;
        const bool fits = t.compact && t.vminx - tx0 * tw <= 32767 && t.vminx - (tx0 + ntx - 1) * tw >= -32768 &&
                          t.vminy - (row_first + ty0 * row_step) * tw <= 32767 && t.vminy - (row_first + k1 * row_step) * tw >= -32768;
        spill = ntx > MAX_BIN_SPAN || nty > MAX_BIN_SPAN || !fits;
    }
    const bool binned = valid && !spill;
    const uint32_t nb = binned ? (uint32_t)(ntx * nty) : 0u;
    // the triangle relative to its smallest coordinates (store_bin_rec); built where it is used, not held across the branches
#define MIRHI_COMPACT_C0 make_uint4((uint32_t)t.vminx, (uint32_t)t.vminy, (uint32_t)(t.X[0] - t.vminx) | ((uint32_t)(t.Y[0] - t.vminy) << 16), \
                                    (uint32_t)(t.X[1] - t.vminx) | ((uint32_t)(t.Y[1] - t.vminy) << 16))
#define MIRHI_COMPACT_C1 make_uint4((uint32_t)(t.X[2] - t.vminx) | ((uint32_t)(t.Y[2] - t.vminy) << 16), __float_as_uint(t.z0), __float_as_uint(t.zx), __float_as_uint(t.zy))
    if (__ballot(nb > 1u) == 0ull) {
        // Fine meshes: no triangle of the wave overlaps more than one tile.  A lane is its own pair -- no enumeration
        // through LDS, the record goes out of the registers it was built in.
        const uint32_t tile = (uint32_t)ty0 * P.tiles_x + (uint32_t)tx0;
        const bool act = nb == 1u;
        uint32_t who;
        const uint32_t raw = reserve_bin_slots(P, act, tile, lane, lt, who, xcd);
        overlap();
        const uint32_t slot = (uint32_t)__shfl((int)raw, (int)(who & 0xFFu)) + (who >> 8);
        // Opening lanes first, as a step of its own: the lane that opens a page and the lanes that wait for it may sit in this very
        // wave, and lanes on the other side of a branch do not run until this side is through.
        uint32_t page = PAGE_EMPTY;
        const bool in_list = act && slot < P.sub_cap;
        if (in_list && slot < fixed_recs) page = (tile * fixed_recs + slot) >> BIN_PAGE_LOG2;
        if (in_list && page == PAGE_EMPTY && (slot & (BIN_PAGE_RECS - 1u)) == 0u) page = bin_page_alloc(P, bin_table_entry(P, tile, xcd, slot), hw_xcd);
        if (in_list && page == PAGE_EMPTY) page = bin_page_wait(P, bin_table_entry(P, tile, xcd, slot));
        if (act) {
            if (in_list && page != PAGE_NONE) store_bin_rec(P, page, slot, MIRHI_COMPACT_C0, MIRHI_COMPACT_C1, t.idk, tx0, row_first + ty0 * row_step);
            else spill = true;       // list full or pool exhausted: the triangle goes to the big list
        }
        if (valid && spill) emit_big(P, t);
        return;
    }
    // exclusive prefix sum of nb (<= 16) over the wave from five bit planes of ballots
    uint32_t ex = 0, total = 0;
#pragma unroll
    for (uint32_t bit = 0; bit < 5; bit++) {
        const uint64_t m = __ballot(((nb >> bit) & 1u) != 0u);
        ex += (uint32_t)__popcll(m & lt) << bit;
        total += (uint32_t)__popcll(m) << bit;
    }
    if (binned) {
        lds_tri[lane][0] = MIRHI_COMPACT_C0;
        lds_tri[lane][1] = MIRHI_COMPACT_C1;
        lds_tri[lane][2] = make_uint4(t.idk, (uint32_t)tx0 | ((uint32_t)ty0 << 16), 0u, 0u);
        lds_meta[lane] = (uint32_t)ty0 * P.tiles_x + (uint32_t)tx0;
        uint32_t pos = ex;
#pragma unroll
        for (uint32_t k = 0; k < (uint32_t)(MAX_BIN_SPAN * MAX_BIN_SPAN); k++) {
            if ((int32_t)(k % MAX_BIN_SPAN) < ntx && (int32_t)(k / MAX_BIN_SPAN) < nty) lds_owner[pos++] = (uint16_t)(lane | (k << 8));
        }
    }
    __syncthreads();            // one wave per workgroup: orders the LDS writes above before the reads below
    // Three phases per batch of up to BATCH rounds (64 pairs each):
    //   1. the returning atomics of all rounds (one instruction per 64 pairs, reserve_bin_slots) are issued before the first
    //      result is consumed.  BATCH = 8 for the small-scope variant of the kernel: with four, C2 (about 320 pairs a wave) paid
    //      a second serial round trip, 7.4 -> 8.2 us; 4 for the occupancy-oriented one (8 spills at 72 VGPRs).
    //   2. lanes whose slot opens a page take it from the pool and publish it -- for ALL rounds, before
    //   3. the records are stored; a lane whose page another lane opens waits for the table entry here (see "bin pages").
#pragma unroll 1
    for (uint32_t it0 = 0; it0 * GEOM_THREADS < total; it0 += BATCH) {
        uint32_t slot[BATCH];      // phase 1: atomic result (held by the reserving lane); from phase 2 on: the lane's own slot
        uint32_t aux[BATCH];       // phase 1: reserving lane | rank within its group << 8; from phase 2 on: the pool page, if known
#pragma unroll
        for (uint32_t k = 0; k < BATCH; k++) {
            slot[k] = 0; aux[k] = lane;
            const uint32_t p = (it0 + k) * GEOM_THREADS + lane;
            if ((it0 + k) * GEOM_THREADS >= total) continue;
            const bool act = p < total;
            uint32_t tile = 0;
            if (act) {
                const uint32_t o = lds_owner[p], kk = o >> 8;
                tile = (lds_meta[o & 0xFFu] & 0x7FFFFFFFu) + (kk / MAX_BIN_SPAN) * P.tiles_x + (kk % MAX_BIN_SPAN);
            }
            slot[k] = reserve_bin_slots(P, act, tile, lane, lt, aux[k], xcd);
        }
        if (it0 == 0u) { GSTAMP(7); overlap(); }
#pragma unroll
        for (uint32_t k = 0; k < BATCH; k++) {
            if ((it0 + k) * GEOM_THREADS >= total) continue;
            const uint32_t p = (it0 + k) * GEOM_THREADS + lane;
            const uint32_t s = (uint32_t)__shfl((int)slot[k], (int)(aux[k] & 0xFFu)) + (aux[k] >> 8);
            slot[k] = s; aux[k] = PAGE_EMPTY;
            if (p < total && s < P.sub_cap && (s & (BIN_PAGE_RECS - 1u)) == 0u && s >= fixed_recs) {
                const uint32_t o = lds_owner[p], kk = o >> 8;
                const uint32_t tile = (lds_meta[o & 0xFFu] & 0x7FFFFFFFu) + (kk / MAX_BIN_SPAN) * P.tiles_x + (kk % MAX_BIN_SPAN);
                aux[k] = bin_page_alloc(P, bin_table_entry(P, tile, xcd, s), hw_xcd);
            }
        }
#pragma unroll
        for (uint32_t k = 0; k < BATCH; k++) {
            if ((it0 + k) * GEOM_THREADS >= total) continue;
            const uint32_t p = (it0 + k) * GEOM_THREADS + lane;
            if (p < total) {
                const uint32_t o = lds_owner[p], ol = o & 0xFFu, kk = o >> 8;
                const uint32_t tile = (lds_meta[ol] & 0x7FFFFFFFu) + (kk / MAX_BIN_SPAN) * P.tiles_x + (kk % MAX_BIN_SPAN);
                const uint32_t s = slot[k];
                uint32_t page = PAGE_NONE;
                if (s < P.sub_cap) {
                    if (s < fixed_recs) page = (tile * fixed_recs + s) >> BIN_PAGE_LOG2;
                    else if (aux[k] != PAGE_EMPTY) page = aux[k];
                    else page = bin_page_wait(P, bin_table_entry(P, tile, xcd, s));
                }
                if (page != PAGE_NONE) {
                    const uint4 m2 = lds_tri[ol][2];
                    store_bin_rec(P, page, s, lds_tri[ol][0], lds_tri[ol][1], m2.x, (int32_t)(m2.y & 0xFFFFu) + (int32_t)(kk % MAX_BIN_SPAN),
                                  row_first + ((int32_t)(m2.y >> 16) + (int32_t)(kk / MAX_BIN_SPAN)) * row_step);
                } else {
                    atomicOr(&lds_meta[ol], 0x80000000u);   // list full / pool exhausted: the owner sends the triangle to the big list, once
                }
            }
        }
    }
    if (total == 0u) overlap();     // (no pair in the whole wave: the batch loop never ran)
    __syncthreads();
    if (binned && (lds_meta[lane] >> 31)) spill = true;    // (idempotent resolve: being in some bins as well is harmless)
    if (valid && spill) emit_big(P, t);
#undef MIRHI_COMPACT_C0
#undef MIRHI_COMPACT_C1
}

// One wave per workgroup; draws are padded to whole waves so the draw (and with it every uniform, pointer
// and pipeline-state word) is wave-uniform and lives in SGPRs.  One lane per triangle up to the screen-space setup,
// then bin_triangle_pairs.
// WAVES: resident waves per SIMD the register allocation aims at.  7 (72 VGPRs) for scopes with enough triangles to fill the
// chip several times over (C4 geometry 61.4 -> 57.4 us); 5 (up to 102) for small scopes, where a wave is alone on its SIMD
// and only the length of its dependent instruction stream counts (C2: 7.4 us against 8.2 with the tighter allocation).
template <int WAVES>
__device__ __forceinline__ void geometry_body(const PassParams* __restrict__ params, const GeometryHead& H) {
    ParamsRef P = *(ParamsPtr)(uintptr_t)params;
    // 5.4 KB of LDS per one-wave workgroup: 30 fit a CU, which is what lets 7 waves per SIMD be resident (at 7.9 KB the LDS
    // capped the kernel at 5).  The clipper's polygon slots (2.5 KB: clipping lanes take turns, 8 at a time) reuse the
    // triangle staging area of the binning step, which is over by then (one wave: LDS accesses stay in program order).
    __shared__ uint4 lds_tri[GEOM_THREADS][3];
    static_assert(sizeof(f4) * CLIP_BATCH * 2 * CLIP_MAX_VERTS <= sizeof(uint4) * GEOM_THREADS * 3, "polygon slots must fit the staging area");
    f4 (*poly)[2][CLIP_MAX_VERTS] = reinterpret_cast<f4 (*)[2][CLIP_MAX_VERTS]>(&lds_tri[0][0]);
    __shared__ uint32_t lds_meta[GEOM_THREADS];
    __shared__ uint16_t lds_owner[PAIR_MAX];
    if (MIRHI_GEOM_PRIO) __builtin_amdgcn_s_setprio(MIRHI_GEOM_PRIO);
    GSTAMP(0);
    // H.tris_per_wave (64, 32 or 16; divides 64, and draws are padded to 64 slots, so a workgroup never straddles two draws): a small scope
    // is a few latency-bound waves on a mostly idle chip -- with fewer triangles per wave the (triangle, tile) pairs of a wave, which all 64
    // lanes share out among themselves (bin_triangle_pairs), take fewer rounds, and the wave's dependent chain is that much shorter.
    const uint32_t tpw = H.tris_per_wave;
    const uint32_t slot0 = blockIdx.x * tpw;
    // One non-indexed TRIANGLE-program draw (GeometryHead::vb0): the vertices are requested here, off the kernel arguments alone -- the loads are on their
    // way while the draw descriptor's and the parameters' scalar loads are (a scope of one draw: slot = triangle)
    const bool head_fetch = H.vb0 != nullptr;
    uint32_t hv[3][6];
    if (head_fetch && threadIdx.x < tpw && slot0 + threadIdx.x < H.tris0) {
#pragma unroll
        for (uint32_t k = 0; k < 3; k++) {
            const uint8_t* v = H.vb0 + (size_t)(H.first0 + 3u * (slot0 + threadIdx.x) + k) * H.stride0;
#pragma unroll
            for (uint32_t w = 0; w < 6; w++) hv[k][w] = ldu(v, 4u * w);
        }
    }
    uint32_t lo = 0, hi = H.num_draws;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (const_draws(H.draws)[mid].slot_base <= slot0) lo = mid; else hi = mid;
    }
    DrawRef D = const_draws(H.draws)[lo];
    const uint32_t tri = slot0 - D.slot_base + threadIdx.x;
    const bool has_tri = threadIdx.x < tpw && tri < D.tri_count;
    GSTAMP_SYNC(4);
    const uint32_t prim = D.prim_base + tri;
    bool valid = false;
    uint32_t any = 0;
    ScreenTri t;
    f4 c[3];
    // TRIANGLE program on an sRGB8 target: the vertex colours come in with the positions (same 24-byte vertices, same cache lines --
    // fetched later they were a second round trip of 0.5 us), the packed flat colour is computed behind the bin reservations
    // (only in the small-scope register allocation: the occupancy-oriented one has no register to spare and its scopes are meshes)
    constexpr bool EARLY_COLOUR = WAVES < 7;
    uint32_t col0[3] = {0u, 0u, 0u};       // colour of vertex 0; is_flat: the other two vertices carry the same bits
    bool is_flat = true;
    const bool want_flat = EARLY_COLOUR && P.flat_color != nullptr && D.program == 0;      // (evaluated where it is used in the other allocation)
    bool dropped = false;
    if (D.program == 3) {
        // pixel/model_pbr.hlsl:174-178 `if (baseColor.a < alphaCutoff) discard;` decided per draw where one decision covers it: alpha is
        // baseColorFactor.a, or a texel alpha in [0,1] times it.  A draw whose texels could fall on both sides of the cutoff needs the
        // discard per fragment, before the depth write: fragment_discard_enable routes the draw to a scope that does that (alpha_scope:
        // raster_small_masked tests alpha in front of the depth key; an ordered segment: ordered_record); anywhere else it is reported,
        // not rendered.
        const CBytePtr M = cb(D.material);
        const float fa = ldcf(M, 12), cutoff = ldcf(M, 44);
        float lo = fa, hi = fa;
        if (ldcu(M, 48) != 0u) { lo = fa < 0.0f ? fa : 0.0f; hi = fa > 0.0f ? fa : 0.0f; }
        if (hi < cutoff) dropped = true;
        else if (!(lo >= cutoff) && !P.ordered_recs && !P.alpha_scope) {
            dropped = true;
            if (threadIdx.x == 0) __hip_atomic_fetch_or(P.status, STATUS_ALPHA_TEST_TEXTURED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (has_tri && !dropped && head_fetch) {
#pragma unroll
        for (uint32_t k = 0; k < 3; k++) {
            c[k] = {__uint_as_float(hv[k][0]), __uint_as_float(hv[k][1]), __uint_as_float(hv[k][2]), 1.0f};          // vertex/triangle.hlsl:19
            if (EARLY_COLOUR && want_flat) {
                if (k == 0) { col0[0] = hv[0][3]; col0[1] = hv[0][4]; col0[2] = hv[0][5]; }
                else is_flat = is_flat && hv[k][3] == col0[0] && hv[k][4] == col0[1] && hv[k][5] == col0[2];
            }
        }
    }
    if (has_tri && !dropped && !head_fetch) {
#pragma unroll
        for (uint32_t k = 0; k < 3; k++) {
            const uint32_t vidx = fetch_index(D, 3u * tri + k);
            if (D.vs_words) {                      // MODEL programs: clip position from the vertex pre-pass
                const uint4 w0 = reinterpret_cast<const uint4*>(D.vs_out)[vidx];
                c[k] = {__uint_as_float(w0.x), __uint_as_float(w0.y), __uint_as_float(w0.z), __uint_as_float(w0.w)};
            } else {
                c[k] = vs_position(D, vidx, nullptr);
                if (EARLY_COLOUR && want_flat) {
                    const uint8_t* v = D.vb + (size_t)vidx * D.stride;
                    const uint32_t r = ldu(v, 12), g = ldu(v, 16), b = ldu(v, 20);
                    if (k == 0) { col0[0] = r; col0[1] = g; col0[2] = b; }
                    else is_flat = is_flat && r == col0[0] && g == col0[1] && b == col0[2];
                }
            }
        }
    }
    if (has_tri && !dropped) {
        GSTAMP_SYNC(5);
        const uint32_t o0 = outcode_view(c[0]), o1 = outcode_view(c[1]), o2 = outcode_view(c[2]);
        if (!(o0 & o1 & o2)) {
            any = outcode_clip(c[0], D.gx, D.gy) | outcode_clip(c[1], D.gx, D.gy) | outcode_clip(c[2], D.gx, D.gy);
            bool in_band = true;
            if (any == 0 && (P.tile_row_step > 1u || P.tile_row_begin != 0u || P.tile_row_end != P.tiles_y)) {
                // Tile-row split: every rank sees all triangles and most of them touch none of its rows.  Those are dropped before the setup arithmetic
                // (the bulk of this kernel's instructions) by a conservative test: the vertices' window y from the 1-ulp reciprocal, a pixel of margin
                // either way for the snap and for the approximation (|error| < 0.01 px inside the guard band), the tile rows that range spans, and whether
                // one of them is this rank's (owned_rows).  Whatever passes takes the exact path; what that drops, this may keep -- never the reverse.
                if (c[0].w > 0.0f && c[1].w > 0.0f && c[2].w > 0.0f) {
                    const float y0 = c[0].y * __builtin_amdgcn_rcpf(c[0].w), y1 = c[1].y * __builtin_amdgcn_rcpf(c[1].w), y2 = c[2].y * __builtin_amdgcn_rcpf(c[2].w);
                    const float lo = fminf(y0, fminf(y1, y2)) * D.hh + D.cy - 1.0f, hi = fmaxf(y0, fmaxf(y1, y2)) * D.hh + D.cy + 1.0f;
                    // (window y grows with clip y when hh > 0; a flipped viewport swaps the two: take the range either way)
                    const float ylo = fminf(lo + 1.0f, hi - 1.0f) - 1.0f, yhi = fmaxf(lo + 1.0f, hi - 1.0f) + 1.0f;
                    const int32_t r0 = (int32_t)floorf(fmaxf(ylo, -65536.0f)) >> TILE_LOG2, r1 = (int32_t)floorf(fminf(yhi, 65536.0f)) >> TILE_LOG2;
                    int32_t k0, k1;
                    owned_rows(P, r0, r1, k0, k1);
                    in_band = k1 >= k0;
                }
            }
            if (any == 0 && in_band) valid = setup_triangle(P, D, c, prim, t);
        }
    }
    GSTAMP_SYNC(6);
    if (P.prim_draw && has_tri) P.prim_draw[prim - P.first_prim] = lo;      // (several draws in the scope)
    auto flat_colour = [&]() {
        if (EARLY_COLOUR && want_flat && (valid || any)) {
            // flat-shaded triangle (all three vertex colours equal): shade it once here instead of once per pixel
            P.flat_color[prim] = is_flat ? pack_bgra8_srgb({__uint_as_float(col0[0]), __uint_as_float(col0[1]), __uint_as_float(col0[2]), 1.0f}) : 0u;
        }
    };
    if constexpr (!EARLY_COLOUR) {      // (72-register allocation: as before, in front of the binning -- nothing may stay live across it that need not)
        if (P.flat_color && D.program == 0 && (valid || any)) {
            const uint8_t* v0 = D.vb + (size_t)fetch_index(D, 3u * tri) * D.stride;
            const uint8_t* v1 = D.vb + (size_t)fetch_index(D, 3u * tri + 1u) * D.stride;
            const uint8_t* v2 = D.vb + (size_t)fetch_index(D, 3u * tri + 2u) * D.stride;
            const uint32_t r = ldu(v0, 12), g = ldu(v0, 16), b = ldu(v0, 20);
            const bool flat = r == ldu(v1, 12) && r == ldu(v2, 12) && g == ldu(v1, 16) && g == ldu(v2, 16) && b == ldu(v1, 20) && b == ldu(v2, 20);
            P.flat_color[prim] = flat ? pack_bgra8_srgb({__uint_as_float(r), __uint_as_float(g), __uint_as_float(b), 1.0f}) : 0u;
        }
    }
    GSTAMP(1);
    if (P.ordered_recs) {
        // ordered segment: no bins -- triangle t of the segment sits at ordered_recs[t], in primitive order by construction
        if (has_tri && !dropped && any == 0) {
            uint4* slot = reinterpret_cast<uint4*>(P.ordered_recs) + (size_t)(prim - P.ordered_first) * 3u;
            if (valid) store_tri(slot, t);
            else { slot[0] = make_uint4(0u, 0u, 0u, 0u); slot[1] = make_uint4(0u, 0u, 0u, 0u); slot[2] = make_uint4(0u, 0u, 1u, 0u); }   // empty pixel box
        } else if (has_tri && dropped) {
            uint4* slot = reinterpret_cast<uint4*>(P.ordered_recs) + (size_t)(prim - P.ordered_first) * 3u;
            slot[0] = make_uint4(0u, 0u, 0u, 0u); slot[1] = make_uint4(0u, 0u, 0u, 0u); slot[2] = make_uint4(0u, 0u, 1u, 0u);
        }
        if constexpr (EARLY_COLOUR) flat_colour();
    } else if constexpr (EARLY_COLOUR) {
        bin_triangle_pairs<8u>(P, valid, t, lds_tri, lds_meta, lds_owner, flat_colour);
    } else {
        bin_triangle_pairs<4u>(P, valid, t, lds_tri, lds_meta, lds_owner, [] {});
    }
    GSTAMP(2);
    uint64_t todo = __ballot(any != 0);
    while (todo) {                                   // rare: triangles crossing the near / far / guard planes
        const uint32_t rank = (uint32_t)__popcll(todo & ((1ull << (threadIdx.x & 63u)) - 1ull));
        const bool mine = any != 0 && ((todo >> (threadIdx.x & 63u)) & 1ull) && rank < CLIP_BATCH;
        if (mine) clip_and_emit(P, D, poly[rank], c[0], c[1], c[2], any, prim);
        todo &= ~__ballot(mine);
    }
    GSTAMP(3);
}

template <int WAVES>
__global__ __launch_bounds__(GEOM_THREADS) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void geometry_kernel(const PassParams* __restrict__ params, const GeometryHead H) {
    geometry_body<WAVES>(params, H);
}
// the scopes of one batched submit (see raster_kernel_batch): grid (waves of the largest scope, scopes)
template <int WAVES>
__global__ __launch_bounds__(GEOM_THREADS) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void geometry_kernel_batch(const GeometryBatch B) {
    const uint32_t y = blockIdx.y;
    if (blockIdx.x >= B.blocks[y]) return;
    const GeometryHead H = B.head[y];
    geometry_body<WAVES>(B.params[y], H);
}

#endif  // MIRHI_GEOMETRY_HIP_H
