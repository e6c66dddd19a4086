// mirhi_kernels.hip -- gfx950 (CDNA4) kernels of the compute rasterizer.
//
// Two kernels per rendering scope (DESIGN.md "Kernels"):
//   geometry_kernel  one lane per input triangle: index + vertex fetch, vertex-shader position,
//                    clip / divide / viewport / snap / cull / depth-plane setup, then tile binning.
//                    Restates SURVEY 8a rows a1, a2, a4, a5 (crates/rhi/src/vertex.rs:20-61,88-170;
//                    command.rs:583-628; shaders/hlsl/vertex/{triangle,model}.hlsl;
//                    pipeline.rs:645-698,976-986; renderer.rs:504-518).
//   raster_kernel    one 256-lane workgroup per 32x32 tile: stages the tile's triangle records
//                    through LDS, resolves coverage (integer edge functions, top-left rule) and
//                    depth (64-bit key, registers only -- depth never leaves the CU unless a depth
//                    image is attached), then runs the fragment programs on the winning primitive
//                    of each pixel and stores the colour once.  Rows a6-a9 (pipeline.rs:976-1025;
//                    rendering.rs:102-115,356-370; depth_buffer.rs:48; shaders/hlsl/pixel/*.hlsl;
//                    lights.hlsli; swapchain.rs:561-570).
//
// Everything that decides coverage, depth or the winning primitive is integer arithmetic or IEEE
// binary32 {+,-,*,/} in the order fixed by DESIGN.md "Pipeline specification"; this file is
// compiled with -ffp-contract=off so results are bit-identical to the CPU oracle.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdint.h>

#include "mirhi_device.h"
#include "mirhi_launch.h"

namespace mirhi {

#ifdef MIRHI_STAMPS
// Diagnostic build only (build.py --stamps -> libmirhi_stamps.so): per-wave s_memtime stamps at phase
// boundaries, written to a buffer nothing else reads.  Never compiled into libmirhi.so.
__device__ uint64_t g_stamps[16384 * 8];
__device__ uint64_t g_stamps_geo[16384 * 4];
#define GSTAMP(k) do { if ((threadIdx.x & 63u) == 0 && blockIdx.x < 16384u) g_stamps_geo[blockIdx.x * 4u + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define STAMP(k) do { if ((threadIdx.x & 63u) == 0) { const uint32_t wv = ((blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)); \
    if (wv < 16384u) g_stamps[wv * 8u + (k)] = __builtin_amdgcn_s_memtime(); } } while (0)
// instruction-count attribution: the raster kernel returns after stage g_stage_limit (1 prologue, 2 fill of the first
// chunk, 3 both lists); the SQ instruction counters of such runs, differenced, give the dynamic cost of each stage
__device__ uint32_t g_stage_limit;
#define STAGE_END(k) do { if (g_stage_limit == (k)) return; } while (0)
#else
#define STAMP(k) do {} while (0)
#define GSTAMP(k) do {} while (0)
#define STAGE_END(k) do {} while (0)
#endif

struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

__device__ __forceinline__ float ldf(const uint8_t* p, uint32_t off) { return *reinterpret_cast<const float*>(p + off); }
__device__ __forceinline__ uint32_t ldu(const uint8_t* p, uint32_t off) { return *reinterpret_cast<const uint32_t*>(p + off); }

// Draw descriptors and uniform blocks are read-only for the whole launch.  Reading them through the constant
// address space lets the compiler use scalar loads (SGPRs, scalar cache) whenever the address is wave-uniform;
// through a plain pointer it must assume the kernel's own stores may alias and falls back to vector loads.
#define MIRHI_CONST __attribute__((address_space(4)))
typedef const MIRHI_CONST DrawDesc* DrawPtr;
typedef const MIRHI_CONST float* CFloatPtr;
typedef const MIRHI_CONST uint8_t* CBytePtr;
__device__ __forceinline__ DrawPtr const_draws(const DrawDesc* p) { return (DrawPtr)(uintptr_t)p; }
__device__ __forceinline__ CFloatPtr cf(const float* p) { return (CFloatPtr)(uintptr_t)p; }
__device__ __forceinline__ CBytePtr cb(const uint8_t* p) { return (CBytePtr)(uintptr_t)p; }
__device__ __forceinline__ float ldcf(CBytePtr p, uint32_t off) { return *reinterpret_cast<const MIRHI_CONST float*>(p + off); }
__device__ __forceinline__ uint32_t ldcu(CBytePtr p, uint32_t off) { return *reinterpret_cast<const MIRHI_CONST uint32_t*>(p + off); }
typedef const MIRHI_CONST DrawDesc& DrawRef;
typedef const MIRHI_CONST PassParams* ParamsPtr;
typedef const MIRHI_CONST PassParams& ParamsRef;   // scalar (s_load) access to the pass parameters in device memory
// opaque to the optimiser: loads through the result cannot be hoisted above this point
__device__ __forceinline__ ParamsPtr launder_params(ParamsPtr p) { asm volatile("" : "+s"(p)); return p; }

// HLSL mul(M, v), M column-major (vertex/model.hlsl:44,48); accumulation order = oracle's.
__device__ __forceinline__ f4 mat4_mul(CFloatPtr m, f4 v) {
    f4 r;
    r.x = ((m[0] * v.x + m[4] * v.y) + m[8] * v.z) + m[12] * v.w;
    r.y = ((m[1] * v.x + m[5] * v.y) + m[9] * v.z) + m[13] * v.w;
    r.z = ((m[2] * v.x + m[6] * v.y) + m[10] * v.z) + m[14] * v.w;
    r.w = ((m[3] * v.x + m[7] * v.y) + m[11] * v.z) + m[15] * v.w;
    return r;
}
__device__ __forceinline__ f3 mat3_mul(CFloatPtr m, f3 v) {
    f3 r;
    r.x = (m[0] * v.x + m[4] * v.y) + m[8] * v.z;
    r.y = (m[1] * v.x + m[5] * v.y) + m[9] * v.z;
    r.z = (m[2] * v.x + m[6] * v.y) + m[10] * v.z;
    return r;
}
__device__ __forceinline__ float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ f3 add3(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 sub3(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 scale3(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ f3 mul3(f3 a, f3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ f3 cross3(f3 a, f3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ float length3(f3 a) { return sqrtf(dot3(a, a)); }
__device__ __forceinline__ f3 normalize3(f3 a) { const float inv = 1.0f / sqrtf(dot3(a, a)); return {a.x * inv, a.y * inv, a.z * inv}; }
__device__ __forceinline__ float saturatef(float x) { return x > 0.0f ? (x < 1.0f ? x : 1.0f) : 0.0f; }

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
// screen-space triangle record (TriRec, 48 B) and its tile-relative form (TileRec, 64 B, LDS only)
// ------------------------------------------------------------------------------------------------
struct ScreenTri {
    int32_t X[3], Y[3];          // 1/256 px, orientation normalised (interior has E > 0)
    float z0, zx, zy;
    int32_t minx, maxx, miny, maxy;   // inclusive pixel bbox (scissor-clamped)
    uint32_t idk, boxed;
};

__device__ __forceinline__ void store_tri(uint4* dst, const ScreenTri& t) {
    dst[0] = make_uint4((uint32_t)t.X[0], (uint32_t)t.Y[0], (uint32_t)t.X[1], (uint32_t)t.Y[1]);
    dst[1] = make_uint4((uint32_t)t.X[2], (uint32_t)t.Y[2], __float_as_uint(t.z0), __float_as_uint(t.zx));
    dst[2] = make_uint4(__float_as_uint(t.zy), t.idk, (uint32_t)t.minx | ((uint32_t)t.maxx << 16) | (t.boxed << 31),
                        (uint32_t)t.miny | ((uint32_t)t.maxy << 16));
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
        const float iw = 1.0f / c[i].w;
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
    t.zx = (dz1 * fy2 - dz2 * fy1) / area;
    t.zy = (dz2 * fx1 - dz1 * fx2) / area;
    t.z0 = z[0];
    const int32_t xmin = min(t.X[0], min(t.X[1], t.X[2])), xmax = max(t.X[0], max(t.X[1], t.X[2]));
    const int32_t ymin = min(t.Y[0], min(t.Y[1], t.Y[2])), ymax = max(t.Y[0], max(t.Y[1], t.Y[2]));
    int32_t px0 = (xmin + 127) >> 8, px1 = (xmax - 128) >> 8;
    int32_t py0 = (ymin + 127) >> 8, py1 = (ymax - 128) >> 8;
    const bool cut = D.scissor_partial && (px0 < D.sx0 || px1 > D.sx1 || py0 < D.sy0 || py1 > D.sy1);
    px0 = max(px0, D.sx0); px1 = min(px1, D.sx1); py0 = max(py0, D.sy0); py1 = min(py1, D.sy1);
    if (px0 > px1 || py0 > py1) return false;
    // tile rows outside this device's band are not rasterized here (tile-row split)
    const int32_t band0 = (int32_t)P.tile_row_begin * TILE, band1 = (int32_t)P.tile_row_end * TILE - 1;
    if (py1 < band0 || py0 > band1) return false;
    t.minx = px0; t.maxx = px1; t.miny = py0; t.maxy = py1;
    t.boxed = cut ? 1u : 0u;
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
                const float tt = dp / (dp - dq);
                tmp[m++] = {p.x + tt * (q.x - p.x), p.y + tt * (q.y - p.y), p.z + tt * (q.z - p.z), p.w + tt * (q.w - p.w)};
            }
        }
        n = m;
        f4* s = in; in = tmp; tmp = s;
    }
    for (int i = 1; i + 1 < n; i++) {
        const f4 tri[3] = {in[0], in[i], in[i + 1]};
        ScreenTri t;
        if (setup_triangle(P, D, tri, prim, t)) emit_big(P, t);
    }
}

__device__ __forceinline__ uint32_t find_draw(ParamsRef P, uint32_t prim) {
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
__global__ __launch_bounds__(GEOM_THREADS) void vertex_kernel(const PassParams* __restrict__ params) {
    ParamsRef P = *(ParamsPtr)(uintptr_t)params;
    const uint32_t slot0 = blockIdx.x * GEOM_THREADS;
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
    uint4* out = reinterpret_cast<uint4*>(J.out) + (size_t)vidx * J.words;
    out[0] = make_uint4(__float_as_uint(c.x), __float_as_uint(c.y), __float_as_uint(c.z), __float_as_uint(c.w));
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

__device__ __forceinline__ uint32_t pack_bgra8_srgb(f4 c);

// Pair-parallel binning: the wave's (triangle, bin) pairs -- one per triangle for fine meshes, about five for scattered
// 50-pixel triangles, sixteen at most -- are enumerated densely through LDS and dealt to the lanes 64 at a time, one
// record copy per pair.  Every lane stays busy; a lane walking its own <= 16 bins (the first design) issued 2-3x the
// instructions per wave, and sixteen lanes per triangle 10x (measured: C2 907 / 987 / 1089 Mtris/s for walk / wide / pairs).
constexpr uint32_t PAIR_MAX = GEOM_THREADS * MAX_BIN_SPAN * MAX_BIN_SPAN;
__device__ __forceinline__ void bin_triangle_pairs(ParamsRef P, bool valid, const ScreenTri& t, uint4 (*lds_tri)[3],
                                                   uint32_t* lds_meta, uint16_t* lds_owner) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t lt = (1ull << lane) - 1ull;
    int32_t tx0 = 0, ty0 = 0, ntx = 0, nty = 0;
    bool spill = false;
    if (valid) {
        tx0 = t.minx >> TILE_LOG2; ty0 = max(t.miny >> TILE_LOG2, (int32_t)P.tile_row_begin);
        ntx = (t.maxx >> TILE_LOG2) - tx0 + 1;
        nty = min(t.maxy >> TILE_LOG2, (int32_t)P.tile_row_end - 1) - ty0 + 1;
        spill = ntx > MAX_BIN_SPAN || nty > MAX_BIN_SPAN;
    }
    const bool binned = valid && !spill;
    const uint32_t nb = binned ? (uint32_t)(ntx * nty) : 0u;
    // exclusive prefix sum of nb (<= 16) over the wave from five bit planes of ballots
    uint32_t ex = 0, total = 0;
#pragma unroll
    for (uint32_t bit = 0; bit < 5; bit++) {
        const uint64_t m = __ballot(((nb >> bit) & 1u) != 0u);
        ex += (uint32_t)__popcll(m & lt) << bit;
        total += (uint32_t)__popcll(m) << bit;
    }
    if (binned) {
        lds_tri[lane][0] = make_uint4((uint32_t)t.X[0], (uint32_t)t.Y[0], (uint32_t)t.X[1], (uint32_t)t.Y[1]);
        lds_tri[lane][1] = make_uint4((uint32_t)t.X[2], (uint32_t)t.Y[2], __float_as_uint(t.z0), __float_as_uint(t.zx));
        lds_tri[lane][2] = make_uint4(__float_as_uint(t.zy), t.idk, (uint32_t)t.minx | ((uint32_t)t.maxx << 16) | (t.boxed << 31),
                                      (uint32_t)t.miny | ((uint32_t)t.maxy << 16));
        lds_meta[lane] = (uint32_t)(ty0 - (int32_t)P.tile_row_begin) * P.tiles_x + (uint32_t)tx0;
        uint32_t pos = ex;
#pragma unroll
        for (uint32_t k = 0; k < (uint32_t)(MAX_BIN_SPAN * MAX_BIN_SPAN); k++) {
            if ((int32_t)(k % MAX_BIN_SPAN) < ntx && (int32_t)(k / MAX_BIN_SPAN) < nty) lds_owner[pos++] = (uint16_t)(lane | (k << 8));
        }
    }
    __syncthreads();            // one wave per workgroup: orders the LDS writes above before the reads below
    // All returning atomics of the wave are issued before the first result is consumed.  Lanes of a round that target
    // the same tile (mesh order: most of them) are grouped and the group's first lane reserves the whole range with one
    // atomic; grouping stops at the first small group (scattered input would only serialise its atomics).
    constexpr uint32_t ROUNDS = PAIR_MAX / GEOM_THREADS;
    constexpr int GROUP_ROUNDS = 8, GROUP_MIN = 2;
    uint32_t raw[ROUNDS];      // atomic result (held by the reserving lane)
    uint32_t who[ROUNDS];      // reserving lane | rank within its group << 8
#pragma unroll
    for (uint32_t it = 0; it < ROUNDS; it++) {
        raw[it] = 0; who[it] = lane;
        if (it * GEOM_THREADS >= total) break;
        const uint32_t p = it * GEOM_THREADS + lane;
        const bool act = p < total;
        uint32_t tile = 0;
        if (act) {
            const uint32_t o = lds_owner[p], kk = o >> 8;
            tile = lds_meta[o & 0xFFu] + (kk / MAX_BIN_SPAN) * P.tiles_x + (kk % MAX_BIN_SPAN);   // (flag bit not set yet)
        }
        uint64_t rem = __ballot(act);
        for (int round = 0; round < GROUP_ROUNDS && rem; round++) {
            const int leader = __ffsll((long long)rem) - 1;
            const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)tile, leader);
            const uint64_t grp = __ballot(act && tile == t0) & rem;
            if (__popcll(grp) < GROUP_MIN) break;
            if (act && tile == t0 && ((rem >> lane) & 1ull)) who[it] = (uint32_t)leader | ((uint32_t)__popcll(grp & lt) << 8);
            if ((int)lane == leader) raw[it] = atomicAdd(&P.bin_count[t0], (uint32_t)__popcll(grp));
            rem &= ~grp;
        }
        if (act && ((rem >> lane) & 1ull)) raw[it] = atomicAdd(&P.bin_count[tile], 1u);   // ungrouped lanes
    }
#pragma unroll
    for (uint32_t it = 0; it < ROUNDS; it++) {
        if (it * GEOM_THREADS >= total) break;
        const uint32_t p = it * GEOM_THREADS + lane;
        const uint32_t slot = (uint32_t)__shfl((int)raw[it], (int)(who[it] & 0xFFu)) + (who[it] >> 8);
        if (p < total) {
            const uint32_t o = lds_owner[p], ol = o & 0xFFu, kk = o >> 8;
            const uint32_t tile = (lds_meta[ol] & 0x7FFFFFFFu) + (kk / MAX_BIN_SPAN) * P.tiles_x + (kk % MAX_BIN_SPAN);
            if (slot < P.bin_cap) {
                uint4* dst = reinterpret_cast<uint4*>(P.bin_recs) + ((size_t)tile * P.bin_cap + slot) * 3u;
                dst[0] = lds_tri[ol][0]; dst[1] = lds_tri[ol][1]; dst[2] = lds_tri[ol][2];
            } else {
                atomicOr(&lds_meta[ol], 0x80000000u);   // bin full: the owner sends the triangle to the big list, once
            }
        }
    }
    __syncthreads();
    if (binned && (lds_meta[lane] >> 31)) spill = true;    // (idempotent resolve: being in some bins as well is harmless)
    if (valid && spill) emit_big(P, t);
}

// One wave per workgroup; draws are padded to whole waves so the draw (and with it every uniform, pointer
// and pipeline-state word) is wave-uniform and lives in SGPRs.  One lane per triangle up to the screen-space setup,
// then bin_triangle_pairs.
__global__ __launch_bounds__(GEOM_THREADS) void geometry_kernel(const PassParams* __restrict__ params, const GeometryHead H) {
    ParamsRef P = *(ParamsPtr)(uintptr_t)params;
    __shared__ f4 poly[CLIP_BATCH][2][CLIP_MAX_VERTS];   // 2.5 KB: clipping lanes take turns, 8 at a time
    __shared__ uint4 lds_tri[GEOM_THREADS][3];
    __shared__ uint32_t lds_meta[GEOM_THREADS];
    __shared__ uint16_t lds_owner[PAIR_MAX];
    GSTAMP(0);
    const uint32_t slot0 = blockIdx.x * GEOM_THREADS;
    uint32_t lo = 0, hi = H.num_draws;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (const_draws(H.draws)[mid].slot_base <= slot0) lo = mid; else hi = mid;
    }
    DrawRef D = const_draws(H.draws)[lo];
    const uint32_t tri = slot0 - D.slot_base + threadIdx.x;
    const uint32_t prim = D.prim_base + tri;
    bool valid = false;
    uint32_t any = 0;
    ScreenTri t;
    f4 c[3];
    bool dropped = false;
    if (D.program == 3) {
        // pixel/model_pbr.hlsl:174-178 `if (baseColor.a < alphaCutoff) discard;` decided per draw: alpha is
        // baseColorFactor.a, or a texel alpha in [0,1] times it.  A draw whose texels could fall on both sides of
        // the cutoff would need a per-fragment discard before the depth write: reported, not rendered.
        const CBytePtr M = cb(D.material);
        const float fa = ldcf(M, 12), cutoff = ldcf(M, 44);
        float lo = fa, hi = fa;
        if (ldcu(M, 48) != 0u) { lo = fa < 0.0f ? fa : 0.0f; hi = fa > 0.0f ? fa : 0.0f; }
        if (hi < cutoff) dropped = true;
        else if (!(lo >= cutoff)) {
            dropped = true;
            if (threadIdx.x == 0) __hip_atomic_fetch_or(P.status, STATUS_ALPHA_TEST_TEXTURED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (tri < D.tri_count && !dropped) {
#pragma unroll
        for (uint32_t k = 0; k < 3; k++) {
            const uint32_t vidx = fetch_index(D, 3u * tri + k);
            if (D.vs_words) {                      // MODEL programs: clip position from the vertex pre-pass
                const uint4 w0 = reinterpret_cast<const uint4*>(D.vs_out)[(size_t)vidx * D.vs_words];
                c[k] = {__uint_as_float(w0.x), __uint_as_float(w0.y), __uint_as_float(w0.z), __uint_as_float(w0.w)};
            } else {
                c[k] = vs_position(D, vidx, nullptr);
            }
        }
        const uint32_t o0 = outcode_view(c[0]), o1 = outcode_view(c[1]), o2 = outcode_view(c[2]);
        if (!(o0 & o1 & o2)) {
            any = outcode_clip(c[0], D.gx, D.gy) | outcode_clip(c[1], D.gx, D.gy) | outcode_clip(c[2], D.gx, D.gy);
            bool in_band = true;
            if (any == 0 && (P.tile_row_begin != 0u || P.tile_row_end != P.tiles_y)) {
                // tile-row split (one band per GPU): a triangle whose three vertices lie above the band, or below it, is
                // dropped before the setup arithmetic -- every rank sees all triangles, most belong to other bands.
                // Clip-space test with a one-pixel margin for the snap: ys = (y/w)*hh + cy, w > 0 here.
                const float top = D.cy - ((float)(P.tile_row_begin * TILE) - 1.0f), bot = D.cy - ((float)(P.tile_row_end * TILE) + 1.0f);
                const bool above = c[0].y * D.hh + top * c[0].w < 0.0f && c[1].y * D.hh + top * c[1].w < 0.0f && c[2].y * D.hh + top * c[2].w < 0.0f;
                const bool below = c[0].y * D.hh + bot * c[0].w > 0.0f && c[1].y * D.hh + bot * c[1].w > 0.0f && c[2].y * D.hh + bot * c[2].w > 0.0f;
                in_band = !(above || below) || !(c[0].w > 0.0f && c[1].w > 0.0f && c[2].w > 0.0f);
            }
            if (any == 0 && in_band) valid = setup_triangle(P, D, c, prim, t);
        }
    }
    if (P.flat_color && D.program == 0 && (valid || any)) {
        // flat-shaded triangle (all three vertex colours equal): shade it once here instead of once per pixel
        const uint8_t* v0 = D.vb + (size_t)fetch_index(D, 3u * tri) * D.stride;
        const uint8_t* v1 = D.vb + (size_t)fetch_index(D, 3u * tri + 1u) * D.stride;
        const uint8_t* v2 = D.vb + (size_t)fetch_index(D, 3u * tri + 2u) * D.stride;
        const uint32_t r = ldu(v0, 12), g = ldu(v0, 16), b = ldu(v0, 20);
        const bool flat = r == ldu(v1, 12) && r == ldu(v2, 12) && g == ldu(v1, 16) && g == ldu(v2, 16) && b == ldu(v1, 20) && b == ldu(v2, 20);
        P.flat_color[prim] = flat ? pack_bgra8_srgb({__uint_as_float(r), __uint_as_float(g), __uint_as_float(b), 1.0f}) : 0u;
    }
    GSTAMP(1);
    bin_triangle_pairs(P, valid, t, lds_tri, lds_meta, lds_owner);
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

// ------------------------------------------------------------------------------------------------
// a8: fragment programs.  Colour is tolerance-checked (|dRGB| < 1e-4 vs the oracle), not bit-exact, so
// this part may contract to FMA and use the 1-ulp hardware rcp / rsq / exp2 / log2.
// ------------------------------------------------------------------------------------------------
#pragma clang fp contract(fast)

__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float frsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
// pow(x, y) for x >= 0 as exp2(y * log2(x)) (HLSL pow lowering); pow(0, y>0) = 0
// log2 near 1 comes from the series of ln(1+t) (t = x-1 is exact there): the hardware v_log_f32 has an absolute
// error of ~2^-22 around 1, which a Blinn-Phong exponent of up to 2048 would amplify past the 1e-4 colour bound.
__device__ __forceinline__ float flog2(float x) {
    const float t = x - 1.0f;
    const float p = t * (1.0f + t * (-0.5f + t * (0.33333334f + t * (-0.25f + t * 0.2f))));
    return fabsf(t) < 0.015625f ? p * 1.44269504089f : __builtin_amdgcn_logf(x);
}
__device__ __forceinline__ float fpow(float x, float y) { return __builtin_amdgcn_exp2f(y * flog2(x)); }
__device__ __forceinline__ f3 fnormalize3(f3 a) { const float r = frsq(dot3(a, a)); return {a.x * r, a.y * r, a.z * r}; }
#pragma clang fp contract(off)

__device__ __forceinline__ float attenuation(float distance, float radius) {          // lights.hlsli:63-73
    const float att = 1.0f / (distance * distance + 1.0f);
    float falloff = saturatef(1.0f - distance / radius);
    falloff = falloff * falloff;
    return att * falloff;
}
__device__ __forceinline__ float roughness_to_shininess(float roughness) {             // lights.hlsli:152-159
    const float r = roughness < 0.0f ? 0.0f : (roughness > 1.0f ? 1.0f : roughness);
    return 2048.0f + (2.0f - 2048.0f) * r;
}
__device__ __forceinline__ f3 blinn_phong(f3 L, f3 V, f3 N, f3 lightColor, f3 albedo, float shininess) {  // :95-117
    float NdotL = dot3(N, L);
    if (!(NdotL > 0.0f)) NdotL = 0.0f;
    const f3 diffuse = mul3(scale3(lightColor, NdotL), albedo);
    if (NdotL <= 0.0f) return diffuse;
    const f3 H = normalize3(add3(L, V));
    float NdotH = dot3(N, H);
    if (!(NdotH > 0.0f)) NdotH = 0.0f;
    const float sp = fpow(NdotH, shininess);
    return add3(diffuse, scale3(lightColor, sp));
}

__device__ __forceinline__ f4 unpack_rgba8(uint32_t p) {
    const float s = 1.0f / 255.0f;
    return {(float)(p & 0xFF) * s, (float)((p >> 8) & 0xFF) * s, (float)((p >> 16) & 0xFF) * s, (float)(p >> 24) * s};
}
// repeat addressing of one coordinate: c mod n into [0, n); a mask when n is a power of two (the usual case),
// one division otherwise.  The +1 neighbour wraps by comparison, so a bilinear tap costs two of these, not eight.
__device__ __forceinline__ int32_t wrap_coord(int32_t c, int32_t n) {
    if ((n & (n - 1)) == 0) return c & (n - 1);          // n is wave-uniform: a scalar branch
    c %= n;
    return c < 0 ? c + n : c;
}
// bilinear, repeat, no mips (see oracle sample_bilinear)
__device__ __forceinline__ f4 sample_bilinear(const uint8_t* tex, uint32_t w, uint32_t h, float u, float v) {
    if (!tex || w == 0 || h == 0) return {1.0f, 1.0f, 1.0f, 1.0f};
    const uint32_t* texels = reinterpret_cast<const uint32_t*>(tex);
    if (w == 1 && h == 1) return unpack_rgba8(texels[0]);
    const float fx = u * (float)w - 0.5f, fy = v * (float)h - 0.5f;
    const float x0f = floorf(fx), y0f = floorf(fy);
    const float ax = fx - x0f, ay = fy - y0f;
    const int32_t x0 = wrap_coord((int32_t)x0f, (int32_t)w), y0 = wrap_coord((int32_t)y0f, (int32_t)h);
    const int32_t x1 = x0 + 1 == (int32_t)w ? 0 : x0 + 1, y1 = y0 + 1 == (int32_t)h ? 0 : y0 + 1;
    const uint32_t r0 = (uint32_t)y0 * w, r1 = (uint32_t)y1 * w;
    const f4 c00 = unpack_rgba8(texels[r0 + (uint32_t)x0]), c10 = unpack_rgba8(texels[r0 + (uint32_t)x1]);
    const f4 c01 = unpack_rgba8(texels[r1 + (uint32_t)x0]), c11 = unpack_rgba8(texels[r1 + (uint32_t)x1]);
    f4 r;
#define MIRHI_LERP2(f) { const float top = c00.f + (c10.f - c00.f) * ax; const float bot = c01.f + (c11.f - c01.f) * ax; r.f = top + (bot - top) * ay; }
    MIRHI_LERP2(x) MIRHI_LERP2(y) MIRHI_LERP2(z) MIRHI_LERP2(w)
#undef MIRHI_LERP2
    return r;
}

#pragma clang fp contract(off)
// Everything that feeds pow(NdotH, shininess) must match the oracle bit for bit: an exponent of up to 2048
// turns a 1-ulp difference in NdotH into a 1e-4 relative difference of the specular term.
struct Varyings { f3 world, normal, tangent, bitangent; float u, v; };

__device__ __forceinline__ f3 interp3(const float b[3], f3 a0, f3 a1, f3 a2) {
    return {(b[0] * a0.x + b[1] * a1.x) + b[2] * a2.x, (b[0] * a0.y + b[1] * a1.y) + b[2] * a2.y,
            (b[0] * a0.z + b[1] * a1.z) + b[2] * a2.z};
}

// perspective-correct barycentrics of the pixel centre from the original clip-space triangle
// (2-D homogeneous form relative to the pixel: valid for w <= 0 vertices, no clipped attributes needed)
template <bool FAST>
__device__ __forceinline__ void barycentrics(DrawRef D, const f4 c[3], float pxc, float pyc, float b[3]) {
    float ax[3], ay[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        ax[k] = (c[k].x * D.hw + c[k].w * D.cx) - pxc * c[k].w;
        ay[k] = (c[k].y * D.hh + c[k].w * D.cy) - pyc * c[k].w;
    }
    const float l0 = ax[1] * ay[2] - ax[2] * ay[1];
    const float l1 = ax[2] * ay[0] - ax[0] * ay[2];
    const float l2 = ax[0] * ay[1] - ax[1] * ay[0];
    const float inv = FAST ? __builtin_amdgcn_rcpf((l0 + l1) + l2) : 1.0f / ((l0 + l1) + l2);
    b[0] = l0 * inv; b[1] = l1 * inv; b[2] = l2 * inv;
}

#pragma clang fp contract(fast)
// vertex/triangle.hlsl + pixel/triangle.hlsl: clip = (pos, 1), colour pass-through.  No pow downstream, so FMA
// contraction and the 1-ulp rcp stay ~1e-7 from the oracle (bound 1e-4).  With w = 1 the homogeneous
// barycentrics reduce to ax_k = x_k * W/2 + (cx - px).
__device__ __forceinline__ f4 shade_triangle_program(DrawRef D, uint32_t tri, float pxc, float pyc) {
    float ax[3], ay[3]; f3 col[3];
    const float tx = D.cx - pxc, ty = D.cy - pyc;
#pragma unroll
    for (uint32_t k = 0; k < 3; k++) {
        const uint32_t vidx = fetch_index(D, 3u * tri + k);
        const uint8_t* v = D.vb + (size_t)vidx * D.stride;
        ax[k] = ldf(v, 0) * D.hw + tx;                                       // vertex/triangle.hlsl:19-20
        ay[k] = ldf(v, 4) * D.hh + ty;
        col[k] = {ldf(v, 12), ldf(v, 16), ldf(v, 20)};
    }
    const float l0 = ax[1] * ay[2] - ax[2] * ay[1];
    const float l1 = ax[2] * ay[0] - ax[0] * ay[2];
    const float l2 = ax[0] * ay[1] - ax[1] * ay[0];
    const float inv = __builtin_amdgcn_rcpf((l0 + l1) + l2);
    const float b[3] = {l0 * inv, l1 * inv, l2 * inv};
    const f3 o = interp3(b, col[0], col[1], col[2]);                         // pixel/triangle.hlsl:10-13
    return {o.x, o.y, o.z, 1.0f};
}
#pragma clang fp contract(off)

// a8: Cook-Torrance GGX, shaders/hlsl/pbr.hlsli (shadow pass not on the path: shadow = 1)
#define PBR_PI 3.14159265358979323846f
#define PBR_EPSILON 0.0001f
__device__ __forceinline__ float max0(float x) { return x > 0.0f ? x : 0.0f; }
__device__ __forceinline__ float distribution_ggx(float NdotH, float roughness) {       // pbr.hlsli:55-69
    const float a = roughness * roughness, a2 = a * a;
    const float NdotH2 = NdotH * NdotH;
    float denom = NdotH2 * (a2 - 1.0f) + 1.0f;
    denom = (PBR_PI * denom) * denom;
    return a2 / (denom > PBR_EPSILON ? denom : PBR_EPSILON);
}
__device__ __forceinline__ float geometry_schlick_ggx(float NdotV, float roughness) {   // pbr.hlsli:83-93
    const float r = roughness + 1.0f;
    const float k = (r * r) / 8.0f;
    const float denom = NdotV * (1.0f - k) + k;
    return NdotV / (denom > PBR_EPSILON ? denom : PBR_EPSILON);
}
struct PbrMaterial { f3 albedo; float metallic, roughness; };
__device__ __forceinline__ f3 pbr_direct(f3 N, f3 V, f3 L, f3 radiance, const PbrMaterial& m) {   // pbr.hlsli:292-333
    const f3 H = normalize3(add3(V, L));
    const f3 F0 = {0.04f + (m.albedo.x - 0.04f) * m.metallic, 0.04f + (m.albedo.y - 0.04f) * m.metallic,
                   0.04f + (m.albedo.z - 0.04f) * m.metallic};
    const float NDF = distribution_ggx(max0(dot3(N, H)), m.roughness);
    const float NdotV = max0(dot3(N, V)), NdotL = max0(dot3(N, L));
    const float G = geometry_schlick_ggx(NdotV, m.roughness) * geometry_schlick_ggx(NdotL, m.roughness);
    const float ct = saturatef(max0(dot3(H, V)));
    const float p5 = fpow(1.0f - ct, 5.0f);                                             // FresnelSchlick :131-136
    const f3 F = {F0.x + (1.0f - F0.x) * p5, F0.y + (1.0f - F0.y) * p5, F0.z + (1.0f - F0.z) * p5};
    const float om = 1.0f - m.metallic;
    const f3 kD = {(1.0f - F.x) * om, (1.0f - F.y) * om, (1.0f - F.z) * om};
    const float ndg = NDF * G;
    const float denominator = (4.0f * NdotV) * NdotL + PBR_EPSILON;
    const f3 specular = {(ndg * F.x) / denominator, (ndg * F.y) / denominator, (ndg * F.z) / denominator};
    return {(((kD.x * m.albedo.x) / PBR_PI + specular.x) * radiance.x) * NdotL,
            (((kD.y * m.albedo.y) / PBR_PI + specular.y) * radiance.y) * NdotL,
            (((kD.z * m.albedo.z) / PBR_PI + specular.z) * radiance.z) * NdotL};
}

// pixel/model_pbr.hlsl:159-320 after the shared varying interpolation
__device__ __forceinline__ f4 shade_pbr(DrawRef D, const float b[3], const Varyings vv[3], f3 worldPos, f3 V, f3 N) {
    const CBytePtr M = cb(D.material);                                                  // MaterialData :36-59 (80 B)
    const float u = (b[0] * vv[0].u + b[1] * vv[1].u) + b[2] * vv[2].u;
    const float v = (b[0] * vv[0].v + b[1] * vv[1].v) + b[2] * vv[2].v;
    f4 baseColor = {ldcf(M, 0), ldcf(M, 4), ldcf(M, 8), ldcf(M, 12)};
    float metallic = ldcf(M, 16), roughness = ldcf(M, 20), ao = ldcf(M, 24);
    const float normalScale = ldcf(M, 28);
    f3 emissive = {ldcf(M, 32), ldcf(M, 36), ldcf(M, 40)};
    if (ldcu(M, 48) != 0u) {
        const f4 t = sample_bilinear(D.tex[0], D.tex_w[0], D.tex_h[0], u, v);
        baseColor = {t.x * baseColor.x, t.y * baseColor.y, t.z * baseColor.z, t.w * baseColor.w};
    }
    if (ldcu(M, 56) != 0u) {
        const f4 t = sample_bilinear(D.tex[2], D.tex_w[2], D.tex_h[2], u, v);
        roughness = roughness * t.y; metallic = metallic * t.z;
    }
    if (ldcu(M, 60) != 0u) ao = ao * sample_bilinear(D.tex[3], D.tex_w[3], D.tex_h[3], u, v).x;
    if (ldcu(M, 64) != 0u) {
        const f4 t = sample_bilinear(D.tex[4], D.tex_w[4], D.tex_h[4], u, v);
        emissive = {emissive.x * t.x, emissive.y * t.y, emissive.z * t.z};
    }
    if (ldcu(M, 52) != 0u) {                                                            // GetWorldNormal :124-151
        const f4 nc = sample_bilinear(D.tex[1], D.tex_w[1], D.tex_h[1], u, v);
        const f3 ncm1 = {nc.x - 1.0f, nc.y - 1.0f, nc.z - 1.0f};
        if (!(length3(ncm1) < 0.01f)) {
            const f3 ns = normalize3({(nc.x * 2.0f - 1.0f) * normalScale, (nc.y * 2.0f - 1.0f) * normalScale, nc.z * 2.0f - 1.0f});
            const f3 T = normalize3(interp3(b, vv[0].tangent, vv[1].tangent, vv[2].tangent));
            const f3 Bt = normalize3(interp3(b, vv[0].bitangent, vv[1].bitangent, vv[2].bitangent));
            N = normalize3(add3(add3(scale3(T, ns.x), scale3(Bt, ns.y)), scale3(N, ns.z)));
        }
    }
    PbrMaterial m;
    m.albedo = {baseColor.x, baseColor.y, baseColor.z};
    m.metallic = metallic;
    m.roughness = roughness > 0.04f ? roughness : 0.04f;                                // ClampRoughness :476-479
    f3 lighting = {0.0f, 0.0f, 0.0f};
    {
        const f3 dir = {ldcf(cb(D.lights), 0), ldcf(cb(D.lights), 4), ldcf(cb(D.lights), 8)};
        const float intensity = ldcf(cb(D.lights), 12);
        const f3 color = {ldcf(cb(D.lights), 16), ldcf(cb(D.lights), 20), ldcf(cb(D.lights), 24)};
        lighting = add3(lighting, pbr_direct(N, V, normalize3({-dir.x, -dir.y, -dir.z}), scale3(color, intensity), m));
    }
    const uint32_t numPoint = D.point_lights ? ldcu(cb(D.lights), 32) : 0u;
    const uint32_t numSpot = D.spot_lights ? ldcu(cb(D.lights), 36) : 0u;
    for (uint32_t i = 0; i < numPoint; i++) {
        const CBytePtr Lp = cb(D.point_lights) + 32u * i;
        const f3 pos = {ldcf(Lp, 0), ldcf(Lp, 4), ldcf(Lp, 8)};
        const float radius = ldcf(Lp, 12);
        const f3 color = {ldcf(Lp, 16), ldcf(Lp, 20), ldcf(Lp, 24)};
        const float intensity = ldcf(Lp, 28);
        const f3 lv = sub3(pos, worldPos);
        const float dist = length3(lv);
        const f3 L = scale3(lv, 1.0f / dist);
        lighting = add3(lighting, pbr_direct(N, V, L, scale3(scale3(color, intensity), attenuation(dist, radius)), m));
    }
    for (uint32_t j = 0; j < numSpot; j++) {
        const CBytePtr Ls = cb(D.spot_lights) + 48u * j;
        const f3 pos = {ldcf(Ls, 0), ldcf(Ls, 4), ldcf(Ls, 8)};
        const float innerCos = ldcf(Ls, 12);
        const f3 sdir = {ldcf(Ls, 16), ldcf(Ls, 20), ldcf(Ls, 24)};
        const float outerCos = ldcf(Ls, 28);
        const f3 color = {ldcf(Ls, 32), ldcf(Ls, 36), ldcf(Ls, 40)};
        const float intensity = ldcf(Ls, 44);
        const f3 lv = sub3(pos, worldPos);
        const float dist = length3(lv);
        const f3 L = scale3(lv, 1.0f / dist);
        const float datt = attenuation(dist, 50.0f);
        const f3 sd = normalize3(sdir);
        const float cosAngle = dot3({-L.x, -L.y, -L.z}, sd);
        const float satt = saturatef((cosAngle - outerCos) / (innerCos - outerCos));
        lighting = add3(lighting, pbr_direct(N, V, L, scale3(scale3(scale3(color, intensity), datt), satt), m));
    }
    const float up = N.y * 0.5f + 0.5f;                                                 // CalculateHemisphereAmbient pbr.hlsli:483-492
    const f3 amb = {0.08f + (0.15f - 0.08f) * up, 0.06f + (0.18f - 0.06f) * up, 0.04f + (0.25f - 0.04f) * up};
    const float om = 1.0f - m.metallic;
    const f3 ambient = scale3(scale3(mul3(amb, m.albedo), ao), om);
    lighting = scale3(lighting, 1.0f + (ao - 1.0f) * 0.5f);                             // lerp(1, ao, 0.5) :311
    const f3 col = add3(add3(ambient, lighting), emissive);
    return {col.x, col.y, col.z, baseColor.w};
}

template <bool PBR>
__device__ __forceinline__ f4 shade_model_program(DrawRef D, uint32_t tri, float pxc, float pyc) {
    f4 c[3]; Varyings vv[3];
    const bool full = D.program >= 2;
#pragma unroll
    for (uint32_t k = 0; k < 3; k++) {
        // vertex/model.hlsl outputs, computed once per vertex by vertex_kernel
        const uint4* sv = reinterpret_cast<const uint4*>(D.vs_out) + (size_t)fetch_index(D, 3u * tri + k) * D.vs_words;
        const uint4 w0 = sv[0], w1 = sv[1], w2 = sv[2];
        c[k] = {__uint_as_float(w0.x), __uint_as_float(w0.y), __uint_as_float(w0.z), __uint_as_float(w0.w)};
        vv[k].world = {__uint_as_float(w1.x), __uint_as_float(w1.y), __uint_as_float(w1.z)};
        vv[k].normal = {__uint_as_float(w1.w), __uint_as_float(w2.x), __uint_as_float(w2.y)};
        if (full) {
            const uint4 w3 = sv[3], w4 = sv[4];
            vv[k].u = __uint_as_float(w2.z); vv[k].v = __uint_as_float(w2.w);
            vv[k].tangent = {__uint_as_float(w3.x), __uint_as_float(w3.y), __uint_as_float(w3.z)};
            vv[k].bitangent = {__uint_as_float(w3.w), __uint_as_float(w4.x), __uint_as_float(w4.y)};
        }
    }
    float b[3];
    barycentrics<false>(D, c, pxc, pyc, b);
    const f3 worldPos = interp3(b, vv[0].world, vv[1].world, vv[2].world);
    const f3 Nv = interp3(b, vv[0].normal, vv[1].normal, vv[2].normal);
    const CFloatPtr cam = cf(D.camera);
    const f3 camPos = {cam[48], cam[49], cam[50]};          // cameraPosition @192 B
    const f3 V = normalize3(sub3(camPos, worldPos));
    f3 N = normalize3(Nv);

    if (!full) {                                                             // pixel/model.hlsl:29-82
        const f3 albedo = {0.7f, 0.7f, 0.7f};
        const f3 one = {1.0f, 1.0f, 1.0f};
        const f3 L = normalize3(one);
        const f3 ambient = scale3(scale3(albedo, 0.03f), 1.0f);
        const f3 lighting = blinn_phong(L, V, N, one, albedo, roughness_to_shininess(0.5f));
        const f3 col = add3(ambient, lighting);
        return {col.x, col.y, col.z, 1.0f};
    }
    if (PBR && D.program == 3) return shade_pbr(D, b, vv, worldPos, V, N);
    // pixel/model_full.hlsl:85-150
    const float u = (b[0] * vv[0].u + b[1] * vv[1].u) + b[2] * vv[2].u;
    const float v = (b[0] * vv[0].v + b[1] * vv[1].v) + b[2] * vv[2].v;
    const f4 baseColor = {ldcf(cb(D.material), 0), ldcf(cb(D.material), 4), ldcf(cb(D.material), 8), ldcf(cb(D.material), 12)};
    const float roughness = ldcf(cb(D.material), 20), ao = ldcf(cb(D.material), 24);
    const f4 albedoSample = sample_bilinear(D.tex[0], D.tex_w[0], D.tex_h[0], u, v);
    const f3 albedo = {albedoSample.x * baseColor.x, albedoSample.y * baseColor.y, albedoSample.z * baseColor.z};
    const f4 nc = sample_bilinear(D.tex[1], D.tex_w[1], D.tex_h[1], u, v);
    const f3 ncm1 = {nc.x - 1.0f, nc.y - 1.0f, nc.z - 1.0f};
    const bool hasNormalMap = length3(ncm1) > 0.01f;                          // :94-95
    if (hasNormalMap) {                                                      // GetWorldNormal :63-83
        const f3 ns = {nc.x * 2.0f - 1.0f, nc.y * 2.0f - 1.0f, nc.z * 2.0f - 1.0f};
        const f3 T = normalize3(interp3(b, vv[0].tangent, vv[1].tangent, vv[2].tangent));
        const f3 Bt = normalize3(interp3(b, vv[0].bitangent, vv[1].bitangent, vv[2].bitangent));
        N = normalize3(add3(add3(scale3(T, ns.x), scale3(Bt, ns.y)), scale3(N, ns.z)));
    }
    const f3 ambient = scale3(scale3(albedo, 0.03f), ao);
    f3 lighting = {0.0f, 0.0f, 0.0f};
    const float shininess = roughness_to_shininess(roughness);
    {   // CalculateDirectionalLight lights.hlsli:166-179 (HLSL DirectionalLight layout :17-23)
        const f3 dir = {ldcf(cb(D.lights), 0), ldcf(cb(D.lights), 4), ldcf(cb(D.lights), 8)};
        const float intensity = ldcf(cb(D.lights), 12);
        const f3 color = {ldcf(cb(D.lights), 16), ldcf(cb(D.lights), 20), ldcf(cb(D.lights), 24)};
        const f3 L = normalize3({-dir.x, -dir.y, -dir.z});
        lighting = add3(lighting, blinn_phong(L, V, N, scale3(color, intensity), albedo, shininess));
    }
    const uint32_t numPoint = D.point_lights ? ldcu(cb(D.lights), 32) : 0u;
    const uint32_t numSpot = D.spot_lights ? ldcu(cb(D.lights), 36) : 0u;
    for (uint32_t i = 0; i < numPoint; i++) {                                // CalculatePointLight :182-199
        const CBytePtr Lp = cb(D.point_lights) + 32u * i;
        const f3 pos = {ldcf(Lp, 0), ldcf(Lp, 4), ldcf(Lp, 8)};
        const float radius = ldcf(Lp, 12);
        const f3 color = {ldcf(Lp, 16), ldcf(Lp, 20), ldcf(Lp, 24)};
        const float intensity = ldcf(Lp, 28);
        const f3 lv = sub3(pos, worldPos);
        const float dist = length3(lv);
        const f3 L = scale3(lv, 1.0f / dist);
        const f3 lc = scale3(scale3(color, intensity), attenuation(dist, radius));
        lighting = add3(lighting, blinn_phong(L, V, N, lc, albedo, shininess));
    }
    for (uint32_t j = 0; j < numSpot; j++) {                                 // CalculateSpotLight :202-231
        const CBytePtr Ls = cb(D.spot_lights) + 48u * j;
        const f3 pos = {ldcf(Ls, 0), ldcf(Ls, 4), ldcf(Ls, 8)};
        const float innerCos = ldcf(Ls, 12);
        const f3 sdir = {ldcf(Ls, 16), ldcf(Ls, 20), ldcf(Ls, 24)};
        const float outerCos = ldcf(Ls, 28);
        const f3 color = {ldcf(Ls, 32), ldcf(Ls, 36), ldcf(Ls, 40)};
        const float intensity = ldcf(Ls, 44);
        const f3 lv = sub3(pos, worldPos);
        const float dist = length3(lv);
        const f3 L = scale3(lv, 1.0f / dist);
        const float datt = attenuation(dist, 50.0f);
        const f3 sd = normalize3(sdir);
        const float cosAngle = dot3({-L.x, -L.y, -L.z}, sd);                // CalculateSpotAttenuation :77-81
        const float satt = saturatef((cosAngle - outerCos) / (innerCos - outerCos));
        const f3 lc = scale3(scale3(scale3(color, intensity), datt), satt);
        lighting = add3(lighting, blinn_phong(L, V, N, lc, albedo, shininess));
    }
    const f3 col = add3(ambient, lighting);
    return {col.x, col.y, col.z, albedoSample.w * baseColor.w};
}

#pragma clang fp contract(fast)
// a9: sRGB OETF + UNORM8, BGRA byte order (swapchain.rs:561-570)
__device__ __forceinline__ uint32_t srgb8(float c) {
    c = saturatef(c);
    float e = (c <= 0.0031308f) ? 12.92f * c : 1.055f * __builtin_amdgcn_exp2f((1.0f / 2.4f) * __builtin_amdgcn_logf(c)) - 0.055f;
    e = saturatef(e);
    return (uint32_t)rintf(e * 255.0f);
}
__device__ __forceinline__ uint32_t pack_bgra8_srgb(f4 c) {
    return srgb8(c.z) | (srgb8(c.y) << 8) | (srgb8(c.x) << 16) | ((uint32_t)rintf(saturatef(c.w) * 255.0f) << 24);
}

#pragma clang fp contract(off)

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

// TriRec (screen space) -> TileRec for tile (tx, ty); false if no 8x8 block of the tile can be touched.
//   w0 = { Q0, Q1, Q2, A0 }   Q_i = floor((E_i(tile origin pixel centre) + bias_i) / 256), clamped to +-2^30
//   w1 = { A1, A2, B0, B1 }   A_i = Ya - Yb, B_i = Xb - Xa in 1/256 px (|.| < 2^23, fits v_mad_i32_i24)
//   w2 = { B2, dxt, dyt, z0 } (tile origin pixel centre) - (snapped vertex 0), in pixels (exact), vertex-0 depth
//   w3 = { zx, zy, idk, mask } mask bits 0..15 = 8x8 blocks the triangle may touch, bit 31 = pixel box applies
__device__ __forceinline__ bool make_tile_rec(uint4 out[4], uint32_t& box, const uint4 w0, const uint4 w1, const uint4 w2,
                                              int32_t tx, int32_t ty) {
    const int32_t X[3] = {(int32_t)w0.x, (int32_t)w0.z, (int32_t)w1.x}, Y[3] = {(int32_t)w0.y, (int32_t)w0.w, (int32_t)w1.y};
    const int32_t ox = tx * TILE, oy = ty * TILE;
    const int32_t Ptx = 256 * ox + 128, Pty = 256 * oy + 128;           // tile origin pixel centre, 1/256 px
    int32_t A[3], B[3], Q[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int a = i, b = (i + 1) % 3;
        const int32_t dx = X[b] - X[a], dy = Y[b] - Y[a];
        A[i] = -dy; B[i] = dx;
        const bool topleft = (dy < 0) || (dy == 0 && dx > 0);              // top-left fill rule
        // E_i + bias at the tile origin; every factor fits 24 bits (|X|,|Y| < 2^22, |Pt| < 2^21)
        const int64_t e0 = mul24x24(A[i], Ptx - X[a]) + mul24x24(B[i], Pty - Y[a]) + (topleft ? 0 : -1);
        int64_t q = e0 >> 8;                                 // floor(E/256): E = 256*(q + A*ix + B*iy) + r, 0 <= r < 256
        q = q > (1 << 30) ? (1 << 30) : (q < -(1 << 30) ? -(1 << 30) : q);
        Q[i] = (int32_t)q;
    }
    int32_t bx0 = (int32_t)(w2.z & 0x7FFFu) - ox, bx1 = (int32_t)((w2.z >> 16) & 0x7FFFu) - ox;
    int32_t by0 = (int32_t)(w2.w & 0xFFFFu) - oy, by1 = (int32_t)(w2.w >> 16) - oy;
    bx0 = bx0 < 0 ? 0 : bx0; by0 = by0 < 0 ? 0 : by0;
    bx1 = bx1 > TILE - 1 ? TILE - 1 : bx1; by1 = by1 > TILE - 1 ? TILE - 1 : by1;
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
    const float dxt = ((float)ox + 0.5f) - (float)X[0] * inv256;         // exact: multiples of 2^-8 below 2^15
    const float dyt = ((float)oy + 0.5f) - (float)Y[0] * inv256;
    out[0] = make_uint4((uint32_t)Q[0], (uint32_t)Q[1], (uint32_t)Q[2], (uint32_t)A[0]);
    out[1] = make_uint4((uint32_t)A[1], (uint32_t)A[2], (uint32_t)B[0], (uint32_t)B[1]);
    out[2] = make_uint4((uint32_t)B[2], __float_as_uint(dxt), __float_as_uint(dyt), w1.z);
    out[3] = make_uint4(w1.w, w2.x, w2.y, mask | (w2.z & 0x80000000u));
    box = (uint32_t)bx0 | ((uint32_t)bx1 << 8) | ((uint32_t)by0 << 16) | ((uint32_t)by1 << 24);
    return mask != 0;
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

template <int KEYED, bool BOXED>
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
    for (int b = 0; b < 4; b++) {
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
                upd = inside && pass && idk < st.idk[b];
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

// all records of an LDS chunk: per 64 records one ballot builds the bitmap of records that touch this
// wave's quadrant; LDS latency of the broadcast record reads is hidden by the other waves of the SIMD
template <int KEYED, int TP>
__device__ __forceinline__ void raster_chunk(const uint4* lds_rec, const uint32_t* lds_box, uint32_t n, uint32_t qmask,
                                             int32_t ix0, int32_t iy0, float fix0, float fiy0, ParamsRef P,
                                             PixelState& st, uint32_t qbit0, uint32_t lane) {
    for (uint32_t g = 0; g < n; g += 64u) {
        const uint32_t j = g + lane;
        const uint32_t mymask = j < n ? lds_rec[j * 4u + 3u].w : 0u;
        const bool rel = (mymask & qmask) != 0u, boxed = !TP && (mymask & 0x80000000u) != 0u;
        uint64_t bits = __ballot(rel && !boxed);
        while (bits) {
            const uint32_t cur_j = g + (uint32_t)(__ffsll((long long)bits) - 1);
            bits &= bits - 1;
            const RecRegs cur = load_rec(lds_rec, cur_j);
            raster_record<KEYED, false>(cur, 0u, ix0, iy0, fix0, fiy0, P, st, qbit0);
        }
        if (!TP) {   // without the triangle-parallel path, scissor-cut triangles (rare) take the per-pixel box test here
            uint64_t bbits = __ballot(rel && boxed);
            while (bbits) {
                const uint32_t cur_j = g + (uint32_t)(__ffsll((long long)bbits) - 1);
                bbits &= bbits - 1;
                const RecRegs cur = load_rec(lds_rec, cur_j);
                raster_record<KEYED, true>(cur, lds_box[cur_j], ix0, iy0, fix0, fiy0, P, st, qbit0);
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
    for (int32_t iy = by0; iy <= by1; iy++) {
        int32_t s0 = r0, s1 = r1, s2 = r2;
        const float dy = (float)iy + dyt;
        for (int32_t ix = bx0; ix <= bx1; ix++) {
            if ((s0 | s1 | s2) >= 0) {
                const float dx = (float)ix + dxt;
                const float z = __builtin_fmaf(dy, zy, __builtin_fmaf(dx, zx, z0));
                uint32_t zk = __float_as_uint(__builtin_amdgcn_fmed3f(z, 0.0f, 1.0f)) & 0x7FFFFFFFu;
                if (KEYED == 1) zk = (zk ^ P.zflip) & P.zmask;
                atomicMin(&lds_key[iy * TILE + ix], ((unsigned long long)zk << 32) | idk);
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
template <int KEYED, int TP>
__device__ __forceinline__ void raster_list(const uint4* __restrict__ list, uint32_t n_total, uint4* lds_rec, uint32_t* lds_box,
                                            uint32_t* lds_count, unsigned long long* lds_key, uint32_t tx, uint32_t ty,
                                            uint32_t qmask, int32_t ix0, int32_t iy0, float fix0, float fiy0,
                                            ParamsRef P, PixelState& st, uint32_t qbit0, uint32_t tid,
                                            uint32_t lane) {
    const int32_t tpx0 = (int32_t)(tx * TILE), tpy0 = (int32_t)(ty * TILE);
    // Most bins hold fewer than 64 records, so one wave builds all tile records of a tile.  Which wave does it rotates
    // with the tile: wave k of every workgroup sits on the same SIMD, and a fixed choice would load that SIMD alone.
    const uint32_t ftid = (tid + 64u * ((tx + ty) & 3u)) & (RASTER_THREADS - 1u);
    for (uint32_t base = 0; base < n_total; base += RASTER_CHUNK) {
        if (tid == 0) *lds_count = 0;
        __syncthreads();
        // opaque copies: what make_tile_rec derives from the tile coordinates is rebuilt per chunk (a few instructions)
        // instead of being hoisted out of the loops into VGPRs that then spill
        uint32_t txl = tx, tyl = ty;
        asm volatile("" : "+s"(txl), "+s"(tyl));
        const uint32_t i = base + ftid;
        bool hit = false;
        uint4 rec[4]; uint32_t box = 0;
        if (ftid < RASTER_CHUNK && i < n_total) {
            // all three words are requested together: one memory round trip, not two (a bin holds only records whose
            // box overlaps the tile, so the box test below almost never saves the first two loads)
            const uint4 w0 = list[(size_t)i * 3u], w1 = list[(size_t)i * 3u + 1u], w2 = list[(size_t)i * 3u + 2u];
            const int32_t minx = (int32_t)(w2.z & 0x7FFFu), maxx = (int32_t)((w2.z >> 16) & 0x7FFFu);
            const int32_t miny = (int32_t)(w2.w & 0xFFFFu), maxy = (int32_t)(w2.w >> 16);
            hit = !(maxx < tpx0 || minx > tpx0 + TILE - 1 || maxy < tpy0 || miny > tpy0 + TILE - 1);
            if (hit) hit = make_tile_rec(rec, box, w0, w1, w2, (int32_t)txl, (int32_t)tyl);
        }
        bool small = false, boxed = false;
        if (hit) {
            // small (and all scissor-cut) records are resolved right here, triangle-parallel; the rest is staged
            const uint32_t bw = ((box >> 8) & 0xFF) - (box & 0xFF) + 1u, bh = (box >> 24) - ((box >> 16) & 0xFF) + 1u;
            small = bw * bh <= P.tp_max_area;
            boxed = (rec[3].w & 0x80000000u) != 0u;
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
        if (lane == 0 && ball) wbase = atomicAdd(lds_count, (uint32_t)__popcll(ball));
        wbase = __builtin_amdgcn_readfirstlane(wbase);
        if (hit) {
            const uint32_t slot = wbase + (uint32_t)__popcll(ball & ((1ull << lane) - 1ull));
            lds_rec[slot * 4u + 0] = rec[0]; lds_rec[slot * 4u + 1] = rec[1];
            lds_rec[slot * 4u + 2] = rec[2]; lds_rec[slot * 4u + 3] = rec[3];
            if (!TP) lds_box[slot] = box;
        }
        __syncthreads();
        const uint32_t n = *lds_count;
        if (base == 0) { STAMP(5); STAGE_END(2u); }
        if (n) raster_chunk<KEYED, TP>(lds_rec, lds_box, n, qmask, ix0, iy0, fix0, fiy0, P, st, qbit0, lane);
    }
}

// PROGS: bit 0 = pass contains TRIANGLE-program draws, bit 1 = MODEL / MODEL_FULL draws; 4 = any mix that
// includes MODEL_PBR draws (its own variant so that the Cook-Torrance code costs the other variants no registers)
// TP: 1 = the triangle-parallel path (LDS key array) is compiled in; the host enables it for scopes with many
//     triangles per tile, sparse scopes use the leaner pixel-parallel-only variant
template <int PROGS, int KEYED, int TP>
__global__ __launch_bounds__(RASTER_THREADS, (PROGS == 1 ? (TP ? 7 : 8) : (PROGS == 2 ? 5 : 4))) void raster_kernel(const PassParams* __restrict__ params, const RasterHead H) {
    ParamsRef P = *(ParamsPtr)(uintptr_t)params;
    __shared__ uint4 lds_rec[RASTER_CHUNK * 4];
    __shared__ unsigned long long lds_key[TP ? TILE * TILE : 1];   // depth keys written by the triangle-parallel path
    __shared__ uint32_t lds_box[TP ? 1 : RASTER_CHUNK];
    __shared__ uint32_t lds_count;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t q = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave index: uniform, keep it in SGPRs
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
    const uint32_t tile = tyr * H.tiles_x + tx, ty = H.tile_row_begin + tyr;
    const int32_t ix0 = (int32_t)((q & 1u) * 16u + (lane & 7u)), iy0 = (int32_t)((q >> 1) * 16u + (lane >> 3));
    const float fix0 = (float)ix0, fiy0 = (float)iy0;
    // the four 8x8 blocks of quadrant q are bits (2*(q>>1)+by)*4 + 2*(q&1)+bx of the record's block mask
    const uint32_t qbit0 = 1u << ((q >> 1) * 8u + (q & 1u) * 2u);
    const uint32_t qmask = qbit0 * 0x33u;

    STAMP(0);
    // both counters are fetched up front so their latencies overlap
    // (the head of the parameters comes by value: the counter loads depend on the kernarg load alone, not on a second hop)
    const uint32_t count_raw = H.bin_count[tile];
    const uint32_t nbig_raw = *H.big_count;
    const uint32_t count = count_raw < H.bin_cap ? count_raw : H.bin_cap;
    const uint32_t nbig = nbig_raw < H.big_cap ? nbig_raw : H.big_cap;
    // The big-list counters are re-armed right away (no workgroup reads the other parity's counter, and the next scope
    // that uses this workspace is ordered behind this kernel), so nbig_raw need not stay live across the raster loops.
    // The tile's own bin counter is re-armed after the bin pass: every wave of this workgroup reads it above, and the
    // barriers of that pass order those reads before the store.
    if (tid == 0 && tile == 0) {
        *P.big_count_next = 0;                              // the next scope on this workspace appends to the other counter
        P.status[1] = nbig_raw;
    }

    if (TP) for (uint32_t e = tid; e < TILE * TILE; e += RASTER_THREADS) lds_key[e] = ~0ull;
    PixelState st;
#pragma unroll
    for (int b = 0; b < 4; b++) { st.zk[b] = P.init_zk; st.idk[b] = P.init_idk; }
    if (P.depth_load && P.depth) {              // second scope on a kept depth buffer: keys start from the stored depth
        const uint32_t px0 = tx * TILE + (uint32_t)ix0, py0 = ty * TILE + (uint32_t)iy0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const uint32_t px = px0 + (uint32_t)(b & 1) * BLOCK, py = py0 + (uint32_t)(b >> 1) * BLOCK;
            uint32_t zo;
            init_key(P, px, py, px < P.width && py < P.height, st.zk[b], st.idk[b], zo);
        }
    }

    STAMP(1);
    STAGE_END(1u);
    // the tile's bin, then the list every tile tests (large / clipped / spilled triangles): one copy of the code
    const uint4* list = reinterpret_cast<const uint4*>(H.bin_recs) + (size_t)tile * H.bin_cap * 3u;
    uint32_t n_list = count;
#pragma unroll 1
    for (int pass = 0; pass < 2; pass++) {
        if (n_list) raster_list<KEYED, TP>(list, n_list, lds_rec, lds_box, &lds_count, lds_key, tx, ty, qmask, ix0, iy0, fix0, fiy0, P, st,
                                           qbit0, tid, lane);
        if (pass == 0) {
            STAMP(2);
            if (count && tid == 0) H.bin_count[tile] = 0;   // ready for the next scope that uses this workspace
            if (!nbig) break;
            // parameters of this phase are (re)read here, see launder_params
            list = reinterpret_cast<const uint4*>(launder_params((ParamsPtr)(uintptr_t)params)->big_recs);
            n_list = nbig;
        }
    }

    STAMP(3);
    STAGE_END(3u);
    if (TP) __syncthreads();     // every triangle-parallel ds_min of this tile has landed
    // ---- resolve: shade the winning primitive of each pixel, store once ---------------------------
    // merge the pixel-parallel (registers) and triangle-parallel (LDS) results: smaller key wins
    if (TP) {
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const unsigned long long kreg = ((unsigned long long)st.zk[b] << 32) | st.idk[b];
            const unsigned long long klds = lds_key[(iy0 + (b >> 1) * BLOCK) * TILE + ix0 + (b & 1) * BLOCK];
            const unsigned long long kmin = klds < kreg ? klds : kreg;
            st.zk[b] = (uint32_t)(kmin >> 32); st.idk[b] = (uint32_t)kmin;
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
    if (flat_color && R->color_format != 2 && !R->prim_out && !(R->depth && R->depth_store)) {
        // a covered pixel without a flat colour has to be shaded: then the whole wave takes the general loop
        const bool need_shade = (st.idk[0] != init_idk && flat4[0] == 0u) || (st.idk[1] != init_idk && flat4[1] == 0u) ||
                                (st.idk[2] != init_idk && flat4[2] == 0u) || (st.idk[3] != init_idk && flat4[3] == 0u);
        if (__ballot(need_shade) == 0ull) {
            const uint32_t width = R->width, height = R->height, clear_packed = R->clear_packed;
            uint8_t* row0 = reinterpret_cast<uint8_t*>(R->color);
            uint8_t* row1 = row0 + (size_t)BLOCK * width * 4u;                 // the lower pair of blocks: uniform base
            const uint32_t off = (py0 * width + px0) * 4u;
            if ((tx + 1u) * TILE <= width && (ty + 1u) * TILE <= height && !R->color_load) {   // wave-uniform: interior tile
                *reinterpret_cast<uint32_t*>(row0 + off) = st.idk[0] != init_idk ? flat4[0] : clear_packed;
                *reinterpret_cast<uint32_t*>(row0 + off + 4u * BLOCK) = st.idk[1] != init_idk ? flat4[1] : clear_packed;
                *reinterpret_cast<uint32_t*>(row1 + off) = st.idk[2] != init_idk ? flat4[2] : clear_packed;
                *reinterpret_cast<uint32_t*>(row1 + off + 4u * BLOCK) = st.idk[3] != init_idk ? flat4[3] : clear_packed;
            } else {
                const uint32_t color_load = R->color_load;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const uint32_t px = px0 + (uint32_t)(b & 1) * BLOCK, py = py0 + (uint32_t)(b >> 1) * BLOCK;
                    const bool won = st.idk[b] != init_idk;
                    if (px < width && py < height && (won || !color_load))
                        *reinterpret_cast<uint32_t*>((b >> 1 ? row1 : row0) + off + 4u * BLOCK * (uint32_t)(b & 1)) = won ? flat4[b] : clear_packed;
                }
            }
            STAMP(4);
            return;
        }
    }
#pragma unroll 1
    for (int b = 0; b < 4; b++) {
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
        if (!inb) continue;
        if (!(none && P.color_load)) {
            if (P.color_format == 2) reinterpret_cast<float4*>(P.color)[pix] = make_float4(col.x, col.y, col.z, col.w);
            else reinterpret_cast<uint32_t*>(P.color)[pix] = none ? P.clear_packed : (flat ? flat : pack_bgra8_srgb(col));
        }
        if (P.prim_out && !(none && P.color_load)) P.prim_out[pix] = prim;     // LOAD keeps what an earlier scope / segment wrote
        if (P.depth && P.depth_store) {
            const uint32_t zb = (none || !P.zmask) ? zorig : (zkb ^ P.zflip);
            P.depth[pix] = __uint_as_float(zb);
        }
    }
    STAMP(4);
}

// ------------------------------------------------------------------------------------------------
// launch wrappers (host side of this translation unit)
// ------------------------------------------------------------------------------------------------
hipError_t launch_vertex(const PassParams& P, const PassParams* dev_params, hipStream_t stream) {
    if (P.vs_total_slots == 0) return hipSuccess;
    hipLaunchKernelGGL(vertex_kernel, dim3(P.vs_total_slots / GEOM_THREADS), dim3(GEOM_THREADS), 0, stream, dev_params);
    return hipGetLastError();
}

hipError_t launch_geometry(const PassParams& P, const PassParams* dev_params, hipStream_t stream) {
    if (P.total_slots == 0) return hipSuccess;
    const uint32_t blocks = P.total_slots / GEOM_THREADS;
    const GeometryHead H = {P.draws, P.num_draws};
    hipLaunchKernelGGL(geometry_kernel, dim3(blocks), dim3(GEOM_THREADS), 0, stream, dev_params, H);
    return hipGetLastError();
}

template <int KEYED, int TP>
static void launch_raster_k(const PassParams* P, const RasterHead& H, uint32_t programs, dim3 grid, hipStream_t stream) {
    const dim3 block(RASTER_THREADS);
    if (programs == 2) hipLaunchKernelGGL((raster_kernel<2, KEYED, TP>), grid, block, 0, stream, P, H);
    else if (programs == 3) hipLaunchKernelGGL((raster_kernel<3, KEYED, TP>), grid, block, 0, stream, P, H);
    else if (programs >= 4) hipLaunchKernelGGL((raster_kernel<4, KEYED, TP>), grid, block, 0, stream, P, H);
    else hipLaunchKernelGGL((raster_kernel<1, KEYED, TP>), grid, block, 0, stream, P, H);
}

hipError_t launch_raster(const PassParams& P, const PassParams* dev_params, uint32_t* big_count, uint32_t programs, hipStream_t stream) {
    const uint32_t rows = P.tile_row_end - P.tile_row_begin;
    if (rows == 0 || P.tiles_x == 0) return hipSuccess;
    const dim3 grid = P.xcd_swizzle > 1u ? dim3(P.tiles_x * rows) : dim3(P.tiles_x, rows);
    // the plain key (raw float bits) serves LESS / LESS_OR_EQUAL; everything else takes the generic key
    const bool plain = P.zflip == 0u && P.zmask == 0xFFFFFFFFu;
    const RasterHead H = {P.bin_count, P.bin_recs, big_count, P.tiles_x, P.tile_row_begin, P.bin_cap, P.big_cap};
    if (P.pred) launch_raster_k<2, 0>(dev_params, H, programs, grid, stream);          // (the host keeps tp_max_area = 0 for predicate scopes)
    else if (P.tp_max_area) { if (plain) launch_raster_k<0, 1>(dev_params, H, programs, grid, stream); else launch_raster_k<1, 1>(dev_params, H, programs, grid, stream); }
    else { if (plain) launch_raster_k<0, 0>(dev_params, H, programs, grid, stream); else launch_raster_k<1, 0>(dev_params, H, programs, grid, stream); }
    return hipGetLastError();
}

#ifdef MIRHI_STAMPS
extern "C" int mirhi_debug_read_stamps(uint64_t* dst, uint32_t count) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps), (size_t)count * 8, 0, hipMemcpyDeviceToHost);
}
extern "C" int mirhi_debug_read_geo_stamps(uint64_t* dst, uint32_t count) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps_geo), (size_t)count * 8, 0, hipMemcpyDeviceToHost);
}
extern "C" int mirhi_debug_set_stage_limit(uint32_t v) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stage_limit), &v, sizeof v, 0, hipMemcpyHostToDevice);
}
extern "C" int mirhi_debug_clear_stamps() {
    void* p = nullptr;
    hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_stamps));
    if (e != hipSuccess) return (int)e;
    return (int)hipMemset(p, 0, sizeof(uint64_t) * 16384 * 8);
}
#endif

}  // namespace mirhi
