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
#include <hip/hip_ext.h>
#include <cstdlib>
#include <stdint.h>

#include "mirhi_device.h"
#include "mirhi_launch.h"

namespace mirhi {

#include "mirhi_exact.hip.h"
#include "mirhi_common.hip.h"
#include "mirhi_geometry.hip.h"
#include "mirhi_shading.hip.h"
#include "mirhi_raster.hip.h"
#include "mirhi_stats.hip.h"
#include "mirhi_ordered.hip.h"

// the build this code object belongs to (build.py passes the source hash to both translation units; native_device_open compares)
#ifndef MIRHI_SOURCE_HASH
#define MIRHI_SOURCE_HASH "unknown"
#endif
__device__ __attribute__((used)) char g_build_id[17] = MIRHI_SOURCE_HASH;       // (not const: a const at namespace scope is a local symbol, the loader would not find it)
// measurement only (mirhi_device_measure_roundtrip): one wave that does nothing
__global__ __launch_bounds__(64) void noop_kernel(uint32_t) {}
// behind a natively dispatched frame of a tile split: tells the band-exchange stream (hipStreamWaitValue64 on signal memory) that the frame is rendered
__global__ __launch_bounds__(64) void seq_store_kernel(uint64_t* word, uint64_t value) {
    if (threadIdx.x == 0) __hip_atomic_store(word, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------------------------------------------------
// launch wrappers (host side of this translation unit)
// ------------------------------------------------------------------------------------------------
hipError_t upload_srgb_lut(const float* lut, hipStream_t stream) {
    return hipMemcpyToSymbolAsync(HIP_SYMBOL(g_srgb_lut), lut, 256 * sizeof(float), 0, hipMemcpyHostToDevice, stream);
}

// failure of a native dispatch inside the launch wrappers below (they report it with their return value)
static thread_local hipError_t t_native_err = hipSuccess;
static inline hipError_t launch_result() { const hipError_t e = t_native_err; t_native_err = hipSuccess; return e != hipSuccess ? e : hipGetLastError(); }
// A plain launch, or -- when the caller wants the dispatch timed -- one with an event pair attached to the dispatch itself, or a native dispatch.
#define MIRHI_LAUNCH(kernel, grid, block, stream, t, ...)                                                               \
    do {                                                                                                                \
        if ((t).native) {                                                                                               \
            const hipError_t ne__ = native_launch((t).native, reinterpret_cast<const void*>(+kernel), grid, block, (t).native_signal, (t).native_flags, __VA_ARGS__); \
            if (ne__ != hipSuccess) t_native_err = ne__;                                                                \
        }                                                                                                               \
        else if ((t).start || (t).stop) hipExtLaunchKernelGGL(kernel, grid, block, 0, stream, (t).start, (t).stop, 0, __VA_ARGS__);     \
        else hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__);                                           \
    } while (0)

hipError_t launch_noop(NativeQueue* q, uint64_t signal) {
    return native_launch(q, reinterpret_cast<const void*>(+noop_kernel), dim3(1), dim3(64), signal, NATIVE_RELEASE_SYSTEM, (uint32_t)0);
}

hipError_t launch_seq_store(NativeQueue* q, uint64_t* word, uint64_t value) {
    return native_launch(q, reinterpret_cast<const void*>(+seq_store_kernel), dim3(1), dim3(64), 0, NATIVE_RELEASE_SYSTEM, word, value);
}

hipError_t launch_vertex(const PassParams& P, const PassParams* dev_params, hipStream_t stream, LaunchTiming t) {
    if (P.vs_total_slots == 0) return hipSuccess;
    MIRHI_LAUNCH(vertex_kernel, dim3(P.vs_total_slots / GEOM_THREADS), dim3(GEOM_THREADS), stream, t, dev_params);
    return launch_result();
}

hipError_t launch_geometry(const PassParams& P, const PassParams* dev_params, hipStream_t stream, LaunchTiming t) {
    if (P.total_slots == 0) return hipSuccess;
    const uint32_t tpw = (t.tris_per_wave == 16u || t.tris_per_wave == 32u) ? t.tris_per_wave : (uint32_t)GEOM_THREADS;
    const uint32_t blocks = P.total_slots / tpw;
    GeometryHead H = {P.draws, P.num_draws, tpw, nullptr, 0u, 0u, 0u};
    if (t.head_draw && P.num_draws == 1u && t.head_draw->index_type == 0u && t.head_draw->program == 0u && t.head_draw->vs_words == 0u && t.head_draw->stride >= 24u) {
        H.vb0 = t.head_draw->vb; H.stride0 = t.head_draw->stride; H.first0 = t.head_draw->first; H.tris0 = t.head_draw->tri_count;
    }
    // (more waves than the chip holds at five per SIMD: the occupancy-oriented variant)
    if (blocks > 5u * 1024u) MIRHI_LAUNCH(geometry_kernel<7>, dim3(blocks), dim3(GEOM_THREADS), stream, t, dev_params, H);
    else MIRHI_LAUNCH(geometry_kernel<5>, dim3(blocks), dim3(GEOM_THREADS), stream, t, dev_params, H);
    return launch_result();
}

template <int KEYED, int TP, int TEAMS = 1>
static void launch_raster_k(const PassParams* P, const RasterHead& H, uint32_t programs, dim3 grid, hipStream_t stream, LaunchTiming t, bool masked = false) {
    const dim3 block(RASTER_THREADS * TEAMS);
    if constexpr (TP != 0) {    // alpha-masked scope (host: PassParams::alpha_scope, always with the triangle-parallel path and a PBR draw)
        if (masked) {
            MIRHI_LAUNCH((raster_kernel<4, KEYED, TP, TEAMS, true>), grid, block, stream, t, P, H);
            return;
        }
    }
    if (TEAMS > 1) {            // only the pure mesh variants exist with two teams (launch_raster checks)
        if (programs == 2) MIRHI_LAUNCH((raster_kernel<2, KEYED, TP, TEAMS>), grid, block, stream, t, P, H);
        else MIRHI_LAUNCH((raster_kernel<4, KEYED, TP, TEAMS>), grid, block, stream, t, P, H);
        return;
    }
    if (programs == 2) MIRHI_LAUNCH((raster_kernel<2, KEYED, TP>), grid, block, stream, t, P, H);
    else if (programs == 3) MIRHI_LAUNCH((raster_kernel<3, KEYED, TP>), grid, block, stream, t, P, H);
    else if (programs >= 4) MIRHI_LAUNCH((raster_kernel<4, KEYED, TP>), grid, block, stream, t, P, H);
    else MIRHI_LAUNCH((raster_kernel<1, KEYED, TP>), grid, block, stream, t, P, H);
}

hipError_t launch_raster(const PassParams& P, const PassParams* dev_params, uint32_t* big_count, uint32_t programs, hipStream_t stream, LaunchTiming t, bool allow_wide) {
    const uint32_t rows = P.tile_row_end - P.tile_row_begin;
    if (rows == 0 || P.tiles_x == 0) return hipSuccess;
    if (P.ordered_recs) {           // ordered segment: fragments in primitive order (blending)
        const RasterHead HO = {P.bin_count, P.bin_pool, big_count, P.tiles_x, P.tile_row_begin, P.bin_cap, P.big_cap, P.bin_cap, 0u, 0u, P.tile_row_step};
        const dim3 og(P.tiles_x, rows), ob(ORDERED_THREADS);
        if (programs == 2) MIRHI_LAUNCH(ordered_kernel<2>, og, ob, stream, t, dev_params, HO);
        else if (programs == 3) MIRHI_LAUNCH(ordered_kernel<3>, og, ob, stream, t, dev_params, HO);
        else if (programs >= 4) MIRHI_LAUNCH(ordered_kernel<4>, og, ob, stream, t, dev_params, HO);
        else MIRHI_LAUNCH(ordered_kernel<1>, og, ob, stream, t, dev_params, HO);
        return launch_result();
    }
    const dim3 grid = P.xcd_swizzle > 1u ? dim3(P.tiles_x * rows) : dim3(P.tiles_x, rows);
    // the plain key (raw float bits) serves LESS / LESS_OR_EQUAL; everything else takes the generic key
    const bool plain = P.zflip == 0u && P.zmask == 0xFFFFFFFFu;
    const RasterHead H = {P.bin_count, P.bin_pool, big_count, P.tiles_x, P.tile_row_begin, P.bin_cap, P.big_cap, P.sub_cap, P.count_stride, P.fixed_recs, P.tile_row_step};
    if (P.raster_wide && allow_wide && !P.pred && P.tp_max_area && !P.alpha_scope && (programs == 2 || programs >= 4) && P.xcd_swizzle <= 1u) {
        // the wide mesh variants: eight or sixteen waves per tile (host-side choice, PassParams::raster_wide = waves per tile)
        const bool w16 = P.raster_wide >= 16u;
        const dim3 wb(w16 ? 1024 : 512);
#define MIRHI_WIDE(PR, KE) do { if (w16) MIRHI_LAUNCH((raster_kernel_wide<PR, KE, 16>), grid, wb, stream, t, dev_params, H); else MIRHI_LAUNCH((raster_kernel_wide<PR, KE, 8>), grid, wb, stream, t, dev_params, H); } while (0)
        if (programs == 2) { if (plain) MIRHI_WIDE(2, 0); else MIRHI_WIDE(2, 1); }
        else { if (plain) MIRHI_WIDE(4, 0); else MIRHI_WIDE(4, 1); }
#undef MIRHI_WIDE
    }
    else if (P.pred) launch_raster_k<2, 0>(dev_params, H, programs, grid, stream, t);          // (the host keeps tp_max_area = 0 for predicate scopes)
    else if (P.tp_max_area && P.raster_teams == 2u && (programs == 2 || programs >= 4)) {
        if (plain) launch_raster_k<0, 1, 2>(dev_params, H, programs, grid, stream, t, P.alpha_scope != 0u); else launch_raster_k<1, 1, 2>(dev_params, H, programs, grid, stream, t, P.alpha_scope != 0u);
    }
    else if (P.tp_max_area) { if (plain) launch_raster_k<0, 1>(dev_params, H, programs, grid, stream, t, P.alpha_scope != 0u); else launch_raster_k<1, 1>(dev_params, H, programs, grid, stream, t, P.alpha_scope != 0u); }
    else { if (plain) launch_raster_k<0, 0>(dev_params, H, programs, grid, stream, t); else launch_raster_k<1, 0>(dev_params, H, programs, grid, stream, t); }
    return launch_result();
}

// ---- batched launches --------------------------------------------------------------------------------------------
uint64_t raster_variant_key(const PassParams& P, uint32_t programs) {
    const bool plain = P.zflip == 0u && P.zmask == 0xFFFFFFFFu;
    const uint32_t keyed = P.pred ? 2u : (plain ? 0u : 1u);
    const uint32_t tp = (!P.pred && P.tp_max_area) ? 1u : 0u;
    const uint32_t teams = (tp && P.raster_teams == 2u && (programs == 2 || programs >= 4)) ? 2u : 1u;
    const uint32_t prog = programs >= 4 ? 4u : programs;
    return (uint64_t)prog | ((uint64_t)keyed << 4) | ((uint64_t)tp << 8) | ((uint64_t)teams << 12) | ((uint64_t)(P.xcd_swizzle > 1u ? 1u : 0u) << 16) |
           ((uint64_t)P.tiles_x << 20) | ((uint64_t)(P.tile_row_end - P.tile_row_begin) << 36) | ((uint64_t)(P.ordered_recs ? 1u : 0u) << 52) |
           ((uint64_t)(P.raster_wide >> 3) << 53);
}

hipError_t launch_vertex_batch(const PassParams* const* P, const PassParams* const* dev_params, uint32_t n, hipStream_t stream) {
    GeometryBatch B{};
    uint32_t most = 0;
    for (uint32_t i = 0; i < n; i++) { B.params[i] = dev_params[i]; most = P[i]->vs_total_slots > most ? P[i]->vs_total_slots : most; }
    if (most == 0) return hipSuccess;
    hipLaunchKernelGGL(vertex_kernel_batch, dim3(most / GEOM_THREADS, n), dim3(GEOM_THREADS), 0, stream, B);
    return launch_result();
}

hipError_t launch_geometry_batch(const PassParams* const* P, const PassParams* const* dev_params, uint32_t n, hipStream_t stream) {
    GeometryBatch B{};
    uint32_t most = 0, total = 0;
    for (uint32_t i = 0; i < n; i++) {
        B.params[i] = dev_params[i];
        B.head[i] = GeometryHead{P[i]->draws, P[i]->num_draws, (uint32_t)GEOM_THREADS, nullptr, 0u, 0u, 0u};
        B.blocks[i] = P[i]->total_slots / GEOM_THREADS;
        most = B.blocks[i] > most ? B.blocks[i] : most; total += B.blocks[i];
    }
    if (most == 0) return hipSuccess;
    if (total > 5u * 1024u) hipLaunchKernelGGL(geometry_kernel_batch<7>, dim3(most, n), dim3(GEOM_THREADS), 0, stream, B);
    else hipLaunchKernelGGL(geometry_kernel_batch<5>, dim3(most, n), dim3(GEOM_THREADS), 0, stream, B);
    return launch_result();
}

template <int KEYED, int TP, int TEAMS = 1>
static void launch_raster_batch_k(const RasterBatch& B, uint32_t programs, dim3 grid, hipStream_t stream, LaunchTiming t) {
    const dim3 block(RASTER_THREADS * TEAMS);
    if (TEAMS > 1) {
        if (programs == 2) MIRHI_LAUNCH((raster_kernel_batch<2, KEYED, TP, TEAMS>), grid, block, stream, t, B);
        else MIRHI_LAUNCH((raster_kernel_batch<4, KEYED, TP, TEAMS>), grid, block, stream, t, B);
        return;
    }
    if (programs == 2) MIRHI_LAUNCH((raster_kernel_batch<2, KEYED, TP>), grid, block, stream, t, B);
    else if (programs == 3) MIRHI_LAUNCH((raster_kernel_batch<3, KEYED, TP>), grid, block, stream, t, B);
    else if (programs >= 4) MIRHI_LAUNCH((raster_kernel_batch<4, KEYED, TP>), grid, block, stream, t, B);
    else MIRHI_LAUNCH((raster_kernel_batch<1, KEYED, TP>), grid, block, stream, t, B);
}

hipError_t launch_raster_batch(const PassParams* const* Ps, const PassParams* const* dev_params, uint32_t* const* big_count, uint32_t n, uint32_t programs, hipStream_t stream, hipEvent_t stop) {
    LaunchTiming t{}; t.stop = stop;
    const PassParams& P = *Ps[0];
    const uint32_t rows = P.tile_row_end - P.tile_row_begin;
    if (rows == 0 || P.tiles_x == 0) return hipSuccess;
    RasterBatch B{};
    for (uint32_t i = 0; i < n; i++) {
        const PassParams& Q = *Ps[i];
        B.params[i] = dev_params[i];
        B.head[i] = RasterHead{Q.bin_count, Q.bin_pool, big_count[i], Q.tiles_x, Q.tile_row_begin, Q.bin_cap, Q.big_cap, Q.sub_cap, Q.count_stride, Q.fixed_recs, Q.tile_row_step};
    }
    const dim3 grid(P.tiles_x, rows, n);           // (the XCD run-length order of a 1-D grid is a measurement knob: such scopes are not batched)
    const bool plain = P.zflip == 0u && P.zmask == 0xFFFFFFFFu;
    // only the variants a frame loop meets are instantiated in batched form: LESS / LESS_OR_EQUAL keys, with and without the
    // triangle-parallel path and the two-team mesh mode; everything else goes scope by scope (mirhi_queue_submit checks)
    if (P.tp_max_area && P.raster_teams == 2u && (programs == 2 || programs >= 4)) launch_raster_batch_k<0, 1, 2>(B, programs, grid, stream, t);
    else if (P.tp_max_area) launch_raster_batch_k<0, 1>(B, programs, grid, stream, t);
    else launch_raster_batch_k<0, 0>(B, programs, grid, stream, t);
    (void)plain;
    return launch_result();
}
bool raster_batchable(const PassParams& P) { return !P.pred && P.zflip == 0u && P.zmask == 0xFFFFFFFFu && P.xcd_swizzle <= 1u && !P.ordered_recs && !P.alpha_scope && !P.raster_wide; }

hipError_t launch_fragment_count(const PassParams& P, const PassParams* dev_params, uint32_t* big_count, hipStream_t stream, LaunchTiming t) {
    const uint32_t rows = P.tile_row_end - P.tile_row_begin;
    if (rows == 0 || P.tiles_x == 0 || P.ordered_recs) return hipSuccess;
    const RasterHead H = {P.bin_count, P.bin_pool, big_count, P.tiles_x, P.tile_row_begin, P.bin_cap, P.big_cap, P.sub_cap, P.count_stride, P.fixed_recs, P.tile_row_step};
    MIRHI_LAUNCH(fragment_count_kernel, dim3(P.tiles_x, rows), dim3(RASTER_THREADS), stream, t, dev_params, H);
    return launch_result();
}

hipError_t launch_winner_count(const uint32_t* prim, uint32_t pixels, unsigned long long* stats, hipStream_t stream) {
    if (!pixels) return hipSuccess;
    const uint32_t blocks = (pixels + RASTER_THREADS * 16u - 1u) / (RASTER_THREADS * 16u);
    hipLaunchKernelGGL(winner_count_kernel, dim3(blocks), dim3(RASTER_THREADS), 0, stream, prim, pixels, stats);
    return launch_result();
}

#ifdef MIRHI_STAMPS
extern "C" int mirhi_debug_read_stamps(uint64_t* dst, uint32_t count) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps), (size_t)count * 8, 0, hipMemcpyDeviceToHost);
}
extern "C" int mirhi_debug_read_geo_stamps(uint64_t* dst, uint32_t count) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps_geo), (size_t)count * 8, 0, hipMemcpyDeviceToHost);
}
extern "C" int mirhi_debug_read_geo_clock(uint64_t* dst, uint32_t count) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps_geo_rt), (size_t)count * 8, 0, hipMemcpyDeviceToHost);
}
extern "C" int mirhi_debug_set_stage_limit(uint32_t v) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stage_limit), &v, sizeof v, 0, hipMemcpyHostToDevice);
}
extern "C" int mirhi_debug_clear_stamps() {
    void* p = nullptr;
    hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_stamps));
    if (e != hipSuccess) return (int)e;
    return (int)hipMemset(p, 0, sizeof(uint64_t) * 32768 * 8);
}
#endif

}  // namespace mirhi
