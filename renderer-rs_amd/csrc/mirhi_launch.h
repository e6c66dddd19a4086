// mirhi_launch.h -- host-callable launchers of the kernels in mirhi_kernels.hip
#pragma once
#include <hip/hip_runtime.h>

#include "mirhi_device.h"
#include "mirhi_native.h"

namespace mirhi {
// P: host copy (launch geometry); dev_params: the same parameters in device memory, read by the kernels.
// Timing: a non-null event pair is attached to the dispatch itself (hipExtLaunchKernelGGL): hipEventElapsedTime(start, stop)
// is then the kernel's own begin -> end on the GPU clock, what rocprofv3 --kernel-trace reports, with no event-record
// commands of its own in the stream.  Null events: a plain launch.  A stop event alone: the dispatch's completion signal, which is
// how a submit's fence is signalled (mirhi_queue_submit).
// native: dispatch on this AQL queue instead of a HIP stream (mirhi_native.h); native_signal: an hsa_signal_t handle decremented at the
// kernel's end (a submit's fence), 0 = none.  Never together with start / stop.
// tris_per_wave: GeometryHead::tris_per_wave (0 = 64); head_draw: the HOST copy of the scope's only draw descriptor (GeometryHead::vb0 is filled from it), or nullptr
struct LaunchTiming { hipEvent_t start = nullptr, stop = nullptr; NativeQueue* native = nullptr; uint64_t native_signal = 0; uint32_t native_flags = 0; uint32_t tris_per_wave = 0;
                      const DrawDesc* head_draw = nullptr; };
hipError_t launch_vertex(const PassParams& P, const PassParams* dev_params, hipStream_t stream, LaunchTiming t = {});      // no-op unless the scope uses MODEL programs
hipError_t launch_geometry(const PassParams& P, const PassParams* dev_params, hipStream_t stream, LaunchTiming t = {});
// big_count: the large-triangle counter of this submit's parity (dev_params carries the same pointer)
// programs: bit0 TRIANGLE, bit1 MODEL / MODEL_FULL, bit2 MODEL_PBR
// allow_wide: false = the scope's plain / two-team variant even if its plan names a wide one (PassParams::raster_wide): the submit's choice
hipError_t launch_raster(const PassParams& P, const PassParams* dev_params, uint32_t* big_count, uint32_t programs, hipStream_t stream, LaunchTiming t = {}, bool allow_wide = true);
// Batched forms: n (2 .. MAX_BATCH) independent scopes of equal target shape and equal kernel variants (raster_variant_key) in one
// launch each; P[i] / dev_params[i] / big_count[i] as above, per scope.  Ordered (blended) scopes are never batched.
bool raster_batchable(const PassParams& P);                                 // a variant that exists in batched form
uint64_t raster_variant_key(const PassParams& P, uint32_t programs);        // equal keys <=> the same raster_kernel instantiation and grid
hipError_t launch_vertex_batch(const PassParams* const* P, const PassParams* const* dev_params, uint32_t n, hipStream_t stream);
hipError_t launch_geometry_batch(const PassParams* const* P, const PassParams* const* dev_params, uint32_t n, hipStream_t stream);
hipError_t launch_raster_batch(const PassParams* const* P, const PassParams* const* dev_params, uint32_t* const* big_count, uint32_t n, uint32_t programs, hipStream_t stream,
                               hipEvent_t stop = nullptr);      // stop: signalled by the dispatch's completion (a submit's fence)
// profiling only: counts the fragments the scope's binned triangles cover (before any depth test) into P.frag_stats[1];
// runs between the geometry and the raster kernel (the raster kernel re-arms the bin counters)
hipError_t launch_fragment_count(const PassParams& P, const PassParams* dev_params, uint32_t* big_count, hipStream_t stream, LaunchTiming t = {});
// statistics pass only: adds the number of pixels of `prim` (the scope's primitive-id image, NO_PRIM where nothing won) that hold
// a primitive to stats[0]
hipError_t launch_winner_count(const uint32_t* prim, uint32_t pixels, unsigned long long* stats, hipStream_t stream);
// measurement: one empty one-wave kernel on an AQL queue, `signal` decremented at its end (mirhi_device_measure_roundtrip)
hipError_t launch_noop(NativeQueue* q, uint64_t signal);
// one single-lane kernel on an AQL queue that stores `value` into `word` (signal memory, system scope): a HIP stream waits for it (hipStreamWaitValue64)
hipError_t launch_seq_store(NativeQueue* q, uint64_t* word, uint64_t value);
// sRGB byte -> linear table of the current device (R8G8B8A8_SRGB textures); 256 floats
hipError_t upload_srgb_lut(const float* lut, hipStream_t stream);   // asynchronous: the caller synchronises `stream`
}  // namespace mirhi
