// device code of the overlap experiment: a latency-bound "geometry" kernel and a "raster" kernel that waits for it inside the launch
#include <hip/hip_runtime.h>
#include <stdint.h>
struct GArgs { uint32_t* done; uint32_t* scratch; uint32_t spin; uint32_t pad; };
extern "C" __global__ __launch_bounds__(64) void geo_kernel(GArgs a) {
    // ~spin x 100 ns of dependent work per wave, then one store and the wave's "done" count
    uint32_t v = threadIdx.x;
    for (uint32_t i = 0; i < a.spin; i++) { __builtin_amdgcn_s_sleep(4); v = v * 1664525u + 1013904223u; }
    a.scratch[blockIdx.x * 64 + threadIdx.x] = v;
    __builtin_amdgcn_s_waitcnt(0);
    if (threadIdx.x == 0) __hip_atomic_fetch_add(&a.done[(blockIdx.x & 7u) * 32u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
struct RArgs { uint32_t* done; uint32_t* frame; uint32_t* stats; uint32_t target; uint32_t wait; };
extern "C" __global__ __launch_bounds__(256) void ras_kernel(RArgs a) {
    __shared__ uint32_t ok;
    if (a.wait) {
        if (threadIdx.x == 0) {
            uint32_t spins = 0, sum = 0;
            for (; spins < (1u << 11); spins++) {
                sum = 0;
                for (uint32_t k = 0; k < 8u; k++) sum += __hip_atomic_load(&a.done[k * 32u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((int32_t)(sum - a.target) >= 0) break;
                __builtin_amdgcn_s_sleep(32);
            }
            ok = (int32_t)(sum - a.target) >= 0;
            if (!ok) atomicAdd(&a.stats[0], 1u);            // timeouts
            if (spins) atomicAdd(&a.stats[1], 1u);          // workgroups that had to wait at all
        }
        __syncthreads();
        if (!ok) return;
    }
    const uint32_t id = blockIdx.y * gridDim.x + blockIdx.x;
    uint32_t v = id;
    for (int i = 0; i < 40; i++) { __builtin_amdgcn_s_sleep(4); v = v * 1664525u + 1013904223u; }     // ~4 us of "raster" per workgroup
    a.frame[(size_t)id * 256 + threadIdx.x] = v;
}
