// Returning global atomics under contention, shaped like the geometry kernel's bin reservation:
//   W one-wave workgroups, each lane adds to one of HOT counters.
//   mode 0  every wave's 64 lanes spread over ~30 of the same 232 counters (an index order that is spatially incoherent:
//           every wave hits every hot tile) -- the dancer asset
//   mode 1  same, but each XCD has its own copy of the counters (index + xcc_id * HOT): does keeping a counter's cache
//           line inside one XCD's L2 pay?
//   mode 2  consecutive waves hit consecutive counters (coherent mesh order: ~4 counters per wave)
//   mode 3  no sharing at all (every lane its own counter)
// usage: atomic_contention [waves=269]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int HOT = 232;
__global__ void k(uint32_t* ctr, uint32_t* sink, int mode) {
    const uint32_t lane = threadIdx.x, w = blockIdx.x;
    uint32_t h = (w * 2654435761u) ^ (lane * 40503u); h ^= h >> 13;
    uint32_t idx;
    if (mode == 0 || mode == 1) idx = (h % 30u) * 7u % HOT + (h >> 20) % 8u;        // ~30 distinct of the hot set per wave
    else if (mode == 2) idx = (w * 4u + (lane >> 4)) % HOT;
    else idx = w * 64u + lane;
    if (mode == 1) idx += HOT * 2u * (__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u);
    const uint32_t old = atomicAdd(&ctr[idx], 1u);
    // a dependent second round, as the record store depends on the slot
    sink[(w * 64u + lane)] = old;
}
int main(int argc, char** argv) {
    const int waves = argc > 1 ? atoi(argv[1]) : 269;
    uint32_t *ctr, *sink;
    CK(hipMalloc(&ctr, 4u << 20)); CK(hipMalloc(&sink, (size_t)waves * 64 * 4 + 4096));
    CK(hipMemset(ctr, 0, 4u << 20));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int mode = 0; mode < 4; mode++) {
        for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k, dim3(waves), dim3(64), 0, 0, ctr, sink, mode);
        CK(hipDeviceSynchronize());
        const int n = 2000;
        CK(hipEventRecord(a, 0));
        for (int i = 0; i < n; i++) hipLaunchKernelGGL(k, dim3(waves), dim3(64), 0, 0, ctr, sink, mode);
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("mode %d: %.2f us per launch (%d waves, back to back)\n", mode, 1e3 * ms / n, waves);
    }
    return 0;
}
