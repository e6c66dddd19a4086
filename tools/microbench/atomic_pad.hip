// Returning agent-scope atomics shaped like C2's bin reservation: W one-wave workgroups, each lane R rounds of one atomicAdd on a random one of T counters
// (every wave spreads over the whole frame).  Does it matter how the counters are laid out?  (a) contiguous u32 (32 tiles' counters share a 128-byte line),
// (b) one counter per 128-byte line, (c) one per 64 bytes, (d) contiguous, one copy per XCD.  Reports the kernel's duration (event pair) and a wave's own
// time from its first atomic issued to its last result (s_memtime).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ __launch_bounds__(64) void k(uint32_t* ctr, uint32_t T, uint32_t stride, uint32_t xcd_stride, uint32_t rounds, uint32_t* sink, uint64_t* span) {
    const uint32_t xcd = xcd_stride ? (__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u) : 0u;
    uint32_t h = (blockIdx.x * 64u + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    uint32_t r[8];
#pragma unroll
    for (uint32_t i = 0; i < 8; i++) {          // all rounds' atomics in flight before the first result is used (bin_triangle_pairs, BATCH = 8)
        h = h * 1664525u + 1013904223u;
        r[i] = i < rounds ? atomicAdd(&ctr[xcd * xcd_stride + ((h >> 8) % T) * stride], 1u) : 0u;
    }
#pragma unroll
    for (uint32_t i = 0; i < 8; i++) acc += r[i];
    __builtin_amdgcn_s_waitcnt(0);
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (acc == 0xFFFFFFFFu) sink[0] = acc;
    if (threadIdx.x == 0) span[blockIdx.x] = t1 - t0;
}
int main() {
    const uint32_t T = 2040;
    uint32_t *ctr, *sink; uint64_t* span;
    CK(hipMalloc(&ctr, 8u * T * 128)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&span, 8192 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct V { const char* name; uint32_t stride, xcd_stride; } vs[] = {{"contiguous u32 counters", 1, 0}, {"one counter per 64 bytes", 16, 0}, {"one counter per 128 bytes", 32, 0}, {"contiguous, one copy per XCD", 1, T}, {"128 bytes apart, one copy per XCD", 32, T * 32}};
    for (uint32_t W : {157u, 625u}) for (uint32_t rounds : {2u, 5u}) {
        if ((W == 157 && rounds == 2) || (W == 625 && rounds == 5)) continue;          // 157 waves x 5 rounds = 625 waves x ~1.3: the two shapes of C2 (64 / 16 triangles per wave)
        for (auto& v : vs) {
            double best = 1e9, wave = 0;
            for (int it = 0; it < 20; it++) {
                CK(hipMemset(ctr, 0, 8u * T * 128));
                CK(hipDeviceSynchronize());
                hipExtLaunchKernelGGL(k, dim3(W), dim3(64), 0, 0, e0, e1, 0, ctr, T, v.stride, v.xcd_stride, rounds, sink, span);
                CK(hipDeviceSynchronize());
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                std::vector<uint64_t> h(W); CK(hipMemcpy(h.data(), span, W * 8, hipMemcpyDeviceToHost));
                std::sort(h.begin(), h.end());
                if (ms * 1e3 < best) { best = ms * 1e3; wave = (double)h[W / 2] / 100.0; }      // s_memtime: 100 MHz
            }
            printf("%u waves x %u rounds, %-36s: kernel %.2f us, median wave (first atomic -> last result) %.2f us\n", W, rounds, v.name, best, wave);
        }
    }
    return 0;
}
