// wg_dispatch.hip -- what does it cost to get a grid of short workgroups through the chip?  The raster kernel launches one 256-lane
// workgroup per 32x32 tile (2040 at 1080p) whose average wave lives ~1.6 us; this measures kernels of the same SHAPE (threads per
// workgroup, LDS per workgroup, registers per lane) whose workgroups do almost nothing (one dependent global load, one store), back
// to back on one stream, for several ways of cutting the same 522,240 lanes into workgroups.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>

template <int THREADS, int LDS_WORDS, int WAVES_PER_EU>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(WAVES_PER_EU, WAVES_PER_EU))) void shaped(const uint32_t* __restrict__ counts, uint32_t* __restrict__ out) {
    __shared__ uint32_t lds[LDS_WORDS];
    const uint32_t c = counts[blockIdx.x];                       // one dependent load, like the tile's bin counter
    if (threadIdx.x < 64) lds[threadIdx.x] = c;
    __syncthreads();
    if (c != 0xFFFFFFFFu) out[blockIdx.x * THREADS + threadIdx.x] = lds[threadIdx.x & 63] + threadIdx.x;   // 4 bytes per lane
}

template <class F> double back_to_back(F f, hipStream_t s, int n = 2000) {
    for (int i = 0; i < 50; i++) f();
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; i++) f();
    hipStreamSynchronize(s);
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
}

int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    uint32_t *counts, *out;
    hipMalloc(&counts, 1 << 20); hipMemset(counts, 0, 1 << 20);
    hipMalloc(&out, 64 << 20);
    const uint32_t lanes = 2040u * 256u;
    printf("%-44s %8s\n", "shape (same 522,240 lanes, 4 B stored per lane)", "us");
#define RUN(T, L, W, label) printf("%-44s %8.2f\n", label, back_to_back([&] { hipLaunchKernelGGL((shaped<T, L, W>), dim3(lanes / T), dim3(T), 0, s, counts, out); }, s));
    RUN(64, 64, 8, "8160 workgroups x 64 lanes, 0.25 KB LDS");
    RUN(128, 64, 8, "4080 workgroups x 128 lanes, 0.25 KB LDS");
    RUN(256, 64, 8, "2040 workgroups x 256 lanes, 0.25 KB LDS");
    RUN(256, 3328, 8, "2040 workgroups x 256 lanes, 13 KB LDS");
    RUN(512, 64, 8, "1020 workgroups x 512 lanes, 0.25 KB LDS");
    RUN(512, 6656, 8, "1020 workgroups x 512 lanes, 26 KB LDS");
    RUN(1024, 64, 8, "510 workgroups x 1024 lanes, 0.25 KB LDS");
    RUN(1024, 13312, 8, "510 workgroups x 1024 lanes, 52 KB LDS");
    printf("%-44s %8.2f\n", "empty 1 x 64", back_to_back([&] { hipLaunchKernelGGL((shaped<64, 64, 8>), dim3(1), dim3(64), 0, s, counts, out); }, s));
    return 0;
}
