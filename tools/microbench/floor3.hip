#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
struct PP { uint32_t w[40]; uint32_t* cnt; uint32_t* big; uint32_t* out; float* depth; };
// NB: big dead-code region between the counter loads and the stores (executed only when counters != 0)
template <int N, bool EXEC>
__global__ __launch_bounds__(256) void kF(PP P) {
    __shared__ uint4 lds[1024 + 70];
    uint32_t tile = blockIdx.x, tx = tile % P.w[2], ty = P.w[4] + tile / P.w[2], l = threadIdx.x & 63, q = threadIdx.x >> 6;
    uint32_t c = P.cnt[tile], nb = *P.big;
    uint32_t px = tx * 32 + (q & 1) * 16 + (l & 7), py = ty * 32 + (q >> 1) * 16 + (l >> 3);
    uint32_t col = P.w[20];
    if (EXEC || (c | nb)) {
        float x = (float)(c + l), y0 = x, y1 = x + 1, y2 = x + 2, y3 = x + 3;
#pragma unroll
        for (int i = 0; i < N; i++) { y0 = y0 * 1.0001f + (float)(i * 7 + 1); y1 = y1 * 1.0002f + (float)(i * 5 + 2); y2 = y2 * 1.0003f + (float)(i * 3 + 3); y3 = y3 * 1.0004f + (float)(i + 4); }
        lds[threadIdx.x] = make_uint4((uint32_t)y0, (uint32_t)y1, (uint32_t)y2, (uint32_t)y3); __syncthreads(); col += lds[(threadIdx.x + 1) & 255].x;
    }
    for (int b = 0; b < 4; b++) { uint32_t x = px + (b & 1) * 8, y = py + (b >> 1) * 8; if (x < P.w[0] && y < P.w[1]) P.out[y * P.w[0] + x] = col + P.w[31 + b]; }
}
template <class F> void timeit(const char* name, F f, hipStream_t s, int n = 1000) {
    for (int i = 0; i < 20; i++) f();
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; i++) f();
    hipStreamSynchronize(s);
    auto t1 = std::chrono::steady_clock::now();
    printf("%-44s back-to-back %.2f us\n", name, std::chrono::duration<double, std::micro>(t1 - t0).count() / n);
}
int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    uint32_t* out; hipMalloc(&out, 64 << 20);
    uint32_t* cnt; hipMalloc(&cnt, 1 << 20); hipMemset(cnt, 0, 1 << 20);
    PP P{}; P.w[0] = 1920; P.w[1] = 1080; P.w[2] = 60; P.w[4] = 0; P.w[20] = 0xff202020; P.cnt = cnt; P.big = cnt + 4096; P.out = out; P.depth = nullptr;
#define T(N, E) timeit("N=" #N " exec=" #E, [&] { hipLaunchKernelGGL((kF<N, E>), dim3(2040), dim3(256), 0, s, P); }, s);
    T(8, false) T(256, false) T(1024, false) T(4096, false)
    T(8, true) T(64, true) T(256, true) T(1024, true)
    // alternate two different kernels (like geometry/raster) to see whether the instruction cache survives
    timeit("alternate N=1024(dead) / N=256(dead)", [&] { hipLaunchKernelGGL((kF<1024, false>), dim3(2040), dim3(256), 0, s, P); hipLaunchKernelGGL((kF<256, false>), dim3(157), dim3(256), 0, s, P); }, s);
    return 0;
}
