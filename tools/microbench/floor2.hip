#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
struct PP { uint32_t w[40]; uint32_t* cnt; uint32_t* big; uint32_t* out; float* depth; };
__global__ __launch_bounds__(256) void kA(uint32_t* out, uint32_t W) {
    uint32_t tile = blockIdx.x, tx = tile % 60, ty = tile / 60, l = threadIdx.x & 63, q = threadIdx.x >> 6;
    uint32_t px = tx * 32 + (q & 1) * 16 + (l & 7), py = ty * 32 + (q >> 1) * 16 + (l >> 3);
    for (int b = 0; b < 4; b++) { uint32_t x = px + (b & 1) * 8, y = py + (b >> 1) * 8; if (x < W && y < 1080) out[y * W + x] = 0xff202020; }
}
__global__ __launch_bounds__(256) void kB(uint32_t* out, uint32_t W, const uint32_t* cnt, const uint32_t* big) {
    uint32_t tile = blockIdx.x, tx = tile % 60, ty = tile / 60, l = threadIdx.x & 63, q = threadIdx.x >> 6;
    uint32_t c = cnt[tile], nb = *big;
    uint32_t px = tx * 32 + (q & 1) * 16 + (l & 7), py = ty * 32 + (q >> 1) * 16 + (l >> 3);
    uint32_t col = 0xff202020;
    if (c | nb) col = c + nb;
    for (int b = 0; b < 4; b++) { uint32_t x = px + (b & 1) * 8, y = py + (b >> 1) * 8; if (x < W && y < 1080) out[y * W + x] = col; }
}
__global__ __launch_bounds__(256) void kC(uint32_t* out, uint32_t W, const uint32_t* cnt, const uint32_t* big) {
    __shared__ uint4 lds[1024 + 70];
    uint32_t tile = blockIdx.x, tx = tile % 60, ty = tile / 60, l = threadIdx.x & 63, q = threadIdx.x >> 6;
    uint32_t c = cnt[tile], nb = *big;
    uint32_t px = tx * 32 + (q & 1) * 16 + (l & 7), py = ty * 32 + (q >> 1) * 16 + (l >> 3);
    uint32_t col = 0xff202020;
    if (c | nb) { lds[threadIdx.x] = make_uint4(c, nb, 0, 0); __syncthreads(); col = lds[(threadIdx.x + 1) & 255].x; }
    for (int b = 0; b < 4; b++) { uint32_t x = px + (b & 1) * 8, y = py + (b >> 1) * 8; if (x < W && y < 1080) out[y * W + x] = col; }
}
__global__ __launch_bounds__(256) void kD(PP P) {
    __shared__ uint4 lds[1024 + 70];
    uint32_t tile = blockIdx.x, tx = tile % P.w[2], ty = P.w[4] + tile / P.w[2], l = threadIdx.x & 63, q = threadIdx.x >> 6;
    uint32_t c = P.cnt[tile], nb = *P.big;
    uint32_t px = tx * 32 + (q & 1) * 16 + (l & 7), py = ty * 32 + (q >> 1) * 16 + (l >> 3);
    uint32_t col = P.w[20];
    if (P.w[30] && P.depth) col += (uint32_t)P.depth[py * P.w[0] + px];
    if (c | nb) { lds[threadIdx.x] = make_uint4(c, nb, 0, 0); __syncthreads(); col = lds[(threadIdx.x + 1) & 255].x; }
    for (int b = 0; b < 4; b++) { uint32_t x = px + (b & 1) * 8, y = py + (b >> 1) * 8; if (x < P.w[0] && y < P.w[1]) P.out[y * P.w[0] + x] = col + P.w[31 + b]; }
}
__global__ __launch_bounds__(256) void kE(PP P) {   // + scratch
    __shared__ uint4 lds[1024 + 70];
    volatile uint32_t sc[16];
    uint32_t tile = blockIdx.x, tx = tile % P.w[2], ty = P.w[4] + tile / P.w[2], l = threadIdx.x & 63, q = threadIdx.x >> 6;
    uint32_t c = P.cnt[tile], nb = *P.big;
    for (int i = 0; i < 16; i++) sc[i] = c + i;
    uint32_t px = tx * 32 + (q & 1) * 16 + (l & 7), py = ty * 32 + (q >> 1) * 16 + (l >> 3);
    uint32_t col = P.w[20] + sc[(c + 3) & 15];
    if (c | nb) { lds[threadIdx.x] = make_uint4(c, nb, 0, 0); __syncthreads(); col = lds[(threadIdx.x + 1) & 255].x; }
    for (int b = 0; b < 4; b++) { uint32_t x = px + (b & 1) * 8, y = py + (b >> 1) * 8; if (x < P.w[0] && y < P.w[1]) P.out[y * P.w[0] + x] = col + P.w[31 + b]; }
}
template <class F> void timeit(const char* name, F f, hipStream_t s, int n = 1000) {
    for (int i = 0; i < 20; i++) f();
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; i++) f();
    hipStreamSynchronize(s);
    auto t1 = std::chrono::steady_clock::now();
    printf("%-36s back-to-back %.2f us\n", name, std::chrono::duration<double, std::micro>(t1 - t0).count() / n);
}
int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    uint32_t* out; hipMalloc(&out, 64 << 20);
    uint32_t* cnt; hipMalloc(&cnt, 1 << 20); hipMemset(cnt, 0, 1 << 20);
    PP P{}; P.w[0] = 1920; P.w[1] = 1080; P.w[2] = 60; P.w[4] = 0; P.w[20] = 0xff202020; P.cnt = cnt; P.big = cnt + 4096; P.out = out; P.depth = nullptr;
    timeit("A store 4px/thread", [&] { hipLaunchKernelGGL(kA, dim3(2040), dim3(256), 0, s, out, 1920u); }, s);
    timeit("B +2 counter loads", [&] { hipLaunchKernelGGL(kB, dim3(2040), dim3(256), 0, s, out, 1920u, cnt, cnt + 4096); }, s);
    timeit("C +17KB LDS", [&] { hipLaunchKernelGGL(kC, dim3(2040), dim3(256), 0, s, out, 1920u, cnt, cnt + 4096); }, s);
    timeit("D +176B kernarg struct", [&] { hipLaunchKernelGGL(kD, dim3(2040), dim3(256), 0, s, P); }, s);
    timeit("E +scratch 64B", [&] { hipLaunchKernelGGL(kE, dim3(2040), dim3(256), 0, s, P); }, s);
    timeit("memset 4B + A", [&] { hipMemsetAsync(cnt, 0, 4, s); hipLaunchKernelGGL(kA, dim3(2040), dim3(256), 0, s, out, 1920u); }, s);
    uint32_t* hp; hipHostMalloc(&hp, 64);
    timeit("A + D2H 4B copy", [&] { hipLaunchKernelGGL(kA, dim3(2040), dim3(256), 0, s, out, 1920u); hipMemcpyAsync(hp, cnt, 4, hipMemcpyDeviceToHost, s); }, s);
    return 0;
}
