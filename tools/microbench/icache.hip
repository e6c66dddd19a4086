// how much does straight-line code size cost per launch? (cold instruction cache at every dispatch?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
template <int N> __global__ void k_code(float* out, float a) {
    float x = a + threadIdx.x;
#pragma unroll
    for (int i = 0; i < N; i++) { x = x * 1.0001f + (float)(i * 7 + 1); x = x - (float)(i * 3); }   // 2 v_ ops with distinct literals: ~16 B each pair
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
template <class F> void timeit(const char* name, F f, hipStream_t s, int n = 500) {
    for (int i = 0; i < 20; i++) f();
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; i++) f();
    hipStreamSynchronize(s);
    auto t1 = std::chrono::steady_clock::now();
    printf("%-28s back-to-back %.2f us\n", name, std::chrono::duration<double, std::micro>(t1 - t0).count() / n);
}
int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    float* out; hipMalloc(&out, 64 << 20);
#define T(N, G) { char nm[64]; snprintf(nm, 64, "code N=%d grid %d", N, G); timeit(nm, [&] { hipLaunchKernelGGL(k_code<N>, dim3(G), dim3(256), 0, s, out, 1.0f); }, s); }
    T(16, 2040) T(128, 2040) T(512, 2040) T(2048, 2040) T(8192, 2040)
    T(16, 64) T(512, 64) T(2048, 64) T(8192, 64)
    return 0;
}
