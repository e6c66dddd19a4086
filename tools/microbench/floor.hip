// micro-benchmark: kernel duration floor on this box (hipEvent pairs vs back-to-back wall)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#include <vector>
struct Big { uint32_t w[48]; void* p[4]; };
__global__ void k_empty() {}
__global__ void k_store(uint32_t* out, uint32_t n) { uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) out[i] = i; }
__global__ void k_bigarg(Big b, uint32_t* out) { uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; out[i] = b.w[i & 31] + b.w[47]; }
__global__ void k_chain(const uint32_t* __restrict__ idx, uint32_t* out, int depth) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; uint32_t v = i & 1023;
    for (int d = 0; d < depth; d++) v = idx[v];
    out[i] = v;
}
template <class F> void timeit(const char* name, F f, hipStream_t s, int n = 500) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; i++) f();
    hipStreamSynchronize(s);
    double evsum = 0;
    for (int i = 0; i < n; i++) { hipEventRecord(a, s); f(); hipEventRecord(b, s); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); evsum += ms; }
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; i++) f();
    hipStreamSynchronize(s);
    auto t1 = std::chrono::steady_clock::now();
    printf("%-28s event %.2f us   back-to-back %.2f us\n", name, 1e3 * evsum / n, std::chrono::duration<double, std::micro>(t1 - t0).count() / n);
}
int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    uint32_t* out; hipMalloc(&out, 64 << 20);
    uint32_t* idx; hipMalloc(&idx, 4096);
    std::vector<uint32_t> h(1024); for (int i = 0; i < 1024; i++) h[i] = (i * 37 + 11) & 1023;
    hipMemcpy(idx, h.data(), 4096, hipMemcpyHostToDevice);
    Big big{}; 
    timeit("empty 1x64", [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); }, s);
    timeit("empty 2040x256", [&] { hipLaunchKernelGGL(k_empty, dim3(2040), dim3(256), 0, s); }, s);
    timeit("store 2040x256 (2MB)", [&] { hipLaunchKernelGGL(k_store, dim3(2040), dim3(256), 0, s, out, 2040u * 256u); }, s);
    timeit("store 8100x256 (8.3MB)", [&] { hipLaunchKernelGGL(k_store, dim3(8100), dim3(256), 0, s, out, 8100u * 256u); }, s);
    timeit("bigarg 2040x256", [&] { hipLaunchKernelGGL(k_bigarg, dim3(2040), dim3(256), 0, s, big, out); }, s);
    for (int d : {1, 2, 4, 8}) { char nm[64]; snprintf(nm, 64, "chain depth %d 2040x256", d); timeit(nm, [&] { hipLaunchKernelGGL(k_chain, dim3(2040), dim3(256), 0, s, idx, out, d); }, s); }
    timeit("memsetAsync 4B", [&] { hipMemsetAsync(out, 0, 4, s); }, s);
    timeit("2 kernels", [&] { hipLaunchKernelGGL(k_empty, dim3(157), dim3(64), 0, s); hipLaunchKernelGGL(k_store, dim3(2040), dim3(256), 0, s, out, 2040u * 256u); }, s);
    // graph of 8 x (2 kernels)
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 8; i++) { hipLaunchKernelGGL(k_empty, dim3(157), dim3(64), 0, s); hipLaunchKernelGGL(k_store, dim3(2040), dim3(256), 0, s, out, 2040u * 256u); }
    hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    timeit("graph 8x(2 kernels)", [&] { hipGraphLaunch(ge, s); }, s, 200);
    return 0;
}
