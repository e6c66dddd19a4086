// host cost of the ways HIP offers to launch one kernel with a ~100-byte argument block (what mirhi_queue_submit pays twice per frame)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
struct Args { const void* p; uint32_t w[22]; };          // 96 bytes
__global__ void k(const void* params, Args a) { if (threadIdx.x == 9999 && params) ((volatile uint32_t*)a.p)[0] = a.w[3]; }
int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    void* d; CK(hipMalloc(&d, 4096));
    Args a{}; a.p = d;
    const int n = 40000;
    for (int rep = 0; rep < 2; rep++) {
        CK(hipStreamSynchronize(st));
        double t0 = now();
        for (int i = 0; i < n; i++) hipLaunchKernelGGL(k, dim3(157), dim3(64), 0, st, (const void*)d, a);
        double t1 = now(); CK(hipStreamSynchronize(st));
        printf("hipLaunchKernelGGL            : host %.2f us (%.2f incl. drain)\n", 1e6 * (t1 - t0) / n, 1e6 * (now() - t0) / n);
        void* args[2] = {(void*)&d, (void*)&a};
        t0 = now();
        for (int i = 0; i < n; i++) CK(hipLaunchKernel((const void*)k, dim3(157), dim3(64), args, 0, st));
        t1 = now(); CK(hipStreamSynchronize(st));
        printf("hipLaunchKernel               : host %.2f us (%.2f incl. drain)\n", 1e6 * (t1 - t0) / n, 1e6 * (now() - t0) / n);
        hipFunction_t f; CK(hipGetFuncBySymbol(&f, (const void*)k));
        t0 = now();
        for (int i = 0; i < n; i++) CK(hipModuleLaunchKernel(f, 157, 1, 1, 64, 1, 1, 0, st, args, nullptr));
        t1 = now(); CK(hipStreamSynchronize(st));
        printf("hipModuleLaunchKernel(params) : host %.2f us (%.2f incl. drain)\n", 1e6 * (t1 - t0) / n, 1e6 * (now() - t0) / n);
        struct { const void* p; Args a; } packed{d, a};
        size_t sz = sizeof packed;
        void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &packed, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
        t0 = now();
        for (int i = 0; i < n; i++) CK(hipModuleLaunchKernel(f, 157, 1, 1, 64, 1, 1, 0, st, nullptr, extra));
        t1 = now(); CK(hipStreamSynchronize(st));
        printf("hipModuleLaunchKernel(extra)  : host %.2f us (%.2f incl. drain)\n", 1e6 * (t1 - t0) / n, 1e6 * (now() - t0) / n);
        hipEvent_t ev[4]; for (auto& e : ev) CK(hipEventCreate(&e));
        t0 = now();
        for (int i = 0; i < n; i++) CK(hipExtModuleLaunchKernel(f, 157 * 64, 1, 1, 64, 1, 1, 0, st, nullptr, extra, nullptr, ev[i & 3], 0));
        t1 = now(); CK(hipStreamSynchronize(st));
        printf("hipExtModuleLaunchKernel+stop : host %.2f us (%.2f incl. drain)\n", 1e6 * (t1 - t0) / n, 1e6 * (now() - t0) / n);
        t0 = now();
        for (int i = 0; i < n; i++) hipExtLaunchKernelGGL(k, dim3(157), dim3(64), 0, st, nullptr, ev[i & 3], 0, (const void*)d, a);
        t1 = now(); CK(hipStreamSynchronize(st));
        printf("hipExtLaunchKernelGGL+stop    : host %.2f us (%.2f incl. drain)\n", 1e6 * (t1 - t0) / n, 1e6 * (now() - t0) / n);
    }
    return 0;
}
