// host cost of launching two dependent small kernels: two hipLaunchKernelGGL calls vs one hipGraphLaunch of the captured pair
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void ka(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void kb(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[1] += 1; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    int* d; (void)hipMalloc(&d, 64); (void)hipMemset(d, 0, 64);
    const int lanes = 4, n = 20000;
    hipStream_t st[lanes]; hipGraphExec_t ge[lanes];
    for (int i = 0; i < lanes; i++) {
        (void)hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
        hipGraph_t g;
        (void)hipStreamBeginCapture(st[i], hipStreamCaptureModeThreadLocal);
        hipLaunchKernelGGL(ka, dim3(157), dim3(64), 0, st[i], d);
        hipLaunchKernelGGL(kb, dim3(60, 34), dim3(256), 0, st[i], d);
        (void)hipStreamEndCapture(st[i], &g);
        (void)hipGraphInstantiate(&ge[i], g, nullptr, nullptr, 0);
    }
    for (int rep = 0; rep < 2; rep++) {
        (void)hipDeviceSynchronize();
        double t0 = now();
        for (int i = 0; i < n; i++) {
            hipLaunchKernelGGL(ka, dim3(157), dim3(64), 0, st[i % lanes], d);
            hipLaunchKernelGGL(kb, dim3(60, 34), dim3(256), 0, st[i % lanes], d);
        }
        double t1 = now();
        (void)hipDeviceSynchronize();
        double t2 = now();
        printf("2 launches : host %.2f us per pair, %.2f us incl. drain\n", 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n);
        t0 = now();
        for (int i = 0; i < n; i++) (void)hipGraphLaunch(ge[i % lanes], st[i % lanes]);
        t1 = now();
        (void)hipDeviceSynchronize();
        t2 = now();
        printf("graph launch: host %.2f us per pair, %.2f us incl. drain\n", 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n);
    }
    return 0;
}
