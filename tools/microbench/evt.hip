#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty() {}
__global__ void k_spin(unsigned long long ticks) { unsigned long long t0 = __builtin_amdgcn_s_memrealtime(); while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {} }
int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 5; mode++) {
        double sum = 0; int n = 200;
        for (int i = 0; i < n + 20; i++) {
            hipEventRecord(a, s);
            if (mode == 1) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s);
            if (mode == 2) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, 1000ull);   // 10 us at 100 MHz
            if (mode == 3) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, 3000ull);   // 30 us
            if (mode == 4) hipLaunchKernelGGL(k_spin, dim3(2040), dim3(256), 0, s, 3000ull);
            hipEventRecord(b, s);
            hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (i >= 20) sum += ms;
        }
        const char* names[] = {"nothing", "empty kernel", "spin 10us", "spin 30us", "spin 30us 2040x256"};
        printf("%-22s event pair %.2f us\n", names[mode], 1e3 * sum / n);
    }
    // back-to-back (no sync between pairs), like the library does
    const int n = 200; hipEvent_t ea[n], eb[n];
    for (int i = 0; i < n; i++) { hipEventCreate(&ea[i]); hipEventCreate(&eb[i]); }
    for (int mode = 0; mode < 3; mode++) {
        for (int i = 0; i < n; i++) { hipEventRecord(ea[i], s); if (mode == 1) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, 3000ull); if (mode == 2) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); hipEventRecord(eb[i], s); }
        hipStreamSynchronize(s);
        double sum = 0; for (int i = 0; i < n; i++) { float ms; hipEventElapsedTime(&ms, ea[i], eb[i]); sum += ms; }
        printf("pipelined %-12s event pair %.2f us\n", mode == 0 ? "nothing" : (mode == 1 ? "spin 30us" : "empty"), 1e3 * sum / n);
    }
    return 0;
}
