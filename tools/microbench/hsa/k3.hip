// device code of stale.cpp: every workgroup reads the whole (small) buffer, so its lines sit in the L2 of every XCD afterwards
#include <hip/hip_runtime.h>
#include <stdint.h>
struct Args { const uint32_t* x; uint32_t n; uint32_t expect; uint32_t* bad; };
extern "C" __global__ __launch_bounds__(256) void check_kernel(Args a) {
    uint32_t wrong = 0;
    for (uint32_t i = threadIdx.x; i < a.n; i += 256) wrong += a.x[i] != a.expect;
    if (wrong) atomicAdd(a.bad, wrong);
}
extern "C" __global__ __launch_bounds__(256) void fill_kernel(uint32_t* x, uint32_t n, uint32_t v) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) x[i] = v;
}
