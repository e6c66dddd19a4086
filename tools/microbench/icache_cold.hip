// What does a wave pay for running code its CU's instruction cache has not seen in this dispatch?  One wave per workgroup (W workgroups), a straight-line
// stream of N VALU instructions (8 bytes each) executed twice in a loop: pass 1 fetches the code, pass 2 finds it in the instruction cache.  Also: the same
// kernel launched again right away (is the cache kept across dispatches?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <int N>
__global__ __launch_bounds__(64) void k(uint32_t* out, uint64_t* t) {
    uint32_t a = threadIdx.x, b = blockIdx.x;
    uint64_t s[3];
    for (int pass = 0; pass < 2; pass++) {
        s[pass] = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < N; i++) asm volatile("v_add_u32 %0, %0, %1\n\tv_xor_b32 %1, %1, %0" : "+v"(a), "+v"(b));      // 2 x 4-byte... (VOP2: 4 bytes each)
        asm volatile("s_nop 0" ::: "memory");
    }
    s[2] = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = a ^ b;
    if (threadIdx.x == 0) { t[blockIdx.x * 2] = s[1] - s[0]; t[blockIdx.x * 2 + 1] = s[2] - s[1]; }
}
template <int N> void run(uint32_t W, uint32_t* out, uint64_t* t) {
    for (int launch = 0; launch < 3; launch++) {
        hipLaunchKernelGGL(k<N>, dim3(W), dim3(64), 0, 0, out, t);
        CK(hipDeviceSynchronize());
        std::vector<uint64_t> h(2 * W); CK(hipMemcpy(h.data(), t, 2 * W * 8, hipMemcpyDeviceToHost));
        std::vector<uint64_t> p1, p2; for (uint32_t i = 0; i < W; i++) { p1.push_back(h[2 * i]); p2.push_back(h[2 * i + 1]); }
        std::sort(p1.begin(), p1.end()); std::sort(p2.begin(), p2.end());
        printf("%5d instructions (%3d KB), %4u waves, launch %d: first pass median %6llu ticks max %6llu, second pass median %6llu ticks\n", 2 * N, 2 * N * 4 / 1024, W, launch,
               (unsigned long long)p1[W / 2], (unsigned long long)p1[W - 1], (unsigned long long)p2[W / 2]);
    }
}
int main() {
    uint32_t* out; uint64_t* t; CK(hipMalloc(&out, 4096 * 64 * 4)); CK(hipMalloc(&t, 4096 * 16));
    run<256>(157, out, t); run<1024>(157, out, t); run<4096>(157, out, t); run<1024>(625, out, t); run<1024>(1, out, t);
    return 0;
}
