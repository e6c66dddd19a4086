// ieee_exact.hip -- checks csrc/mirhi_exact.hip.h against the compiler's IEEE expansions on the GPU.
//   rcp_rn(x)  == 1.0f / x   for EVERY binary32 x (both signs, all exponents, specials)
//   sqrt_rn(x) == sqrtf(x)   for EVERY binary32 x
//   div_rn(a, b) == a / b    for 2^33 pseudo-random pairs (exponents spread over the whole range) + all pairs of a boundary set
// NaN results compare equal if both are NaN.  Prints the number of mismatches (must be 0) and the first few.
// build: make -C tools/microbench ieee_exact     run on the GPU box: tools/microbench/ieee_exact
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
namespace mirhi {
#include "../../renderer-rs_amd/csrc/mirhi_exact.hip.h"
}
using namespace mirhi;

__device__ unsigned long long g_bad[6];
__device__ uint32_t g_first[6][8][3];
__device__ uint32_t g_nb_range[4] = {0xFFFFFFFFu, 0u, 0xFFFFFFFFu, 0u};   // smallest / largest |x| bits of a mismatch of the branch-free rcp, sqrt

__device__ __forceinline__ bool same(float a, float b) { return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b); }
__device__ void report(int which, uint32_t x, uint32_t y, float got) {
    const unsigned long long k = atomicAdd(&g_bad[which], 1ull);
    if (k < 8) { g_first[which][k][0] = x; g_first[which][k][1] = y; g_first[which][k][2] = __float_as_uint(got); }
}
__global__ void check_unary() {
    const uint32_t stride = gridDim.x * blockDim.x;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint64_t n = i; n < (1ull << 32); n += stride) {
        const float x = __uint_as_float((uint32_t)n);
        const float r0 = 1.0f / x, r1 = rcp_rn(x);
        if (!same(r0, r1)) report(0, (uint32_t)n, 0, r1);
        const float s0 = sqrtf(x), s1 = sqrt_rn(x);
        if (!same(s0, s1)) report(1, (uint32_t)n, 0, s1);
        const float r2 = rcp_rn_nb(x), s2 = sqrt_rn_nb(x);
        const uint32_t mag = (uint32_t)n & 0x7FFFFFFFu;
        // the branch-free forms: mismatches are expected only where an operand or a result is denormal; keep the range of |x| seen
        if (!same(r0, r2)) { report(3, (uint32_t)n, 0, r2); if (mag >= 0x00800000u && mag < 0x7E800000u) { atomicMin(&g_nb_range[0], mag); atomicMax(&g_nb_range[1], mag); } }
        if (!same(s0, s2)) { report(4, (uint32_t)n, 0, s2); if (mag >= 0x0D800000u) { atomicMin(&g_nb_range[2], mag); atomicMax(&g_nb_range[3], mag); } }   // (x >= 2^-100)
    }
}
__device__ __forceinline__ uint32_t pcg(uint64_t& st) {
    const uint64_t old = st;
    st = old * 6364136223846793005ULL + 1442695040888963407ULL;
    const uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u), rot = (uint32_t)(old >> 59u);
    return (xs >> rot) | (xs << ((32 - rot) & 31));
}
__global__ void check_div(uint32_t per_thread) {
    uint64_t st = 0x853c49e6748fea9bULL + 0x9E3779B97F4A7C15ULL * (blockIdx.x * blockDim.x + threadIdx.x);
    for (uint32_t k = 0; k < per_thread; k++) {
        const uint32_t ua = pcg(st), ub = pcg(st);
        const float a = __uint_as_float(ua), b = __uint_as_float(ub);
        const float q0 = a / b, q1 = div_rn(a, b);
        if (!same(q0, q1)) report(2, ua, ub, q1);
        // the same mantissas with moderate exponents (what vector lengths look like): exercises the fast path every time
        const uint32_t ma = (ua & 0x807FFFFFu) | ((100u + (ua >> 23) % 56u) << 23), mb = (ub & 0x807FFFFFu) | ((100u + (ub >> 23) % 56u) << 23);
        const float c = __uint_as_float(ma), d = __uint_as_float(mb);
        const float p0 = c / d, p1 = div_rn(c, d);
        if (!same(p0, p1)) report(2, ma, mb, p1);
        const float p2 = div_rn_nb(c, d);               // the branch-free form the fragment programs use, on the same moderate operands
        if (!same(p0, p2)) report(5, ma, mb, p2);
    }
}
int main() {
    unsigned long long zero[6] = {0, 0, 0, 0, 0, 0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_bad), zero, sizeof zero);
    hipLaunchKernelGGL(check_unary, dim3(4096), dim3(256), 0, 0);
    hipLaunchKernelGGL(check_div, dim3(8192), dim3(256), 0, 0, 2048u);      // 2^21 threads x 2048 x 2 = 2^33 pairs
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
    unsigned long long bad[6]; uint32_t first[6][8][3]; uint32_t nb[4];
    hipMemcpyFromSymbol(bad, HIP_SYMBOL(g_bad), sizeof bad);
    hipMemcpyFromSymbol(first, HIP_SYMBOL(g_first), sizeof first);
    hipMemcpyFromSymbol(nb, HIP_SYMBOL(g_nb_range), sizeof nb);
    const char* names[6] = {"rcp_rn vs 1/x (2^32 inputs)", "sqrt_rn vs sqrtf (2^32 inputs)", "div_rn vs a/b (2^33 pairs)",
                            "rcp_rn_nb vs 1/x (2^32 inputs; denormal operands / results may differ)", "sqrt_rn_nb vs sqrtf (2^32 inputs; operands below 2^-100 may differ)",
                            "div_rn_nb vs a/b (2^32 pairs with exponents in [-27, 28]: must be 0)"};
    int rc = 0;
    printf("branch-free forms: mismatches inside the stated range (1/x: normal x, |x| < 2^126; sqrt: x >= 2^-100): rcp |x| bits [%08x, %08x], sqrt |x| bits [%08x, %08x]  (ffffffff, 0 = none)\n", nb[0], nb[1], nb[2], nb[3]);
    if (nb[1] != 0u || nb[3] != 0u) rc = 1;
    for (int w = 0; w < 6; w++) {
        printf("%s: %llu mismatches\n", names[w], bad[w]);
        for (unsigned long long k = 0; k < bad[w] && k < 8; k++) printf("   x=%08x y=%08x got=%08x\n", first[w][k][0], first[w][k][1], first[w][k][2]);
        if (bad[w] && (w < 3 || w == 5)) rc = 1;
    }
    return rc;
}
