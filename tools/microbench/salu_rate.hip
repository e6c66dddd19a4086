// SALU issue rate and SALU/VALU co-issue on gfx950: cycles per instruction per SIMD with 4 and 8 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define R8(s) s "\n" s "\n" s "\n" s "\n" s "\n" s "\n" s "\n" s
template <int OP>
__global__ void k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t e0 = threadIdx.x + seed, e1 = e0 * 3u + 1u, e2 = e0 ^ 5u;
    uint32_t s0 = seed, s1 = seed * 7u;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if (OP == 0) asm volatile(R8("s_add_u32 %0, %0, %1") : "+s"(s0) : "s"(s1) : "scc");
            if (OP == 1) asm volatile(R8("s_and_b32 %0, %1, %2") : "+s"(s0) : "s"(s1), "s"(seed) : "scc");
            if (OP == 2) asm volatile(R8("v_add_u32 %0, %1, %2") : "+v"(e0) : "v"(e1), "v"(e2));
            if (OP == 3) asm volatile(R8("v_add_u32 %0, %2, %3\ns_and_b32 %1, %4, %5") : "+v"(e0), "+s"(s0) : "v"(e1), "v"(e2), "s"(s1), "s"(seed) : "scc");   // 16
            if (OP == 4) asm volatile(R8("v_or3_b32 %0, %2, %3, %2\ns_and_b32 %1, %4, %5") : "+v"(e0), "+s"(s0) : "v"(e1), "v"(e2), "s"(s1), "s"(seed) : "scc");   // 16
            if (OP == 5) asm volatile(R8("v_or3_b32 %0, %1, %2, %1") : "+v"(e0) : "v"(e1), "v"(e2));
            if (OP == 6) asm volatile(R8("s_ff1_i32_b64 %0, s[2:3]") : "+s"(s0) :: "scc");
            if (OP == 7) asm volatile(R8("s_cmp_eq_u32 %0, 0\ns_cbranch_scc1 1f\ns_nop 0\n1:") :: "s"(s1) : "scc");   // taken branch? s1 != 0 -> not taken
        }
    }
    if (e0 + s0 == 0x12345) out[2] = 1;
}
template <int OP> void run(const char* name, uint32_t* d, int w, int per = 64) {
    const int iters = 2000;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int wgs = 256 * (w > 4 ? w / 4 : 1), thr = 256 * (w > 4 ? 4 : w);
    hipLaunchKernelGGL(k<OP>, dim3(wgs), dim3(thr), 0, 0, d, 10, 1u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(wgs), dim3(thr), 0, 0, d, iters, 1u);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("%-36s %d waves/SIMD: %.2f cycles@2.4GHz per instr per SIMD\n", name, w, ms * 1e6 / ((double)iters * per) / w * 2.4);
}
int main() {
    uint32_t* d; (void)hipMalloc(&d, 64);
    for (int w : {4, 8}) {
        run<0>("s_add_u32 (dependent chain)", d, w); run<1>("s_and_b32 (independent)", d, w); run<2>("v_add_u32", d, w);
        run<3>("v_add_u32 + s_and_b32 interleaved", d, w, 128); run<5>("v_or3_b32", d, w); run<4>("v_or3_b32 + s_and_b32 interleaved", d, w, 128);
        run<6>("s_ff1_i32_b64", d, w); run<7>("s_cmp + s_cbranch (not taken) + nop", d, w, 192);
    }
    return 0;
}
