// second VALU issue-rate probe: select / compare forms (is v_cndmask_b32 really slow?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define R8(s) s "\n" s "\n" s "\n" s "\n" s "\n" s "\n" s "\n" s
template <int OP>
__global__ void k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a = threadIdx.x + seed, b = a * 3u + 1u;
    uint32_t e0 = a, e1 = b, e2 = a ^ 5u, e3 = b + 9u;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if (OP == 0) asm volatile(R8("v_cndmask_b32 %0, %1, %2, vcc") : "+v"(e0) : "v"(e1), "v"(e2) : );
            if (OP == 1) asm volatile(R8("v_cndmask_b32_e64 %0, %1, %2, s[20:21]") : "+v"(e0) : "v"(e1), "v"(e2) : "s20", "s21");
            if (OP == 2) asm volatile("v_cndmask_b32 %0, %0, %4, vcc\nv_cndmask_b32 %1, %1, %4, vcc\nv_cndmask_b32 %2, %2, %4, vcc\nv_cndmask_b32 %3, %3, %4, vcc\nv_cndmask_b32 %0, %0, %4, vcc\nv_cndmask_b32 %1, %1, %4, vcc\nv_cndmask_b32 %2, %2, %4, vcc\nv_cndmask_b32 %3, %3, %4, vcc" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3) : "v"(a));
            if (OP == 3) asm volatile(R8("v_mov_b32 %0, %1") : "+v"(e0) : "v"(e1));
            if (OP == 4) asm volatile(R8("v_and_b32 %0, %1, %2") : "+v"(e0) : "v"(e1), "v"(e2));
            if (OP == 5) asm volatile(R8("v_lshlrev_b32 %0, 3, %1") : "+v"(e0) : "v"(e1));
            if (OP == 6) asm volatile(R8("v_add_f32 %0, %1, %2") : "+v"(e0) : "v"(e1), "v"(e2));
            if (OP == 7) asm volatile(R8("v_cmp_lt_i32_e64 s[20:21], %0, %1") :: "v"(e0), "v"(e1) : "s20", "s21");
            if (OP == 8) asm volatile(R8("v_cmp_lt_i32 vcc, %0, %1") :: "v"(e0), "v"(e1) : "vcc");
            if (OP == 9) asm volatile(R8("v_cmp_lt_i32 vcc, %1, %2\nv_cndmask_b32 %0, %1, %2, vcc") : "+v"(e0) : "v"(e1), "v"(e2) : "vcc");      // 16 instr
            if (OP == 10) asm volatile(R8("v_max_u32 %0, %1, %2") : "+v"(e0) : "v"(e1), "v"(e2));
            if (OP == 11) asm volatile(R8("v_add_u32 %0, %1, %2") : "+v"(e0) : "v"(e1), "v"(e2));
            if (OP == 12) asm volatile(R8("v_fma_f32 %0, %1, %2, %1") : "+v"(e0) : "v"(e1), "v"(e2));
            if (OP == 13) asm volatile(R8("v_mad_u32_u24 %0, %1, %2, %1") : "+v"(e0) : "v"(e1), "v"(e2));
            if (OP == 14) asm volatile(R8("v_add3_u32 %0, %1, %2, %1") : "+v"(e0) : "v"(e1), "v"(e2));
            if (OP == 15) asm volatile(R8("v_lshl_add_u32 %0, %1, 3, %2") : "+v"(e0) : "v"(e1), "v"(e2));
            if (OP == 16) asm volatile(R8("v_med3_f32 %0, %1, 0, 1.0") : "+v"(e0) : "v"(e1));
            if (OP == 17) asm volatile(R8("v_mul_f32 %0, %1, %2") : "+v"(e0) : "v"(e1), "v"(e2));
            if (OP == 18) asm volatile(R8("v_sub_u32 %0, %1, %2") : "+v"(e0) : "v"(e1), "v"(e2));
            if (OP == 19) asm volatile(R8("v_xor_b32 %0, %1, %2") : "+v"(e0) : "v"(e1), "v"(e2));
            if (OP == 20) asm volatile(R8("v_cvt_f32_i32 %0, %1") : "+v"(e0) : "v"(e1));
            if (OP == 21) asm volatile(R8("v_bfe_u32 %0, %1, 3, 5") : "+v"(e0) : "v"(e1));
        }
    }
    if (e0 + e1 + e2 + e3 == 0x12345) out[2] = 1;
}
template <int OP> void run(const char* name, uint32_t* d, int per = 64) {
    const int iters = 2000, w = 4;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256 * w), 0, 0, d, 10, 1u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256 * w), 0, 0, d, iters, 1u);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double n = (double)iters * per;
    printf("%-34s %.2f cycles@2.4GHz per instr per SIMD (4 waves/SIMD)\n", name, ms * 1e6 / n / w * 2.4);
}
int main() {
    uint32_t* d; (void)hipMalloc(&d, 64);
    run<0>("v_cndmask_b32 vcc (dst != src)", d); run<1>("v_cndmask_b32_e64 sgpr mask", d); run<2>("v_cndmask_b32 vcc (dst == src0)", d);
    run<3>("v_mov_b32", d); run<4>("v_and_b32", d); run<5>("v_lshlrev_b32", d); run<6>("v_add_f32", d);
    run<7>("v_cmp_lt_i32_e64 -> sgpr", d); run<8>("v_cmp_lt_i32 -> vcc", d); run<9>("v_cmp + v_cndmask pair (per instr)", d, 128);
    run<10>("v_max_u32", d); run<11>("v_add_u32 3-operand", d); run<12>("v_fma_f32", d); run<13>("v_mad_u32_u24", d); run<14>("v_add3_u32", d);
    run<15>("v_lshl_add_u32", d); run<16>("v_med3_f32", d); run<17>("v_mul_f32", d); run<18>("v_sub_u32", d); run<19>("v_xor_b32", d);
    run<20>("v_cvt_f32_i32", d); run<21>("v_bfe_u32", d);
    return 0;
}
