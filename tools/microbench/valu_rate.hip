// VALU issue-rate microbenchmark: cycles per wave64 instruction for the ops of the raster inner loop.
// One workgroup of 256 lanes (one wave per SIMD) per CU; each wave runs N dependent-free instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 64
template <int OP>
__global__ void k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a = threadIdx.x + seed, b = a * 3u + 1u, c = a ^ 0x55u, d = a + 7u;
    uint32_t e0 = a, e1 = b, e2 = c, e3 = d, e4 = a + 1, e5 = b + 1, e6 = c + 1, e7 = d + 1;
    unsigned long long acc = 0;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < REP / 8; r++) {
            if (OP == 0) {        // v_add_u32
                asm volatile("v_add_u32 %0, %0, %8\nv_add_u32 %1, %1, %8\nv_add_u32 %2, %2, %8\nv_add_u32 %3, %3, %8\nv_add_u32 %4, %4, %8\nv_add_u32 %5, %5, %8\nv_add_u32 %6, %6, %8\nv_add_u32 %7, %7, %8"
                    : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5), "+v"(e6), "+v"(e7) : "v"(a));
            } else if (OP == 1) { // v_mad_i32_i24
                asm volatile("v_mad_i32_i24 %0, %8, %9, %0\nv_mad_i32_i24 %1, %8, %9, %1\nv_mad_i32_i24 %2, %8, %9, %2\nv_mad_i32_i24 %3, %8, %9, %3\nv_mad_i32_i24 %4, %8, %9, %4\nv_mad_i32_i24 %5, %8, %9, %5\nv_mad_i32_i24 %6, %8, %9, %6\nv_mad_i32_i24 %7, %8, %9, %7"
                    : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5), "+v"(e6), "+v"(e7) : "v"(a), "v"(b));
            } else if (OP == 2) { // v_cmp_lt_u64
                asm volatile("v_cmp_lt_u64 vcc, %0, %1\nv_cmp_lt_u64 vcc, %2, %3\nv_cmp_lt_u64 vcc, %0, %1\nv_cmp_lt_u64 vcc, %2, %3\nv_cmp_lt_u64 vcc, %0, %1\nv_cmp_lt_u64 vcc, %2, %3\nv_cmp_lt_u64 vcc, %0, %1\nv_cmp_lt_u64 vcc, %2, %3"
                    :: "v"(((unsigned long long)e0 << 32) | e1), "v"(((unsigned long long)e2 << 32) | e3), "v"(((unsigned long long)e4 << 32) | e5), "v"(((unsigned long long)e6 << 32) | e7) : "vcc");
            } else if (OP == 3) { // v_cmp_lt_u32
                asm volatile("v_cmp_lt_u32 vcc, %0, %1\nv_cmp_lt_u32 vcc, %2, %3\nv_cmp_lt_u32 vcc, %0, %1\nv_cmp_lt_u32 vcc, %2, %3\nv_cmp_lt_u32 vcc, %0, %1\nv_cmp_lt_u32 vcc, %2, %3\nv_cmp_lt_u32 vcc, %0, %1\nv_cmp_lt_u32 vcc, %2, %3"
                    :: "v"(e0), "v"(e1), "v"(e2), "v"(e3) : "vcc");
            } else if (OP == 4) { // v_or3_b32
                asm volatile("v_or3_b32 %0, %0, %8, %9\nv_or3_b32 %1, %1, %8, %9\nv_or3_b32 %2, %2, %8, %9\nv_or3_b32 %3, %3, %8, %9\nv_or3_b32 %4, %4, %8, %9\nv_or3_b32 %5, %5, %8, %9\nv_or3_b32 %6, %6, %8, %9\nv_or3_b32 %7, %7, %8, %9"
                    : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5), "+v"(e6), "+v"(e7) : "v"(a), "v"(b));
            } else if (OP == 5) { // v_fma_f32 clamp
                asm volatile("v_fma_f32 %0, %8, %9, %0 clamp\nv_fma_f32 %1, %8, %9, %1 clamp\nv_fma_f32 %2, %8, %9, %2 clamp\nv_fma_f32 %3, %8, %9, %3 clamp\nv_fma_f32 %4, %8, %9, %4 clamp\nv_fma_f32 %5, %8, %9, %5 clamp\nv_fma_f32 %6, %8, %9, %6 clamp\nv_fma_f32 %7, %8, %9, %7 clamp"
                    : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5), "+v"(e6), "+v"(e7) : "v"(a), "v"(b));
            } else if (OP == 6) { // v_cndmask_b32 (vcc)
                asm volatile("v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\nv_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc"
                    : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5), "+v"(e6), "+v"(e7) : "v"(a) : "vcc");
            } else if (OP == 7) { // v_pk_add_f32 (2 floats per lane)
                unsigned long long p0 = ((unsigned long long)e0 << 32) | e1, p1 = ((unsigned long long)e2 << 32) | e3;
                asm volatile("v_pk_add_f32 %0, %0, %2\nv_pk_add_f32 %1, %1, %2\nv_pk_add_f32 %0, %0, %2\nv_pk_add_f32 %1, %1, %2\nv_pk_add_f32 %0, %0, %2\nv_pk_add_f32 %1, %1, %2\nv_pk_add_f32 %0, %0, %2\nv_pk_add_f32 %1, %1, %2"
                    : "+v"(p0), "+v"(p1) : "v"(((unsigned long long)a << 32) | b));
                acc += p0 + p1;
            } else if (OP == 8) { // v_min_u32 pair emulating key min? v_min_u32
                asm volatile("v_min_u32 %0, %0, %8\nv_min_u32 %1, %1, %8\nv_min_u32 %2, %2, %8\nv_min_u32 %3, %3, %8\nv_min_u32 %4, %4, %8\nv_min_u32 %5, %5, %8\nv_min_u32 %6, %6, %8\nv_min_u32 %7, %7, %8"
                    : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5), "+v"(e6), "+v"(e7) : "v"(a));
            } else if (OP == 9) { // v_mul_lo_u32
                asm volatile("v_mul_lo_u32 %0, %0, %8\nv_mul_lo_u32 %1, %1, %8\nv_mul_lo_u32 %2, %2, %8\nv_mul_lo_u32 %3, %3, %8\nv_mul_lo_u32 %4, %4, %8\nv_mul_lo_u32 %5, %5, %8\nv_mul_lo_u32 %6, %6, %8\nv_mul_lo_u32 %7, %7, %8"
                    : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5), "+v"(e6), "+v"(e7) : "v"(a));
            } else if (OP == 10) { // ds_read_b128 broadcast (all lanes same address)
                uint4 q;
                asm volatile("ds_read_b128 %0, %1\ns_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(0u) : "memory");
                e0 += q.x;
            } else if (OP == 11) { // v_readfirstlane
                uint32_t s;
                asm volatile("v_readfirstlane_b32 %0, %1\nv_readfirstlane_b32 %0, %2\nv_readfirstlane_b32 %0, %3\nv_readfirstlane_b32 %0, %4\nv_readfirstlane_b32 %0, %1\nv_readfirstlane_b32 %0, %2\nv_readfirstlane_b32 %0, %3\nv_readfirstlane_b32 %0, %4" : "=s"(s) : "v"(e0), "v"(e1), "v"(e2), "v"(e3));
                acc += s;
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    acc += e0 + e1 + e2 + e3 + e4 + e5 + e6 + e7;
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = (uint32_t)(t1 - t0); out[1] = (uint32_t)acc; }
    else if (acc == 0x12345) out[2] = 1;
}
template <int OP> void run(const char* name, uint32_t* d, int waves_per_simd) {
    const int iters = 2000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256 * waves_per_simd), 0, 0, d, 10, 1u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256 * waves_per_simd), 0, 0, d, iters, 1u);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    uint32_t h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * REP * (OP == 10 ? 0.125 : 1.0);
    printf("%-22s waves/SIMD %d: %.2f ns per instr per wave-slot (kernel %.1f us) -> %.2f cycles@2.4GHz per instr per SIMD\n", name, waves_per_simd,
           ms * 1e6 / n, ms * 1e3, ms * 1e6 / n / waves_per_simd * 2.4);
}
int main() {
    uint32_t* d; hipMalloc(&d, 64);
    for (int w = 1; w <= 4; w *= 2) {
        run<0>("v_add_u32", d, w); run<1>("v_mad_i32_i24", d, w); run<2>("v_cmp_lt_u64", d, w); run<3>("v_cmp_lt_u32", d, w);
        run<4>("v_or3_b32", d, w); run<5>("v_fma_f32 clamp", d, w); run<6>("v_cndmask_b32", d, w); run<7>("v_pk_add_f32", d, w);
        run<8>("v_min_u32", d, w); run<9>("v_mul_lo_u32", d, w); run<10>("ds_read_b128 bcast", d, w); run<11>("v_readfirstlane", d, w);
    }
    return 0;
}
