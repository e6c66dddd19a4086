// mirhi_exact.hip.h -- correctly rounded 1/x, a/b and sqrt(x) in a third of the instructions of the compiler's IEEE expansions
// Part of the single device translation unit mirhi_kernels.hip (included inside namespace mirhi); also included by
// tools/microbench/ieee_exact.hip, which checks every function against the compiler's IEEE result on the GPU: 1/x and
// sqrt(x) over EVERY binary32 input of the fast range, a/b over 2^33 random pairs.
//
// Why: the fragment programs must reproduce the oracle's N.H bit for bit (a Blinn-Phong exponent of up to 2048 turns one ulp
// into 1e-4 of colour), and the oracle's normalize / length / divide are IEEE operations.  The compiler expands an IEEE
// binary32 division into ten VALU instructions (v_div_scale x2, v_rcp, four FMAs, v_div_fmas, v_div_fixup) and sqrtf into about
// fourteen, most of them for denormal and special operands that vector lengths never are.  C5 spends 25 such operations
// per shaded pixel: ~275 of its ~750 instructions.
//
// How: the hardware approximation (v_rcp_f32 / v_rsq_f32, 1 ulp) is refined with FMA residual steps (Markstein: a reciprocal
// within one ulp refined by two Newton steps with an FMA-exact residual is correctly rounded; a quotient a * RN(1/b) corrected
// by its FMA-exact remainder is correctly rounded).  Operands outside [2^-96, 2^96] ([2^-60, 2^60] for a / b: denormal results, overflow,
// zero, inf, NaN) take the compiler's expansion: same bits as before, everywhere.
#ifndef MIRHI_EXACT_HIP_H
#define MIRHI_EXACT_HIP_H

#pragma clang fp contract(off)

__device__ __forceinline__ bool exact_fast_range(float x) {           // 2^-96 <= |x| <= 2^96
    const uint32_t e = (__float_as_uint(x) >> 23) & 0xFFu;
    return e - 31u <= 192u;                                           // biased exponent in [31, 223]
}
// RN(1 / x)
__device__ __forceinline__ float rcp_rn(float x) {
    if (!exact_fast_range(x)) return 1.0f / x;
    float y = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, y, 1.0f);
    y = __builtin_fmaf(e, y, y);
    e = __builtin_fmaf(-x, y, 1.0f);
    return __builtin_fmaf(e, y, y);
}
// RN(a / b); fast for 2^-60 <= |a|, |b| <= 2^60 (quotient and remainder stay far from overflow and from the denormals)
__device__ __forceinline__ bool exact_div_range(float x) {
    const uint32_t e = (__float_as_uint(x) >> 23) & 0xFFu;
    return e - 67u <= 120u;                                           // biased exponent in [67, 187]
}
__device__ __forceinline__ float div_rn(float a, float b) {
    if (!exact_div_range(b) || !exact_div_range(a)) return a / b;
    float y = __builtin_amdgcn_rcpf(b);
    float e = __builtin_fmaf(-b, y, 1.0f);
    y = __builtin_fmaf(e, y, y);
    e = __builtin_fmaf(-b, y, 1.0f);
    y = __builtin_fmaf(e, y, y);                      // RN(1 / b)
    const float q = a * y;
    const float r = __builtin_fmaf(-b, q, a);         // exact remainder
    return __builtin_fmaf(r, y, q);
}
// RN(sqrt(x))
__device__ __forceinline__ float sqrt_rn(float x) {
    if (!exact_fast_range(x) || x < 0.0f) return sqrtf(x);
    const float r = __builtin_amdgcn_rsqf(x);
    float s = x * r;
    const float h = 0.5f * r;
    float e = __builtin_fmaf(-s, s, x);
    s = __builtin_fmaf(e, h, s);
    e = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(e, h, s);
}

// Branch-free forms for the fragment programs (register budgets there leave no room for the fallback's control flow): the same
// refinement on every operand, zero / infinity / NaN / negative operands answered by the hardware approximation itself (which
// has the IEEE special values).  Identical to the IEEE result for every binary32 operand except, for 1/x, denormal operands or
// results (|x| below 2^-126 or above 2^126) and, for sqrt, operands below 2^-100 (the residual goes denormal) --
// tools/microbench/ieee_exact.hip checks exactly these ranges over all 2^32 operands; squared lengths, dot products and
// barycentric denominators of a fragment are never there.
__device__ __forceinline__ float rcp_rn_nb(float x) {
    const float y0 = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, y0, 1.0f);
    float y = __builtin_fmaf(e, y0, y0);
    e = __builtin_fmaf(-x, y, 1.0f);
    y = __builtin_fmaf(e, y, y);
    return __builtin_amdgcn_classf(x, 0x267) ? y0 : y;          // x is NaN, +-inf, +-0? (0x267 = NaNs, infinities, zeros): keep the raw answer
}
// (a / b: exact whenever a, b and the quotient are normal numbers away from the ends of the range -- the quantities divided in a
// fragment program are cosines, colours and lengths)
__device__ __forceinline__ float div_rn_nb(float a, float b) {
    const float y0 = __builtin_amdgcn_rcpf(b);
    float e = __builtin_fmaf(-b, y0, 1.0f);
    float y = __builtin_fmaf(e, y0, y0);
    e = __builtin_fmaf(-b, y, 1.0f);
    y = __builtin_fmaf(e, y, y);
    float q = a * y;
    const float r = __builtin_fmaf(-b, q, a);
    q = __builtin_fmaf(r, y, q);
    return (__builtin_amdgcn_classf(b, 0x267) || __builtin_amdgcn_classf(a, 0x267)) ? a * y0 : q;
}
__device__ __forceinline__ float sqrt_rn_nb(float x) {
    const float r = __builtin_amdgcn_rsqf(x);
    float s = x * r;
    const float h = 0.5f * r;
    float e = __builtin_fmaf(-s, s, x);
    s = __builtin_fmaf(e, h, s);
    e = __builtin_fmaf(-s, s, x);
    s = __builtin_fmaf(e, h, s);
    // NaN, +-inf, +-0, any negative: v_sqrt_f32 has the IEEE answer (0x27F = NaNs, infinities, negatives, zeros)
    return __builtin_amdgcn_classf(x, 0x27F) ? __builtin_amdgcn_sqrtf(x) : s;
}

// RN(1 / RN(sqrt(x))): the factor of normalize(); one special-operand test for both steps
__device__ __forceinline__ float inv_sqrt_rn_nb(float x) {
    const float r = __builtin_amdgcn_rsqf(x);
    float s = x * r;
    const float h = 0.5f * r;
    float e = __builtin_fmaf(-s, s, x);
    s = __builtin_fmaf(e, h, s);
    e = __builtin_fmaf(-s, s, x);
    s = __builtin_fmaf(e, h, s);                       // RN(sqrt(x))
    const float y0 = __builtin_amdgcn_rcpf(s);
    e = __builtin_fmaf(-s, y0, 1.0f);
    float y = __builtin_fmaf(e, y0, y0);
    e = __builtin_fmaf(-s, y, 1.0f);
    y = __builtin_fmaf(e, y, y);                       // RN(1 / s)
    return __builtin_amdgcn_classf(x, 0x27F) ? __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(x)) : y;
}

#endif  // MIRHI_EXACT_HIP_H
