// mirhi_common.hip.h -- diagnostic stamps, small vector types, constant-address-space access to descriptors and parameters
// Part of the single device translation unit mirhi_kernels.hip (included inside namespace mirhi).
#ifndef MIRHI_COMMON_HIP_H
#define MIRHI_COMMON_HIP_H

#ifdef MIRHI_STAMPS
// Diagnostic build only (build.py --stamps -> libmirhi_stamps.so): per-wave s_memtime stamps at phase
// boundaries, written to a buffer nothing else reads.  Never compiled into libmirhi.so.
__device__ uint64_t g_stamps[32768 * 8];
__device__ uint64_t g_stamps_geo[16384 * 8];
__device__ uint64_t g_stamps_geo_rt[16384 * 2];          // start and end of each geometry wave on the device-wide 100 MHz clock
#define GSTAMP(k) do { if ((threadIdx.x & 63u) == 0 && blockIdx.x < 16384u) { g_stamps_geo[blockIdx.x * 8u + (k)] = __builtin_amdgcn_s_memtime(); if ((k) == 0) g_stamps_geo_rt[blockIdx.x * 2u] = wall_clock64(); if ((k) == 3) g_stamps_geo_rt[blockIdx.x * 2u + 1u] = wall_clock64(); } } while (0)
// diagnostic only: wait for everything outstanding, then stamp (where did the time go: this changes the schedule it measures)
#define GSTAMP_SYNC(k) do { __builtin_amdgcn_s_waitcnt(0); GSTAMP(k); } while (0)
#define STAMP(k) do { if ((threadIdx.x & 63u) == 0) { const uint32_t wv = ((blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)); \
    if (wv < 32768u) { g_stamps[wv * 8u + (k)] = __builtin_amdgcn_s_memtime(); if ((k) == 0) g_stamps[wv * 8u + 6u] = wall_clock64(); if ((k) == 4) g_stamps[wv * 8u + 7u] = wall_clock64(); } } } while (0)   /* slots 6, 7: start and end on the device-wide 100 MHz clock (s_memtime counts per compute unit group: no two waves can be compared) */
// instruction-count attribution: the raster kernel returns after stage g_stage_limit (1 prologue, 2 fill of the first
// chunk, 3 both lists); the SQ instruction counters of such runs, differenced, give the dynamic cost of each stage
__device__ uint32_t g_stage_limit;
#define STAGE_END(k) do { if (g_stage_limit == (k)) return; } while (0)
#else
#define STAMP(k) do {} while (0)
#define GSTAMP(k) do {} while (0)
#define GSTAMP_SYNC(k) do {} while (0)
#define STAGE_END(k) do {} while (0)
#endif

struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

__device__ __forceinline__ float ldf(const uint8_t* p, uint32_t off) { return *reinterpret_cast<const float*>(p + off); }
__device__ __forceinline__ uint32_t ldu(const uint8_t* p, uint32_t off) { return *reinterpret_cast<const uint32_t*>(p + off); }

// Draw descriptors and uniform blocks are read-only for the whole launch.  Reading them through the constant
// address space lets the compiler use scalar loads (SGPRs, scalar cache) whenever the address is wave-uniform;
// through a plain pointer it must assume the kernel's own stores may alias and falls back to vector loads.
#define MIRHI_CONST __attribute__((address_space(4)))
typedef const MIRHI_CONST DrawDesc* DrawPtr;
typedef const MIRHI_CONST float* CFloatPtr;
typedef const MIRHI_CONST uint8_t* CBytePtr;
__device__ __forceinline__ DrawPtr const_draws(const DrawDesc* p) { return (DrawPtr)(uintptr_t)p; }
__device__ __forceinline__ CFloatPtr cf(const float* p) { return (CFloatPtr)(uintptr_t)p; }
__device__ __forceinline__ CBytePtr cb(const uint8_t* p) { return (CBytePtr)(uintptr_t)p; }
__device__ __forceinline__ float ldcf(CBytePtr p, uint32_t off) { return *reinterpret_cast<const MIRHI_CONST float*>(p + off); }
__device__ __forceinline__ uint32_t ldcu(CBytePtr p, uint32_t off) { return *reinterpret_cast<const MIRHI_CONST uint32_t*>(p + off); }
typedef const MIRHI_CONST DrawDesc& DrawRef;
typedef const MIRHI_CONST PassParams* ParamsPtr;
typedef const MIRHI_CONST PassParams& ParamsRef;   // scalar (s_load) access to the pass parameters in device memory
// opaque to the optimiser: loads through the result cannot be hoisted above this point
__device__ __forceinline__ ParamsPtr launder_params(ParamsPtr p) { asm volatile("" : "+s"(p)); return p; }

// HLSL mul(M, v), M column-major (vertex/model.hlsl:44,48); accumulation order = oracle's.
__device__ __forceinline__ f4 mat4_mul(CFloatPtr m, f4 v) {
    f4 r;
    r.x = ((m[0] * v.x + m[4] * v.y) + m[8] * v.z) + m[12] * v.w;
    r.y = ((m[1] * v.x + m[5] * v.y) + m[9] * v.z) + m[13] * v.w;
    r.z = ((m[2] * v.x + m[6] * v.y) + m[10] * v.z) + m[14] * v.w;
    r.w = ((m[3] * v.x + m[7] * v.y) + m[11] * v.z) + m[15] * v.w;
    return r;
}
__device__ __forceinline__ f3 mat3_mul(CFloatPtr m, f3 v) {
    f3 r;
    r.x = (m[0] * v.x + m[4] * v.y) + m[8] * v.z;
    r.y = (m[1] * v.x + m[5] * v.y) + m[9] * v.z;
    r.z = (m[2] * v.x + m[6] * v.y) + m[10] * v.z;
    return r;
}
__device__ __forceinline__ float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ f3 add3(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 sub3(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 scale3(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ f3 mul3(f3 a, f3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ f3 cross3(f3 a, f3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// IEEE sqrt and division, bit for bit the oracle's sqrtf / "/" (mirhi_exact.hip.h: same results as the compiler's expansions at a
// third of their instructions)
__device__ __forceinline__ float length3(f3 a) { return sqrt_rn_nb(dot3(a, a)); }
__device__ __forceinline__ f3 normalize3(f3 a) { const float inv = inv_sqrt_rn_nb(dot3(a, a)); return {a.x * inv, a.y * inv, a.z * inv}; }
__device__ __forceinline__ float saturatef(float x) { return x > 0.0f ? (x < 1.0f ? x : 1.0f) : 0.0f; }

#endif  // MIRHI_COMMON_HIP_H
