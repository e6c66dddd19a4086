// Can a workgroup on one XCD read, inside the SAME launch, what a workgroup on another XCD wrote -- and with which kind of store / load?
// (A one-launch frame -- geometry workgroups first in the grid, raster workgroups waiting on a done-counter -- needs exactly that: the bin
// records written by geometry waves are read by raster workgroups that may sit on any other XCD, each XCD has its own L2, and the buffers
// are reused frame after frame, so every L2 may hold last frame's lines.)
// Producers: the first NP workgroups write `value` into their slice of X, wait for the stores, bump a counter.  Consumers: all other
// workgroups first touch X (so their L2 holds lines -- of THIS launch before the producers wrote, i.e. stale ones, when the poll
// starts early enough, and of the previous launch in any case), poll the counter (bounded), then read X and count words != value.
// Modes of the producer's stores: 0 plain, 1 sc1 (agent scope, write-through), 2 plain + buffer_wbl2 sc1
// Modes of the consumer's loads : 0 plain, 1 sc1, 2 buffer_inv sc1 then plain
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_sc1(uint4* p, uint4 v) { const u32x4 w = {v.x, v.y, v.z, v.w}; asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(w) : "memory"); }
__device__ __forceinline__ uint4 load_sc1(const uint4* p) { u32x4 v; asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ uint4 load_plain(const uint4* p) { u32x4 v; asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return make_uint4(v.x, v.y, v.z, v.w); }

constexpr uint32_t NP = 160;            // producer workgroups (one wave each does the work)
constexpr uint32_t SLICE = 256;         // uint4 per producer = 4 KB

__global__ __launch_bounds__(256) void k(uint4* X, uint32_t* done, uint32_t* bad, uint32_t* timeouts, uint32_t value, int smode, int lmode, uint32_t* xcd_of) {
    const uint32_t wg = blockIdx.x;
    const uint32_t xcd = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u;
    if (threadIdx.x == 0) xcd_of[wg] = xcd;
    if (wg < NP) {
        if (threadIdx.x >= 64) return;
        // a little delay so consumers get to pre-touch X and start polling first
        for (int i = 0; i < 2000; i++) __builtin_amdgcn_s_sleep(8);
        for (uint32_t i = threadIdx.x; i < SLICE; i += 64) {
            const uint4 v = make_uint4(value, value + i, value ^ wg, value);
            if (smode == 1) store_sc1(&X[wg * SLICE + i], v); else X[wg * SLICE + i] = v;
        }
        if (smode == 2) asm volatile("buffer_wbl2 sc1" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0);
        if (threadIdx.x == 0) __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    // consumer: pre-touch the slice it will check (pulls the current -- old -- lines into this XCD's L2)
    const uint32_t p = (wg * 7u) % NP;
    uint32_t sink = 0;
    for (uint32_t i = threadIdx.x; i < SLICE; i += 256) sink += load_plain(&X[p * SLICE + i]).x;
    __shared__ uint32_t ok;
    if (threadIdx.x == 0) {
        uint32_t spins = 0, d = 0;
        for (; spins < (1u << 20); spins++) {
            d = __hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (d >= NP) break;
            __builtin_amdgcn_s_sleep(4);
        }
        ok = d >= NP;
        if (d < NP) atomicAdd(timeouts, 1u);
    }
    __syncthreads();
    if (!ok) return;
    if (lmode == 2) asm volatile("buffer_inv sc1" ::: "memory");
    uint32_t nbad = 0;
    for (uint32_t i = threadIdx.x; i < SLICE; i += 256) {
        const uint4 v = lmode == 1 ? load_sc1(&X[p * SLICE + i]) : load_plain(&X[p * SLICE + i]);
        nbad += (v.x != value) + (v.y != value + i) + (v.z != (value ^ p)) + (v.w != value);
    }
    if (nbad) atomicAdd(bad, nbad);
    if (sink == 0xFFFFFFFFu) bad[1] = sink;
}

// cost of a gather through sc1 loads against plain ones: 2040 x 256 lanes, each lane 4 loads of a 4-byte word from a 40 KB table
__global__ __launch_bounds__(256) void gather(const uint32_t* tab, uint32_t* out, int sc1) {
    const uint32_t id = blockIdx.x * 256 + threadIdx.x;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t idx = ((id >> 3) * 2654435761u + k * 977u) % 10000u;      // ~8 lanes share an entry
        uint32_t v;
        if (sc1) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(tab + idx) : "memory");
        else asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(tab + idx) : "memory");
        s += v;
    }
    out[id] = s;
}

int main() {
    uint4* X; CK(hipMalloc(&X, NP * SLICE * sizeof(uint4)));
    uint32_t *done, *bad, *timeouts, *xcd_of; CK(hipMalloc(&done, 256)); CK(hipMalloc(&bad, 256)); CK(hipMalloc(&timeouts, 256)); CK(hipMalloc(&xcd_of, 4096 * 4));
    CK(hipMemset(X, 0, NP * SLICE * sizeof(uint4)));
    const uint32_t total = NP + 1880;
    for (int smode = 0; smode < 3; smode++)
        for (int lmode = 0; lmode < 3; lmode++) {
            uint32_t h_bad = 0, h_to = 0; double t = 0;
            CK(hipMemset(bad, 0, 256)); CK(hipMemset(timeouts, 0, 256));
            for (uint32_t it = 1; it <= 200; it++) {
                CK(hipMemset(done, 0, 4));
                CK(hipDeviceSynchronize());
                const double t0 = now();
                hipLaunchKernelGGL(k, dim3(total), dim3(256), 0, 0, X, done, bad, timeouts, it * 0x01010101u + (uint32_t)(smode * 3 + lmode), smode, lmode, xcd_of);
                CK(hipDeviceSynchronize());
                t += now() - t0;
            }
            CK(hipMemcpy(&h_bad, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&h_to, timeouts, 4, hipMemcpyDeviceToHost));
            static const char* sn[] = {"plain stores", "sc1 stores", "plain stores + buffer_wbl2 sc1"}; static const char* ln[] = {"plain loads", "sc1 loads", "buffer_inv sc1 + plain loads"};
            printf("%-32s / %-30s: %8u stale words in 200 launches, %u poll timeouts, %.1f us per launch\n", sn[smode], ln[lmode], h_bad, h_to, 1e6 * t / 200);
        }
    static uint32_t hx[4096]; CK(hipMemcpy(hx, xcd_of, total * 4, hipMemcpyDeviceToHost));
    uint32_t mism = 0; for (uint32_t i = 0; i < total; i++) mism += hx[i] != (i & 7u);
    printf("workgroups whose XCC_ID != id %% 8: %u of %u\n", mism, total);
    uint32_t *tab, *out; CK(hipMalloc(&tab, 40000)); CK(hipMalloc(&out, 2040 * 256 * 4)); CK(hipMemset(tab, 1, 40000));
    for (int sc1 = 0; sc1 < 2; sc1++) {
        CK(hipDeviceSynchronize());
        const double t0 = now();
        for (int i = 0; i < 500; i++) hipLaunchKernelGGL(gather, dim3(2040), dim3(256), 0, 0, tab, out, sc1);
        CK(hipDeviceSynchronize());
        printf("gather of a 40 KB table, %s: %.2f us per launch\n", sc1 ? "sc1 loads" : "plain loads", 1e6 * (now() - t0) / 500);
    }
    return 0;
}
