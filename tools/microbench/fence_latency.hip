// What a frame loop pays around its kernels on this part (round 3: the reference-shaped loop re-records and fences every frame):
//   1. host cost of a launch with a small / a 2 KB kernarg segment
//   2. launch -> completion round trip seen by the host: a word in pinned memory written by the kernel and polled by the host,
//      against hipEventRecord + hipEventSynchronize
//   3. reading ~30 parameter words per wave from the kernarg segment against reading them from device memory
//   4. "last workgroup signals": returning atomics on eight counters keyed by workgroup id % 8, a root counter, a pinned word
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <sys/wait.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <csignal>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Small { uint32_t* word; uint32_t value; };
struct Big { uint32_t w[480]; uint32_t* word; uint32_t value; uint32_t* sink; };    // ~2 KB

__global__ void k_small(Small a) { if (blockIdx.x == 0 && threadIdx.x == 0 && a.word) __hip_atomic_store(a.word, a.value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
__global__ void k_big(Big a) { if (blockIdx.x == 0 && threadIdx.x == 0 && a.word) __hip_atomic_store(a.word, a.value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

typedef const __attribute__((address_space(4))) uint32_t* CPtr;
// every wave reads 32 words spread over the block (scalar loads), sums them and stores the sum once per workgroup
__global__ __launch_bounds__(256) void k_read_kernarg(Big a) {
    CPtr p = (CPtr)__builtin_amdgcn_kernarg_segment_ptr();
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 32; i++) s += p[(i * 15) % 480];
    if (threadIdx.x == 0) a.sink[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_read_device(const uint32_t* blk, uint32_t* sink) {
    CPtr p = (CPtr)(uintptr_t)blk;
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 32; i++) s += p[(i * 15) % 480];
    if (threadIdx.x == 0) sink[blockIdx.x] = s;
}

// mode 0: nothing; 1: one counter; 2: counters keyed by id % 8 (128 B apart) + root; the last workgroup writes the pinned word
__global__ __launch_bounds__(256) void k_done(uint32_t* frame, uint32_t* counters, uint32_t* word, uint32_t value, int mode, uint32_t total) {
    const uint32_t id = blockIdx.y * gridDim.x + blockIdx.x;
    frame[(size_t)id * 256 + threadIdx.x] = id;           // (the store the signal has to follow)
    if (mode == 0) return;
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x != 0) return;
    if (mode == 1) {
        const uint32_t old = atomicAdd(&counters[0], 1u);
        if (old == total - 1u) { counters[0] = 0; __hip_atomic_store(word, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
        return;
    }
    const uint32_t k = id & 7u;
    const uint32_t mine = (total - k + 7u) / 8u;          // workgroups with id % 8 == k
    const uint32_t old = atomicAdd(&counters[k * 32u], 1u);
    if (old == mine - 1u) {
        counters[k * 32u] = 0;
        const uint32_t r = atomicAdd(&counters[8u * 32u], 1u);
        if (r == 7u) { counters[8u * 32u] = 0; __hip_atomic_store(word, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
    }
}

static void bar_block_test(hipStream_t st, uint32_t* sink, unsigned flag, const char* name) {
    // 7. can the host store into fine-grained device memory directly (parameter block in VRAM written over the BAR)?  In a child process.
    {
        fflush(stdout);
        uint32_t* fg = nullptr;
        hipError_t e = hipExtMallocWithFlags((void**)&fg, 4096, flag);
        printf("hipExtMallocWithFlags(%s): %s\n", name, hipGetErrorString(e));
        if (e == hipSuccess) {
            fflush(stdout);
            // (no fork: a forked child must not touch the GPU; just try it under a SIGSEGV handler)
            struct sigaction sa; memset(&sa, 0, sizeof sa);
            sa.sa_handler = [](int) { const char msg[] = "host store to fine-grained device memory: SIGSEGV (not host-accessible)\n"; (void)!write(1, msg, sizeof msg - 1); _exit(0); };
            sigaction(SIGSEGV, &sa, nullptr); sigaction(SIGBUS, &sa, nullptr);
            volatile uint32_t* v = fg;
            double t0 = now();
            for (int i = 0; i < 1000; i++) { for (int k = 0; k < 256; k++) v[k] = (uint32_t)(i + k); __builtin_ia32_sfence(); }
            printf("host store of 1 KB to fine-grained device memory + sfence: %.3f us\n", 1e6 * (now() - t0) / 1000);
            hipLaunchKernelGGL(k_read_device, dim3(1), dim3(64), 0, st, (const uint32_t*)fg, sink);
            CK(hipStreamSynchronize(st));
            uint32_t got = 0; CK(hipMemcpy(&got, sink, 4, hipMemcpyDeviceToHost));
            uint32_t want = 0; for (int i = 0; i < 32; i++) want += (uint32_t)(999 + (i * 15) % 480 % 256) * ((i * 15) % 480 < 256 ? 1u : 0u);
            printf("kernel read back sum %u (host wrote words 0..255 = 999 + k)\n", got);
            for (int rep = 0; rep < 2; rep++) {
                const int m = 2000;
                CK(hipStreamSynchronize(st));
                double r0 = now();
                for (int i = 0; i < m; i++) hipLaunchKernelGGL(k_read_device, dim3(2040), dim3(256), 0, st, (const uint32_t*)fg, sink);
                CK(hipStreamSynchronize(st));
                printf("32 parameter words per wave from FINE-GRAINED device memory : %.2f us per launch\n", 1e6 * (now() - r0) / m);
            }
            // stale-read check: the host rewrites the block between launches (no HIP call in between), every launch must see its own values
            int stale = 0;
            for (int i = 1; i <= 3000; i++) {
                for (int k = 0; k < 480; k++) v[k] = (uint32_t)i;
                __builtin_ia32_sfence();
                hipLaunchKernelGGL(k_read_device, dim3(64), dim3(256), 0, st, (const uint32_t*)fg, sink + 1024 * (i & 1));
                if ((i & 63) == 0 || i < 8) {            // sampled with a sync; the others stream
                    CK(hipStreamSynchronize(st));
                    uint32_t out[64]; CK(hipMemcpy(out, sink + 1024 * (i & 1), sizeof out, hipMemcpyDeviceToHost));
                    int bad = 0; for (int b = 0; b < 64; b++) bad += out[b] != 32u * (uint32_t)i;
                    if (bad) printf("  synced launch %d: %d stale blocks (saw %u, want %u)\n", i, bad, out[0], 32u * (uint32_t)i);
                    stale += bad;
                }
            }
            // and streaming, two in flight, checked at the end through per-launch sinks
            for (int i = 1; i <= 256; i++) {
                if (i > 2) CK(hipStreamSynchronize(st));
                for (int k = 0; k < 480; k++) v[k] = (uint32_t)(7 * i);
                __builtin_ia32_sfence();
                hipLaunchKernelGGL(k_read_device, dim3(64), dim3(256), 0, st, (const uint32_t*)fg, sink + 4096 + 64 * i);
            }
            CK(hipStreamSynchronize(st));
            { static uint32_t out[64 * 257]; CK(hipMemcpy(out, sink + 4096, sizeof out, hipMemcpyDeviceToHost));
              for (int i = 1; i <= 256; i++) { int bad = 0; for (int b = 0; b < 64; b++) bad += out[64 * i + b] != 32u * 7u * (uint32_t)i;
                if (bad) printf("  streamed launch %d: %d stale blocks (saw %u, want %u)\n", i, bad, out[64 * i], 32u * 7u * (uint32_t)i); stale += bad; } }
            printf("host rewrites the fine-grained block between launches: %d stale reads\n", stale);
        }
    }
}

int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    volatile uint32_t* word; CK(hipHostMalloc((void**)&word, 4096, hipHostMallocMapped | hipHostMallocCoherent));
    uint32_t* word_dev; CK(hipHostGetDevicePointer((void**)&word_dev, (void*)word, 0));
    uint32_t* sink; CK(hipMalloc(&sink, 1 << 20));
    uint32_t* frame; CK(hipMalloc(&frame, 2040 * 256 * 4));
    uint32_t* counters; CK(hipMalloc(&counters, 4096)); CK(hipMemset(counters, 0, 4096));
    uint32_t* blk; CK(hipMalloc(&blk, 2048)); CK(hipMemset(blk, 1, 2048));
    printf("HIP_FORCE_DEV_KERNARG=%s\n", getenv("HIP_FORCE_DEV_KERNARG") ? getenv("HIP_FORCE_DEV_KERNARG") : "(unset)");
    const int n = 20000;
    Big big; memset(&big, 0, sizeof big); big.sink = sink;
    // 1. host cost of a launch
    for (int rep = 0; rep < 2; rep++) {
        CK(hipStreamSynchronize(st));
        double t0 = now();
        for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, st, Small{nullptr, 0});
        double t1 = now(); CK(hipStreamSynchronize(st)); double t2 = now();
        printf("launch, 16 B kernarg : host %.2f us, %.2f us incl. drain\n", 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n);
        t0 = now();
        for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_big, dim3(1), dim3(64), 0, st, big);
        t1 = now(); CK(hipStreamSynchronize(st)); t2 = now();
        printf("launch, 2 KB kernarg : host %.2f us, %.2f us incl. drain\n", 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n);
    }
    // 2. round trips
    {
        const int m = 5000;
        double t0 = now();
        for (int i = 1; i <= m; i++) {
            hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, st, Small{word_dev, (uint32_t)i});
            while (*word != (uint32_t)i) { }
        }
        printf("round trip, polled pinned word      : %.2f us\n", 1e6 * (now() - t0) / m);
        hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        t0 = now();
        for (int i = 1; i <= m; i++) {
            hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, st, Small{nullptr, 0});
            CK(hipEventRecord(ev, st));
            CK(hipEventSynchronize(ev));
        }
        printf("round trip, event record + sync     : %.2f us\n", 1e6 * (now() - t0) / m);
        t0 = now();
        for (int i = 1; i <= m; i++) {
            hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, st, Small{nullptr, 0});
            CK(hipEventRecord(ev, st));
            while (hipEventQuery(ev) == hipErrorNotReady) { }
        }
        printf("round trip, event record + query    : %.2f us\n", 1e6 * (now() - t0) / m);
        t0 = now();
        for (int i = 1; i <= m; i++) {
            hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, st, Small{nullptr, 0});
            CK(hipStreamSynchronize(st));
        }
        printf("round trip, stream synchronize      : %.2f us\n", 1e6 * (now() - t0) / m);
        CK(hipStreamSynchronize(st));
        t0 = now();
        for (int i = 0; i < n; i++) CK(hipEventRecord(ev, st));
        double t1 = now(); CK(hipStreamSynchronize(st));
        printf("hipEventRecord alone : host %.2f us, %.2f us incl. drain\n", 1e6 * (t1 - t0) / n, 1e6 * (now() - t0) / n);
        // two-deep pipeline, the frame loop's shape: wait for launch i - 2, launch i
        t0 = now();
        for (int i = 1; i <= m; i++) {
            if (i > 2) while ((int32_t)(word[(i & 1) * 16] - (uint32_t)(i - 2)) < 0) { }
            hipLaunchKernelGGL(k_done, dim3(60, 34), dim3(256), 0, st, frame, counters, word_dev + (i & 1) * 16, (uint32_t)i, 1 + 1, 2040u);
        }
        CK(hipStreamSynchronize(st));
        printf("two in flight on one stream, 2040-wg kernel with last-wg signal, polled: %.2f us per launch\n", 1e6 * (now() - t0) / m);
    }
    // 3. parameter words from the kernarg segment against device memory (2040 x 256 lanes, 32 scalar loads per wave)
    for (int rep = 0; rep < 2; rep++) {
        const int m = 2000;
        CK(hipStreamSynchronize(st));
        double t0 = now();
        for (int i = 0; i < m; i++) hipLaunchKernelGGL(k_read_kernarg, dim3(2040), dim3(256), 0, st, big);
        CK(hipStreamSynchronize(st));
        printf("32 parameter words per wave from the kernarg segment : %.2f us per launch\n", 1e6 * (now() - t0) / m);
        t0 = now();
        for (int i = 0; i < m; i++) hipLaunchKernelGGL(k_read_device, dim3(2040), dim3(256), 0, st, blk, sink);
        CK(hipStreamSynchronize(st));
        printf("32 parameter words per wave from device memory       : %.2f us per launch\n", 1e6 * (now() - t0) / m);
    }
    // 4. last workgroup signals
    for (int mode = 0; mode < 3; mode++) {
        const int m = 2000;
        CK(hipStreamSynchronize(st));
        double t0 = now();
        for (int i = 1; i <= m; i++) hipLaunchKernelGGL(k_done, dim3(60, 34), dim3(256), 0, st, frame, counters, word_dev + 32, (uint32_t)i, mode, 2040u);
        CK(hipStreamSynchronize(st));
        printf("2040 workgroups store 2 MB, done signal mode %d (0 none, 1 one counter, 2 id %% 8 counters + root): %.2f us per launch%s\n", mode,
               1e6 * (now() - t0) / m, mode && word[32] != (uint32_t)m ? "  SIGNAL MISSING" : "");
    }
    // 5. completion through the dispatch packet's own signal: hipExtLaunchKernelGGL with a stop event, polled with hipEventQuery
    {
        const int m = 5000;
        hipEvent_t ev[4];
        for (auto& evk : ev) CK(hipEventCreate(&evk));
        CK(hipStreamSynchronize(st));
        double t0 = now();
        for (int i = 1; i <= m; i++) {
            hipExtLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, st, nullptr, ev[0], 0, Small{nullptr, 0});
            while (hipEventQuery(ev[0]) == hipErrorNotReady) { }
        }
        printf("round trip, dispatch stop event + query: %.2f us\n", 1e6 * (now() - t0) / m);
        t0 = now();
        for (int i = 0; i < n; i++) hipExtLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, st, nullptr, ev[i & 3], 0, Small{nullptr, 0});
        double t1 = now(); CK(hipStreamSynchronize(st));
        printf("launch with a stop event : host %.2f us, %.2f us incl. drain\n", 1e6 * (t1 - t0) / n, 1e6 * (now() - t0) / n);
        t0 = now();
        for (int i = 1; i <= m; i++) {
            if (i > 2) while (hipEventQuery(ev[i & 1]) == hipErrorNotReady) { }
            hipExtLaunchKernelGGL(k_done, dim3(60, 34), dim3(256), 0, st, nullptr, ev[i & 1], 0, frame, counters, word_dev, 0u, 0, 2040u);
        }
        CK(hipStreamSynchronize(st));
        printf("two in flight on one stream, 2040-wg kernel, dispatch stop event + query: %.2f us per launch\n", 1e6 * (now() - t0) / m);
        double q0 = now();
        for (int i = 0; i < n; i++) (void)hipEventQuery(ev[0]);
        printf("hipEventQuery of a completed event: %.3f us\n", 1e6 * (now() - q0) / n);
    }
    // 6. a small host-to-device copy from pinned memory in the stream (what a parameter upload would cost per frame)
    {
        uint8_t* pin; CK(hipHostMalloc((void**)&pin, 4096, hipHostMallocDefault));
        CK(hipStreamSynchronize(st));
        double t0 = now();
        for (int i = 0; i < n; i++) CK(hipMemcpyAsync(blk, pin, 1024, hipMemcpyHostToDevice, st));
        double t1 = now(); CK(hipStreamSynchronize(st));
        printf("1 KB H2D copy from pinned memory: host %.2f us, %.2f us incl. drain\n", 1e6 * (t1 - t0) / n, 1e6 * (now() - t0) / n);
        t0 = now();
        for (int i = 0; i < n; i++) { CK(hipMemcpyAsync(blk, pin, 1024, hipMemcpyHostToDevice, st)); hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, st, Small{nullptr, 0}); }
        t1 = now(); CK(hipStreamSynchronize(st));
        printf("1 KB H2D copy + launch: host %.2f us, %.2f us incl. drain\n", 1e6 * (t1 - t0) / n, 1e6 * (now() - t0) / n);
    }
    bar_block_test(st, sink, hipDeviceMallocFinegrained, "finegrained");
    bar_block_test(st, sink, hipDeviceMallocUncached, "uncached");
    return 0;
}
