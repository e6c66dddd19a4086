// mirhi_native.h -- dispatching the rasterizer's kernels by writing the AQL packets ourselves (ROCr user-mode queues), next to the HIP runtime.
//
// Why: a HIP launch costs the host 2.3 - 3.0 us whatever the entry point (tools/microbench/launch_paths.hip) -- 5 - 7 us per frame of two
// or three kernels, in the latency chain of every frame of the reference-shaped loop (wait fence -> record -> submit, renderer.rs:367-449)
// -- while the packet itself is 64 bytes, the kernel arguments ~100 bytes and the doorbell one store: 0.2 us (tools/microbench/hsa/).
// A queue lane gets an AQL queue of its own; kernel arguments go into a ring of fine-grained device memory the host writes directly; a
// submit's fence is the completion signal of its last packet, polled in host memory.  The kernels are the SAME code object the HIP runtime
// loads (extracted from the built object at build time: libmirhi_kernels.hsaco); HIP keeps everything else -- allocation, copies, the
// timed / batched / ordered / split paths, which still launch through it.  If ROCr or the code object is not there, every launch goes
// through HIP as before.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace mirhi {
struct NativeQueue;       // one AQL queue (a queue lane of a device)
// one kernel dispatch: `key` the host address of the __global__ function (its device-side symbol is asked of the HIP runtime once:
// hipKernelNameRefByPtr), `args` the explicit arguments packed as the kernel ABI lays them out; `signal` (an hsa_signal_t handle, 0 = none)
// is decremented when the kernel has finished
// flags: memory scope of the packet's acquire / release fences -- agent scope unless asked for system scope (the first kernel of a submit
// must see what the host wrote, the last one must publish the frame to the host and the copy engines; the ones in between only talk
// to each other: a system-scope fence on every packet cost the frame loop 4 - 7 %)
enum : uint32_t { NATIVE_ACQUIRE_SYSTEM = 1u, NATIVE_RELEASE_SYSTEM = 2u };
hipError_t native_enqueue(NativeQueue* q, const void* key, dim3 grid, dim3 block, const void* args, size_t args_bytes, uint64_t signal, uint32_t flags);

// packs kernel arguments the way the compiler lays out the explicit kernarg segment: each at its natural alignment
struct KernargPacker {
    alignas(16) uint8_t bytes[512];
    size_t size = 0;
    template <typename T> void push(const T& v) {
        size = (size + alignof(T) - 1) & ~(alignof(T) - 1);
        __builtin_memcpy(bytes + size, &v, sizeof(T));
        size += sizeof(T);
    }
};
template <typename... A>
inline hipError_t native_launch(NativeQueue* q, const void* key, dim3 grid, dim3 block, uint64_t signal, uint32_t flags, const A&... a) {
    KernargPacker p;
    (p.push(a), ...);
    return native_enqueue(q, key, grid, block, p.bytes, p.size, signal, flags);
}
}  // namespace mirhi
