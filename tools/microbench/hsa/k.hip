// device code of the HSA direct-dispatch prototype (compiled to a code object: hipcc --genco)
#include <hip/hip_runtime.h>
#include <stdint.h>
struct Args { uint32_t* out; uint32_t value; uint32_t pad; uint32_t* word; };
extern "C" __global__ __launch_bounds__(256) void probe_kernel(Args a) {
    const uint32_t id = blockIdx.y * gridDim.x + blockIdx.x;
    if (threadIdx.x == 0) a.out[id] = a.value + id + (gridDim.x << 16) + (gridDim.y << 24);
    if (id == 0 && threadIdx.x == 0 && a.word) __hip_atomic_store(a.word, a.value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
