// store_layout.hip -- does the LAYOUT of the frame store matter?  2040 workgroups of 256 lanes write a 1920x1080 BGRA8 frame (8.3 MB)
//   blocks  as raster_kernel does: wave q owns a 16x16 quadrant of the 32x32 tile, a lane one pixel in each of its four 8x8 blocks:
//           every store instruction touches 8 rows x 32 B
//   rows    16 B per lane, 8 lanes per 128-byte line: every store instruction writes whole lines of the tile (8 rows x 128 B per wave)
//   linear  4 B per lane, consecutive lanes consecutive addresses (256 B per wave instruction), ignoring tiles
// Back-to-back on one stream and as the gap-free time of 4 streams.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
constexpr uint32_t W = 1920, H = 1080, TX = 60, TY = 34;
__global__ __launch_bounds__(256) void k_blocks(uint32_t* out, uint32_t v) {
    const uint32_t tx = blockIdx.x, ty = blockIdx.y, lane = threadIdx.x & 63u, q = threadIdx.x >> 6;
    const uint32_t ix0 = (q & 1u) * 16u + (lane & 7u), iy0 = (q >> 1) * 16u + (lane >> 3);
#pragma unroll
    for (uint32_t b = 0; b < 4; b++) {
        const uint32_t px = tx * 32u + ix0 + (b & 1u) * 8u, py = ty * 32u + iy0 + (b >> 1) * 8u;
        if (py < H) out[py * W + px] = v + b;
    }
}
__global__ __launch_bounds__(256) void k_rows(uint32_t* out, uint32_t v) {
    const uint32_t tx = blockIdx.x, ty = blockIdx.y, t = threadIdx.x;
    const uint32_t px = tx * 32u + (t & 7u) * 4u, py = ty * 32u + (t >> 3);
    if (py < H) *reinterpret_cast<uint4*>(out + py * W + px) = make_uint4(v, v + 1, v + 2, v + 3);
}
__global__ __launch_bounds__(256) void k_linear(uint32_t* out, uint32_t v) {
    const uint32_t i = (blockIdx.y * TX + blockIdx.x) * 1024u + threadIdx.x;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) if (i + k * 256u < W * H) out[i + k * 256u] = v + k;
}
template <class F> double timeit(F f, hipStream_t* s, int ns, int n = 2000) {
    for (int i = 0; i < 50; i++) f(s[i % ns]);
    for (int i = 0; i < ns; i++) hipStreamSynchronize(s[i]);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; i++) f(s[i % ns]);
    for (int i = 0; i < ns; i++) hipStreamSynchronize(s[i]);
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
}
int main() {
    hipStream_t s[4];
    for (auto& x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    uint32_t* out[4];
    for (auto& p : out) hipMalloc(&p, W * H * 4 + 4096);
    const dim3 grid(TX, TY), block(256);
    printf("%-10s %12s %12s\n", "layout", "1 stream us", "4 streams us");
#define RUN(K, label) { int c = 0; double a = timeit([&](hipStream_t st) { hipLaunchKernelGGL(K, grid, block, 0, st, out[0], 7u); }, s, 1); \
                         double b = timeit([&](hipStream_t st) { hipLaunchKernelGGL(K, grid, block, 0, st, out[c++ & 3], 7u); }, s, 4); printf("%-10s %12.2f %12.2f\n", label, a, b); }
    RUN(k_blocks, "blocks"); RUN(k_rows, "rows"); RUN(k_linear, "linear");
    return 0;
}
