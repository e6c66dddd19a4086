// Would storing textures in 4x4-texel blocks (64 B = half a cache line) cut the lines a bilinear footprint touches (round-2 verdict, item 5)?
// The raster kernel's access shape -- one 256-lane workgroup per 32x32-pixel tile of a 3840x2160 target, a wave = a 16x16 quadrant walked as four
// 8x8 blocks, lane = one pixel of each -- sampling a repeated RGBA8 texture bilinearly at `k` texels per pixel, from a row-major texture and from the
// same texels in 4x4 blocks.  Run under rocprofv3 --kernel-trace --pmc FETCH_SIZE (x2 = bytes, the L2 fetches 128-byte lines) and --stats:
//   texel_layout <texture edge: 1024 | 2048> <k x 100: 100, 200, 400 ...>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int BLOCKED>
__device__ __forceinline__ uint32_t texel(const uint32_t* __restrict__ t, uint32_t x, uint32_t y, uint32_t w) {
    if (BLOCKED) return t[(((y >> 2) * (w >> 2) + (x >> 2)) << 4) + ((y & 3u) << 2) + (x & 3u)];
    return t[y * w + x];
}
template <int BLOCKED>
__global__ __launch_bounds__(256) void sample_kernel(const uint32_t* __restrict__ tex, uint32_t edge, float k, uint32_t* __restrict__ out, uint32_t width, uint32_t height) {
    const uint32_t lane = threadIdx.x & 63u, q = threadIdx.x >> 6;
    const uint32_t qx = blockIdx.x * 32u + (q & 1u) * 16u, qy = blockIdx.y * 32u + (q >> 1) * 16u;
    const float inv = 1.0f / (float)edge;
    uint32_t acc[4];
#pragma unroll
    for (uint32_t b = 0; b < 4; b++) {
        const uint32_t px = qx + (b & 1u) * 8u + (lane & 7u), py = qy + (b >> 1) * 8u + (lane >> 3);
        const float u = ((float)px + 0.5f) * k * inv, v = ((float)py + 0.5f) * k * inv;
        const float fx = u * (float)edge - 0.5f, fy = v * (float)edge - 0.5f;
        const float x0f = floorf(fx), y0f = floorf(fy);
        const uint32_t x0 = (uint32_t)(int32_t)x0f & (edge - 1u), y0 = (uint32_t)(int32_t)y0f & (edge - 1u);
        const uint32_t x1 = (x0 + 1u) & (edge - 1u), y1 = (y0 + 1u) & (edge - 1u);
        const float ax = fx - x0f, ay = fy - y0f;
        const uint32_t c00 = texel<BLOCKED>(tex, x0, y0, edge), c10 = texel<BLOCKED>(tex, x1, y0, edge), c01 = texel<BLOCKED>(tex, x0, y1, edge), c11 = texel<BLOCKED>(tex, x1, y1, edge);
        const float top = (float)(c00 & 255u) + ((float)(c10 & 255u) - (float)(c00 & 255u)) * ax, bot = (float)(c01 & 255u) + ((float)(c11 & 255u) - (float)(c01 & 255u)) * ax;
        acc[b] = (uint32_t)(top + (bot - top) * ay) | (c00 & 0xFF00u);
    }
#pragma unroll
    for (uint32_t b = 0; b < 4; b++) {
        const uint32_t px = qx + (b & 1u) * 8u + (lane & 7u), py = qy + (b >> 1) * 8u + (lane >> 3);
        if (px < width && py < height) __builtin_nontemporal_store(acc[b], out + (size_t)py * width + px);
    }
}

int main(int argc, char** argv) {
    const uint32_t edge = argc > 1 ? (uint32_t)atoi(argv[1]) : 1024u;
    const float k = (argc > 2 ? (float)atoi(argv[2]) : 400.0f) / 100.0f;
    const uint32_t W = 3840, H = 2160;
    std::vector<uint32_t> h((size_t)edge * edge);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u);
    uint32_t *tex, *out;
    CK(hipMalloc(&tex, h.size() * 4)); CK(hipMalloc(&out, (size_t)W * H * 4));
    CK(hipMemcpy(tex, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const dim3 grid(W / 32, (H + 31) / 32);
    for (int layout = 0; layout < 2; layout++) {
        for (int it = 0; it < 12; it++) {
            if (it == 2) CK(hipEventRecord(e0));
            if (layout) hipLaunchKernelGGL(sample_kernel<1>, grid, dim3(256), 0, 0, tex, edge, k, out, W, H);
            else hipLaunchKernelGGL(sample_kernel<0>, grid, dim3(256), 0, 0, tex, edge, k, out, W, H);
        }
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("texture %u^2, %.2f texels per pixel, %s: %.1f us per frame of %u x %u pixels\n", edge, k, layout ? "4x4 blocks" : "row-major ", 1e3f * ms / 10.0f, W, H);
    }
    return 0;
}
