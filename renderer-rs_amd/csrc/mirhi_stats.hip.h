// mirhi_stats.hip.h -- fragment_count_kernel: fragments covered before the depth test (SURVEY 8d "report overdraw separately")
// Part of the single device translation unit mirhi_kernels.hip (included inside namespace mirhi).
#ifndef MIRHI_STATS_HIP_H
#define MIRHI_STATS_HIP_H

// Profiling pass only, never in a timed frame.  One workgroup per tile, one lane per record of the tile's bin (then of the
// big list): the tile record raster_kernel would build (make_tile_rec: same integers, same top-left bias), a walk over the
// record's pixel box inside the tile, one count per covered pixel centre.  The sum over all tiles is the number of fragments
// a GPU's rasterizer would hand to its depth test; divided by the pixels that end up owning a primitive it is the overdraw.
// Runs between geometry_kernel and raster_kernel (the raster kernel re-arms the bin counters).  A triangle that found a bin
// full sits in the big list AND possibly in other bins of its span: such frames (DeviceStats::last_big_list > 0 without
// large triangles) over-count; bench.py reports the big-list length next to the figure.
// (lim_x, lim_y): last pixel of the tile that lies inside the target -- a bin record's box is recomputed from its vertices and may
// reach beyond the target's edge in the last tile column / row (the raster kernel never stores such pixels)
__device__ __forceinline__ uint32_t count_tile_tri(const TileTri& T, int32_t lim_x, int32_t lim_y, int32_t opx = 0, int32_t opy = 0) {
    uint4 rec[4]; uint32_t box = 0;
    if (!make_tile_rec(rec, box, T, opx, opy)) return 0u;
    const int32_t A0 = (int32_t)rec[0].w, A1 = (int32_t)rec[1].x, A2 = (int32_t)rec[1].y;
    const int32_t B0 = (int32_t)rec[1].z, B1 = (int32_t)rec[1].w, B2 = (int32_t)rec[2].x;
    const int32_t bx0 = (int32_t)(box & 0xFF), bx1 = min((int32_t)((box >> 8) & 0xFF), lim_x);
    const int32_t by0 = (int32_t)((box >> 16) & 0xFF), by1 = min((int32_t)(box >> 24), lim_y);
    int32_t r0 = mad24(B0, by0, mad24(A0, bx0, (int32_t)rec[0].x));
    int32_t r1 = mad24(B1, by0, mad24(A1, bx0, (int32_t)rec[0].y));
    int32_t r2 = mad24(B2, by0, mad24(A2, bx0, (int32_t)rec[0].z));
    uint32_t n = 0;
    for (int32_t iy = by0; iy <= by1; iy++) {
        int32_t s0 = r0, s1 = r1, s2 = r2;
        for (int32_t ix = bx0; ix <= bx1; ix++) {
            n += (s0 | s1 | s2) >= 0 ? 1u : 0u;
            s0 += A0; s1 += A1; s2 += A2;
        }
        r0 += B0; r1 += B1; r2 += B2;
    }
    return n;
}

__global__ __launch_bounds__(RASTER_THREADS) void fragment_count_kernel(const PassParams* __restrict__ params, const RasterHead H) {
    ParamsRef P = *(ParamsPtr)(uintptr_t)params;
    __shared__ uint32_t lds_sum;
    const uint32_t tx = blockIdx.x, tyr = blockIdx.y, tid = threadIdx.x;
    const uint32_t tile = tyr * H.tiles_x + tx, ty = H.tile_row_begin + tyr * H.tile_row_step;
    if (tid == 0) lds_sum = 0;
    __syncthreads();
    uint32_t n = 0;
    const int32_t lim_x = (int32_t)P.width - 1 - (int32_t)(tx * TILE), lim_y = (int32_t)P.height - 1 - (int32_t)(ty * TILE);
    const uint4* pool = reinterpret_cast<const uint4*>(H.bin_pool);
    const uint32_t nsub = H.count_stride ? 8u : 1u;
    for (uint32_t k = 0; k < nsub; k++) {
        const uint32_t raw = H.bin_count[(k * H.count_stride + tile) * BIN_COUNT_STRIDE];
        const uint32_t cnt = raw < H.sub_cap ? raw : H.sub_cap;
        for (uint32_t j = tid; j < cnt; j += RASTER_THREADS) {
            const uint32_t page = j < H.fixed_recs ? (tile * H.fixed_recs + j) >> BIN_PAGE_LOG2 : P.bin_table[tile * (uint32_t)BIN_TABLE_ROW + k * 8u + (j >> BIN_PAGE_LOG2)];
            if (page >= PAGE_NONE) continue;
            const size_t ri = ((size_t)page * BIN_PAGE_RECS + (j & (BIN_PAGE_RECS - 1u))) * 2u;
            TileTri T;
            tile_tri_from_bin(T, pool[ri], pool[ri + 1u]);
            n += count_tile_tri(T, lim_x, lim_y);
        }
    }
    const uint32_t nbig_raw = *H.big_count;
    const uint32_t nbig = nbig_raw < H.big_cap ? nbig_raw : H.big_cap;
    const uint4* big = reinterpret_cast<const uint4*>(P.big_recs);
    for (uint32_t i = tid; i < nbig; i += RASTER_THREADS) {
        TileTri T;
        if (tile_tri_from_big(T, big[(size_t)i * 3u], big[(size_t)i * 3u + 1u], big[(size_t)i * 3u + 2u], (int32_t)tx, (int32_t)ty)) n += count_tile_tri(T, lim_x, lim_y, (int32_t)tx * TILE, (int32_t)ty * TILE);
    }
    if (n) atomicAdd(&lds_sum, n);
    __syncthreads();
    if (tid == 0 && lds_sum) atomicAdd(&P.frag_stats[1], (unsigned long long)lds_sum);
}

// Pixels that hold a primitive in the scope's primitive-id image = pixels whose fragment program ran (this design shades the
// winner of the depth resolve only).  The statistics pass renders with a primitive-id image of its own, cleared to NO_PRIM.
__global__ __launch_bounds__(RASTER_THREADS) void winner_count_kernel(const uint32_t* __restrict__ prim, uint32_t pixels, unsigned long long* stats) {
    __shared__ uint32_t lds_sum;
    if (threadIdx.x == 0) lds_sum = 0;
    __syncthreads();
    uint32_t n = 0;
    const uint32_t base = blockIdx.x * RASTER_THREADS * 16u;
#pragma unroll
    for (uint32_t k = 0; k < 16u; k++) {
        const uint32_t i = base + k * RASTER_THREADS + threadIdx.x;
        if (i < pixels) n += prim[i] != NO_PRIM ? 1u : 0u;
    }
    const uint32_t w = (uint32_t)__popcll(__ballot(n & 1u)) + 2u * (uint32_t)__popcll(__ballot(n & 2u)) + 4u * (uint32_t)__popcll(__ballot(n & 4u)) +
                       8u * (uint32_t)__popcll(__ballot(n & 8u)) + 16u * (uint32_t)__popcll(__ballot(n & 16u));
    if ((threadIdx.x & 63u) == 0 && w) atomicAdd(&lds_sum, w);
    __syncthreads();
    if (threadIdx.x == 0 && lds_sum) atomicAdd(&stats[0], (unsigned long long)lds_sum);
}

#endif  // MIRHI_STATS_HIP_H
