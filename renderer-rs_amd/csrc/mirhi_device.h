// mirhi_device.h -- structures shared by the host C-ABI layer and the gfx950 kernels.
//
// Data layout in HBM (see DESIGN.md "Data layout"):
//   DrawDesc[]   one per recorded draw of a rendering scope (pointers into caller buffers + state)
//   BinRec[]     the bin pool: pages of 64 tile-relative triangle records of 32 B.  A tile's bin is a list of pages -- the
//                first one at a fixed place (page = tile), further ones taken from the pool by the geometry kernel and
//                entered in the tile's row of the page table -- written by the geometry kernel, consumed by exactly one
//                raster workgroup
//   BigRec[]     screen-space triangles (48 B, absolute coordinates) too large for the bins, clipped, cut by a scissor, or
//                spilled from a full bin / an exhausted pool; every raster workgroup scans this list
//   counters     bin_count[tiles], pool_next, big_count, status words
#pragma once
#include <stdint.h>

namespace mirhi {

constexpr int TILE = 32;              // screen tile edge in pixels (one raster workgroup)
constexpr int TILE_LOG2 = 5;
constexpr int BLOCK = 8;              // coverage block edge (one wave-iteration covers 8x8 px)
constexpr int RASTER_THREADS = 256;   // 4 waves; wave q owns the 16x16 quadrant q, 4 px per lane
constexpr int RASTER_CHUNK = 192;     // triangle records staged per LDS pass (12 KB)
constexpr int TP_MIN_LANES = 32;      // small records a wave must hold before it resolves them triangle-parallel
constexpr int GEOM_THREADS = 64;
constexpr int MAX_BIN_SPAN = 4;       // triangles spanning more than 4x4 tiles go to the big list
constexpr float GUARD_PX = 16000.0f;  // guard band: snapped coordinates stay inside +-2^22 sub-pixels
constexpr int BIN_PAGE_RECS = 64;     // records per bin page (2 KB)
constexpr int BIN_PAGE_LOG2 = 6;
constexpr int BIN_TABLE_ROW = 64;     // page-table entries per tile: 64 pages of one list, or 8 pages for each of 8 per-XCD lists
constexpr int POOL_COUNTER_STRIDE = 32;        // words between the per-XCD pool counters (128 B)
// Words between two bin counters (64 B): one counter per cache line sector.  Packed 32 to a 128-byte line (rounds 1-3) the counters of a whole row segment of
// tiles queued their atomics on ONE line -- a scattered scope (C2: 50k reservations on 2,040 counters, every wave all over the frame) puts ~800 of them on each
// line, one after the other: tools/microbench/atomic_pad.hip, the reservation pattern of a C2 geometry wave: 5.7 us from its first atomic to its last result
// with packed counters, 1.7 us with one counter per 64 bytes (128 bytes: the same), kernel 8.2 -> 4.0 us.  Counter c lives at bin_count[c * BIN_COUNT_STRIDE].
constexpr uint32_t BIN_COUNT_STRIDE = 16;
constexpr uint32_t PAGE_EMPTY = 0xFFFFFFFFu;   // table entry not (yet) published
constexpr uint32_t PAGE_NONE = 0xFFFFFFFEu;    // the pool was exhausted when this page was asked for: its records go to the big list
constexpr uint32_t NO_PRIM = 0xFFFFFFFFu;
constexpr uint32_t MAX_PRIM_ID = 0xFFFFFFFDu;

enum : uint32_t { STATUS_BIG_OVERFLOW = 1u, STATUS_ALPHA_TEST_TEXTURED = 2u,
                  STATUS_POOL_EXHAUSTED = 4u,      // not an error: records went to the big list instead; the host grows the pool
                  STATUS_PAGE_TIMEOUT = 8u };      // a page was never published (cannot happen by construction; bounded wait)
constexpr uint32_t ORDERED_MARKER = 0xFFFFFFFFu;   // bbox-y word of an ordered slot whose triangle was clipped: pieces live in the big list
constexpr int ORDERED_THREADS = 256;

struct DrawDesc {
    const uint8_t* vb;            // binding 0 base + bind offset
    const uint8_t* ib;            // index base + bind offset (nullptr for draw)
    const float*   camera;        // b0 CameraData
    const float*   object;        // b1 ObjectData
    const uint8_t* lights;        // b2 LightUBO
    const uint8_t* material;      // b3 MaterialData
    const uint8_t* point_lights;  // t0 space1
    const uint8_t* spot_lights;   // t1 space1
    const uint8_t* tex[5];        // t0 albedo, t1 normal, t2 metallic-roughness, t3 occlusion, t4 emissive (RGBA8)
    uint32_t tex_w[5], tex_h[5];
    uint32_t tex_levels[5];       // mip levels stored behind level 0 (1 = no chain: bilinear)
    uint32_t tex_srgb;            // bit t: texture t is R8G8B8A8_SRGB (RGB decoded to linear when sampled)
    uint32_t tex_any_mips;        // some bound texture has a chain: the fragment program evaluates UV derivatives
    uint32_t stride;
    uint32_t index_type;          // 0 none, 2 u16, 4 u32
    uint32_t first;               // first_vertex / first_index
    int32_t  vertex_offset;
    uint32_t tri_count;
    uint32_t prim_base;           // global primitive id of triangle 0
    uint32_t slot_base;           // first geometry-kernel lane of this draw (draws are padded to whole waves)
    uint32_t program;
    uint32_t cull_mode, front_face;
    float    hw, hh, cx, cy;      // viewport half extents and centre
    float    dscale, dmin;        // depth range
    float    gx, gy;              // guard-band plane factors
    int32_t  sx0, sy0, sx1, sy1;  // inclusive scissor (already clamped to render area and extent)
    uint32_t scissor_partial;     // scissor smaller than the target: per-pixel box test needed when it cuts a bbox
    uint32_t vs_words;            // 16-byte words per shaded vertex (3 MODEL, 5 MODEL_FULL, 0 = no vertex pre-pass)
    uint32_t tex_aniso;           // bits 4t..4t+3: max_anisotropy - 1 of texture t (0 = trilinear; set only for textures with a chain)
    const void* vs_attr;          // shaded vertices, the other words (vs_words - 1 per vertex): see VsJob
    const void* vs_out;           // shaded vertices, clip positions (16 B per vertex), indexed like the vertex buffer: see VsJob
};
static_assert(sizeof(DrawDesc) % 16 == 0, "DrawDesc must stay 16-byte sized");

// Vertex-shader pre-pass (vertex/model.hlsl:39-68 run once per vertex, as a GPU's vertex stage does, instead of three
// times per shaded pixel): one job per distinct (vertex buffer range, camera, object, program class) of a scope.
// Shaded vertex = vs_words 16-byte words in TWO streams: the clip position on its own (what the geometry kernel gathers for every
// triangle of every rank: 16 B per vertex, not a 48 / 80-byte row -- C4: 8 MB instead of 24 MB per frame and rank), the rest behind it
//   clip[v]  = clip position
//   attr[v]  = { world.xyz, N.x }, { N.y, N.z, u, v }                     (vs_words - 1 = 2 words, MODEL)
//              + { T.xyz, B.x }, { B.y, B.z, 0, 0 }                       (vs_words - 1 = 4 words, MODEL_FULL / MODEL_PBR)
// A job's output is [count x 16 B clip, rounded up to 256 B][count x (words - 1) x 16 B]: vs_attr_of() below.
struct VsJob {
    const uint8_t* vb;
    const float*   camera;
    const float*   object;
    void*          out;
    uint32_t stride, count, slot_base, words;
};
static_assert(sizeof(VsJob) == 48, "VsJob is 48 bytes");
// the attribute stream of a job's output block `out` for `count` vertices (host and device agree through this one function)
#if defined(__HIPCC__) || defined(__CUDACC__)
__host__ __device__
#endif
inline size_t vs_attr_offset(uint32_t count) { return ((size_t)count * 16u + 255u) & ~(size_t)255u; }

// Screen-space triangle record, 12 dwords = three 16-byte words.  Written once per overlapped tile into
// that tile's bin (and once into the big list for triangles that span more than 4x4 tiles, were clipped,
// or found a bin full); the raster kernel turns it into a tile-relative record in LDS.
//   w0 = { X0, Y0, X1, Y1 }    snapped vertices, 1/256 px, orientation normalised (interior has E > 0)
//   w1 = { X2, Y2, z0, zx }    depth of vertex 0 and depth plane d/dx
//   w2 = { zy, idk, bx, by }   d/dy, primitive id in tie-break order,
//                              bx = minx | maxx << 16 | boxed << 31, by = miny | maxy << 16 (inclusive pixel box)
struct TriRec { uint32_t w[12]; };
static_assert(sizeof(TriRec) == 48, "TriRec is 48 bytes");
typedef TriRec BigRec;    // the big list holds TriRecs

// Bin record, 8 dwords = two 16-byte words: the same triangle relative to the tile whose bin it sits in.  Only triangles whose
// pixel box spans at most 4 x 4 tiles and whose vertices lie within 65535 sub-pixels of each other are binned, so every
// coordinate is a 16-bit number; the pixel box is recomputed from the vertices (a box cut by a scissor goes to the big list).
//   w0 = { ox | oy << 16,      smallest vertex x / y minus the tile origin, 1/256 px, signed 16 bit
//          x0 | y0 << 16, x1 | y1 << 16, x2 | y2 << 16 }   vertices minus (smallest x, smallest y), unsigned 16 bit
//   w1 = { z0, zx, zy, idk }   as in TriRec
struct BinRec { uint32_t w[8]; };
static_assert(sizeof(BinRec) == 32, "BinRec is 32 bytes");

struct PassParams {
    uint32_t width, height;           // colour target extent
    uint32_t tiles_x, tiles_y;
    // Tile rows rasterized on this device (tile-row split): rows tile_row_begin + k * tile_row_step, k = 0 .. tile_row_end - tile_row_begin - 1.
    // One contiguous band per rank: step 1 (begin / end are then the band itself); interleaved rows (rank r owns rows r, r + world, ...): begin = r,
    // step = world, end - begin = the number of rows owned.  Bins, counters and the raster grid are indexed by k ("owned row"), not by the row itself.
    uint32_t tile_row_begin, tile_row_end, tile_row_step;
    uint32_t num_draws, total_tris;
    uint32_t total_slots;             // geometry-kernel lanes (every draw padded to a multiple of 64)
    const DrawDesc* draws;
    const VsJob* vs_jobs; uint32_t num_vs_jobs, vs_total_slots;
    // depth key (DESIGN.md "Depth key"): zk = (bits(z) ^ zflip) & zmask ; idk = idflip ? MAX-id : id
    uint32_t zflip, zmask, idflip;
    uint32_t strict;                  // compare op is LESS / GREATER (ties with the stored depth fail)
    uint32_t pred;                    // != 0: predicate mode -- bit 0/1/2: a fragment passes when its depth is </==/> the scope's
                                      // initial depth; bit 3: the winner's depth replaces it (ALWAYS with depth write)
    uint32_t init_zk, init_idk;       // state of an uncovered pixel when depth is cleared
    uint32_t clear_depth_bits;
    float    clear_color[4];
    uint32_t clear_packed;            // clear colour in the target's B8G8R8A8_SRGB encoding (host-computed)
    uint32_t color_load;              // 1 = keep existing colour where nothing is drawn
    uint32_t color_format;            // mirhi_format
    void*    color;
    float*   depth;                   // optional D32 image (load and/or store)
    uint32_t depth_load, depth_store;
    uint32_t* prim_out;               // optional R32_UINT image
    // workspace
    BinRec*   bin_pool; uint32_t* bin_count; uint32_t bin_cap;     // bin_cap: records a tile's bin can hold (all of its lists)
    uint32_t* bin_table;              // [tiles][BIN_TABLE_ROW] pool page of the list's p-th page (PAGE_EMPTY until published)
    uint32_t* pool_next;              // [8][POOL_COUNTER_STRIDE] dynamic pages handed out in this scope by the waves of each XCD (re-armed by
                                      // the raster kernel): one counter per XCD, a cache line apart -- a single counter that every XCD adds
                                      // to ping-pongs between their L2s (C4: geometry 57 -> 132 us)
    uint32_t  pool_dyn_base, pool_dyn_pages;   // XCD x hands out pool pages [pool_dyn_base + x * pool_dyn_pages, ... + pool_dyn_pages)
    uint32_t  fixed_recs;             // slots [0, fixed_recs) of a tile's single list live in the tile's own fixed pages (page = tile *
                                      // fixed_recs / 64 + slot / 64): 64 x the pages the scope's average triangle density fills, 0 with per-XCD lists
    BigRec*   big_recs; uint32_t* big_count; uint32_t big_cap;
    uint32_t* big_count_next;         // the other parity's counter: zeroed by this scope for the next one
    uint32_t* prim_draw;              // per primitive of the scope (index prim - first_prim): its draw, written by the geometry kernel when
    uint32_t  first_prim;             // the scope has several draws -- one load in the resolve instead of a binary search per pixel
    uint32_t* flat_color;             // per primitive: B8G8R8A8_SRGB colour if its three vertex colours are equal (TRIANGLE
                                      // program, sRGB8 target), else 0; lets the resolve skip interpolation + OETF
    uint32_t tp_max_area;             // records whose pixel box inside the tile has at most this many pixels are resolved
                                      // triangle-parallel (LDS ds_min) instead of pixel-parallel
    uint32_t xcd_swizzle;             // run length G of consecutive tiles placed on one XCD (1 = plain order)
    uint32_t raster_teams;            // 2: the mesh variant with two teams of four waves per tile (host-side choice, see raster_kernel)
    // Per-XCD bins (only together with raster_teams == 2): a tile's bin is eight lists of sub_cap records, one per XCD,
    // each with its own counter (bin_count[xcd * count_stride + tile]) and its own eight entries of the tile's page-table row,
    // so that a hot tile's counter line stays in one XCD's L2 -- see reserve_bin_slots.  Off: count_stride = 0, sub_cap =
    // bin_cap, and the first page of every tile's single list has a fixed place in the pool (page = tile): the common bin
    // needs neither an allocation nor a table entry.
    uint32_t sub_cap, count_stride;
    uint32_t raster_wide;             // 1: the wide mesh variant, sixteen waves per tile (host-side choice from the busy-tile count, see raster_body)
    uint32_t resolve_flat_only;       // 1: a wave whose covered pixels all carry a packed flat colour may store and leave (flat colours exist, depth was not loaded,
                                      // the target is 8-bit, no primitive-id image, no depth store) -- the four conditions as one host-computed word: one scalar load
                                      // in the resolve instead of four dependent ones
    // Busy-tile count (feedback for that choice): every raster workgroup whose tile holds something adds 1 to active[(tile & 7) * 32] (eight
    // counters, one per XCD residue, a cache line apart); the workgroup of tile 0 reports the sum of the PREVIOUS scope's counters
    // (active_prev: the other parity, complete by then) in status[3] and re-arms them.
    uint32_t* active; uint32_t* active_prev;
    // ordered segments (blending; any depth state whose result depends on the order of all fragments): the geometry kernel
    // writes triangle t of the segment to ordered_recs[t] instead of binning it, the ordered kernel walks that array
    TriRec*  ordered_recs; uint32_t ordered_first, ordered_count;
    uint32_t ord_depth_test, ord_depth_write, ord_depth_op;
    uint32_t blend[8];                // enable, src colour, dst colour, colour op, src alpha, dst alpha, alpha op, write mask
    uint32_t alpha_scope;             // 1: a plain (bins + depth key) scope whose pipelines set fragment_discard_enable: records of draws whose texel
                                      // alpha straddles the material's cutoff are resolved triangle-parallel with the alpha test per pixel (raster_small_masked)
    uint32_t* status;                 // pinned host memory: [0] status bits (atomicOr), [1] big-list length, [2] dynamic pages of the last scope,
                                      // [3] busy tiles of the scope before it | 0x80000000
    unsigned long long* frag_stats;   // device counters of the statistics pass (never touched by geometry / raster kernels): [0] pixels that
                                      // ran a fragment program (winners of the depth resolve), [1] fragments covered before the depth test
};

// Kernel arguments passed by value next to the PassParams pointer: what a wave needs before anything else, so that its
// first dependent loads (draw table, bin counter -> bin records) hang off the kernarg load, not off a second memory hop.
struct GeometryHead {
    const DrawDesc* draws; uint32_t num_draws;
    uint32_t tris_per_wave;          // triangles a 64-lane workgroup takes: 64, or 32 / 16 for small scopes (the other lanes help with the pairs)
    // Scopes of ONE non-indexed draw of the TRIANGLE program (the reference's own frame, BASELINE configs[0] and [1]): where its vertices are, by value, so
    // that a wave's vertex loads hang off the kernarg load alone -- beside the descriptor's scalar loads instead of behind them (0.85 us of a wave's 9).
    // vb0 == nullptr: the descriptor says where (every other scope).
    const uint8_t* vb0; uint32_t stride0, first0, tris0;
};
struct RasterHead {
    uint32_t* bin_count; const BinRec* bin_pool; uint32_t* big_count;
    uint32_t tiles_x, tile_row_begin, bin_cap, big_cap;
    uint32_t sub_cap, count_stride;   // per-XCD bins (PassParams); read by the two-team variant only
    uint32_t fixed_recs;              // PassParams::fixed_recs
    uint32_t tile_row_step;           // PassParams::tile_row_step: grid row k is tile row tile_row_begin + k * tile_row_step
};

// A batched launch: the independent rendering scopes of one mirhi_queue_submit (equal target shape, equal raster variant) share
// one vertex, one geometry and one raster launch; their arguments travel by value in the kernarg segment.
constexpr int MAX_BATCH = 8;
struct GeometryBatch { const PassParams* params[MAX_BATCH]; GeometryHead head[MAX_BATCH]; uint32_t blocks[MAX_BATCH]; };
struct RasterBatch { const PassParams* params[MAX_BATCH]; RasterHead head[MAX_BATCH]; };

}  // namespace mirhi
