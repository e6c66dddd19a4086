// mirhi_device.h -- structures shared by the host C-ABI layer and the gfx950 kernels.
//
// Data layout in HBM (see DESIGN.md "Data layout"):
//   DrawDesc[]   one per recorded draw of a rendering scope (pointers into caller buffers + state)
//   TileRec[]    per-tile bins: tiles * bin_cap records of 80 B, written by the geometry kernel,
//                consumed by exactly one raster workgroup
//   BigRec[]     screen-space triangles too large for the bins (or spilled from a full bin);
//                every raster workgroup scans this list
//   counters     bin_count[tiles], big_count, status word
#pragma once
#include <stdint.h>

namespace mirhi {

constexpr int TILE = 32;              // screen tile edge in pixels (one raster workgroup)
constexpr int TILE_LOG2 = 5;
constexpr int BLOCK = 8;              // coverage block edge (one wave-iteration covers 8x8 px)
constexpr int RASTER_THREADS = 256;   // 4 waves; wave q owns the 16x16 quadrant q, 4 px per lane
constexpr int GEOM_THREADS = 256;
constexpr int MAX_BIN_SPAN = 4;       // triangles spanning more than 4x4 tiles go to the big list
constexpr float GUARD_PX = 16000.0f;  // guard band: snapped coordinates stay inside +-2^22 sub-pixels
constexpr uint32_t NO_PRIM = 0xFFFFFFFFu;
constexpr uint32_t MAX_PRIM_ID = 0xFFFFFFFDu;

enum : uint32_t { STATUS_BIG_OVERFLOW = 1u };

struct DrawDesc {
    const uint8_t* vb;            // binding 0 base + bind offset
    const uint8_t* ib;            // index base + bind offset (nullptr for draw)
    const float*   camera;        // b0 CameraData
    const float*   object;        // b1 ObjectData
    const uint8_t* lights;        // b2 LightUBO
    const uint8_t* material;      // b3 MaterialData
    const uint8_t* point_lights;  // t0 space1
    const uint8_t* spot_lights;   // t1 space1
    const uint8_t* tex[2];        // t0 albedo, t1 normal (RGBA8)
    uint32_t tex_w[2], tex_h[2];
    uint32_t stride;
    uint32_t index_type;          // 0 none, 2 u16, 4 u32
    uint32_t first;               // first_vertex / first_index
    int32_t  vertex_offset;
    uint32_t tri_count;
    uint32_t prim_base;           // global primitive id of triangle 0
    uint32_t program;
    uint32_t cull_mode, front_face;
    float    hw, hh, cx, cy;      // viewport half extents and centre
    float    dscale, dmin;        // depth range
    float    gx, gy;              // guard-band plane factors
    int32_t  sx0, sy0, sx1, sy1;  // inclusive scissor (already clamped to render area and extent)
    uint32_t scissor_partial;     // scissor smaller than the target: per-pixel box test needed when it cuts a bbox
    uint32_t pad[2];
};
static_assert(sizeof(DrawDesc) % 16 == 0, "DrawDesc must stay 16-byte sized");

// One triangle as seen by one 32x32 tile. 20 dwords; the first 16 are read for every triangle.
struct TileRec {
    int32_t  Q[3];       // floor((E_i(tile origin pixel centre) + bias_i) / 256), clamped to +-2^30
    int32_t  A0;         // A_i = Ya - Yb, B_i = Xb - Xa in 1/256 px (|.| < 2^23)
    int32_t  A1, A2, B0, B1;
    int32_t  B2;
    float    x0f, y0f;   // snapped vertex 0 in pixels (exact)
    float    z0;
    float    zx, zy;     // depth plane
    uint32_t idk;        // primitive id in tie-break order
    uint32_t mask;       // bits 0..15: 8x8 blocks of the tile the triangle may touch; bit 31: apply box
    uint32_t box;        // tile-relative inclusive pixel box minx | maxx<<8 | miny<<16 | maxy<<24
    uint32_t pad[3];
};
static_assert(sizeof(TileRec) == 80, "TileRec is 80 bytes");

// Orientation-normalised snapped triangle in screen space. 16 dwords.
struct BigRec {
    int32_t  X0, Y0, X1, Y1, X2, Y2;   // 1/256 px
    float    z0, zx, zy;
    uint32_t idk;
    uint32_t bx;         // pixel bbox minx | maxx<<16 (scissor-clamped, inclusive)
    uint32_t by;         // miny | maxy<<16
    uint32_t boxed;      // 1 if the scissor cut the vertex bbox
    uint32_t pad[3];
};
static_assert(sizeof(BigRec) == 64, "BigRec is 64 bytes");

struct PassParams {
    uint32_t width, height;           // colour target extent
    uint32_t tiles_x, tiles_y;
    uint32_t tile_row_begin, tile_row_end;   // band of tile rows rasterized on this device
    uint32_t num_draws, total_tris;
    const DrawDesc* draws;
    // depth key (DESIGN.md "Depth key"): zk = (bits(z) ^ zflip) & zmask ; idk = idflip ? MAX-id : id
    uint32_t zflip, zmask, idflip;
    uint32_t strict;                  // compare op is LESS / GREATER (ties with the stored depth fail)
    uint32_t init_zk, init_idk;       // state of an uncovered pixel when depth is cleared
    uint32_t clear_depth_bits;
    float    clear_color[4];
    uint32_t color_load;              // 1 = keep existing colour where nothing is drawn
    uint32_t color_format;            // mirhi_format
    void*    color;
    float*   depth;                   // optional D32 image (load and/or store)
    uint32_t depth_load, depth_store;
    uint32_t* prim_out;               // optional R32_UINT image
    // workspace
    TileRec*  bin_recs; uint32_t* bin_count; uint32_t bin_cap;
    BigRec*   big_recs; uint32_t* big_count; uint32_t big_cap;
    uint32_t* status;
};

}  // namespace mirhi
