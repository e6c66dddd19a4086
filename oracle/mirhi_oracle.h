/*
 * mirhi_oracle.h -- CPU restatement of the reference's draw path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the parity oracle for the MI355X compute rasterizer.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product path (libmirhi.so) never links, imports or calls anything here.
 *
 * PARITY STATUS: the reference (itsakeyfut/renderer-rs) executes
 * this path inside a Vulkan driver + GPU fixed-function hardware and cannot be
 * built or run in this container (no rustc/cargo, no Vulkan loader/ICD, no DXC).
 * The oracle is therefore a restatement of the *specification* the reference
 * configures:
 *   - vertex formats            crates/rhi/src/vertex.rs:20-61,88-170
 *   - draw / draw_indexed       crates/rhi/src/command.rs:583-628
 *   - pipeline fixed function   crates/rhi/src/pipeline.rs:645-698,960-1057
 *   - attachments / clears      crates/rhi/src/rendering.rs:102-115,356-370
 *   - frame recording           crates/renderer/src/renderer.rs:452-557
 *   - shaders                   shaders/hlsl/{vertex,pixel}/{triangle,model}.hlsl,
 *                               shaders/hlsl/pixel/model_full.hlsl, shaders/hlsl/lights.hlsli
 *   - glam 0.30.9 (Cargo.lock; not vendored): published definitions restated in oracle_glam_*.
 * It is pinned by the reference's own known answers (tests/golden/kats.json: screenshot
 * statistics K1, analytic hello-triangle coverage K2, shader constants K3, matrix values
 * K4, asset counts K5, struct layouts K6).  Raster coverage / interpolation / shading
 * output beyond those KATs is "parity unpinned" by any reference test (the reference has
 * none that renders a pixel).
 */
#ifndef MIRHI_ORACLE_H
#define MIRHI_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORACLE_PROGRAM_TRIANGLE = 0, ORACLE_PROGRAM_MODEL = 1, ORACLE_PROGRAM_MODEL_FULL = 2, ORACLE_PROGRAM_MODEL_PBR = 3 };
enum { ORACLE_CULL_NONE = 0, ORACLE_CULL_FRONT = 1, ORACLE_CULL_BACK = 2, ORACLE_CULL_FRONT_AND_BACK = 3 };
enum { ORACLE_FRONT_CCW = 0, ORACLE_FRONT_CW = 1 };
/* pipeline.rs:411-448 BlendFactor order, :452-476 BlendOp order */
enum { ORACLE_BF_ZERO = 0, ORACLE_BF_ONE, ORACLE_BF_SRC_COLOR, ORACLE_BF_ONE_MINUS_SRC_COLOR, ORACLE_BF_DST_COLOR, ORACLE_BF_ONE_MINUS_DST_COLOR,
       ORACLE_BF_SRC_ALPHA, ORACLE_BF_ONE_MINUS_SRC_ALPHA, ORACLE_BF_DST_ALPHA, ORACLE_BF_ONE_MINUS_DST_ALPHA, ORACLE_BF_CONSTANT_COLOR,
       ORACLE_BF_ONE_MINUS_CONSTANT_COLOR, ORACLE_BF_CONSTANT_ALPHA, ORACLE_BF_ONE_MINUS_CONSTANT_ALPHA, ORACLE_BF_SRC_ALPHA_SATURATE };
enum { ORACLE_BO_ADD = 0, ORACLE_BO_SUBTRACT, ORACLE_BO_REVERSE_SUBTRACT, ORACLE_BO_MIN, ORACLE_BO_MAX };
/* crates/rhi/src/pipeline.rs:375-409 CompareOp order */
enum { ORACLE_CMP_NEVER = 0, ORACLE_CMP_LESS = 1, ORACLE_CMP_EQUAL = 2, ORACLE_CMP_LESS_OR_EQUAL = 3,
       ORACLE_CMP_GREATER = 4, ORACLE_CMP_NOT_EQUAL = 5, ORACLE_CMP_GREATER_OR_EQUAL = 6, ORACLE_CMP_ALWAYS = 7 };

typedef struct {
    const uint8_t* rgba8; /* row-major, 4 bytes per texel; level 0 first, then levels 1..levels-1 back to back */
    uint32_t width, height;
    /* SURVEY 8f rank 3 (texture fidelity): levels > 1 = a mip chain follows level 0 (each level max(1, w>>1) x max(1, h>>1))
     * and the texture is sampled trilinearly; srgb = the RGB bytes are sRGB-encoded and decoded to linear when sampled.
     * 0 levels means 1.  max_anisotropy > 1 (at most 16; only with a chain): anisotropic filtering, see sample_texture. */
    uint32_t levels, srgb, max_anisotropy;
} oracle_texture;

typedef struct {
    /* vertex fetch: binding 0, per-vertex rate (vertex.rs:35-41,130-136) */
    const uint8_t* vertex_data;
    uint32_t vertex_stride;     /* 24 (TriangleVertex) or 48 (Vertex) */
    /* index fetch (command.rs:471-482); index_type 0 = non-indexed draw, 2 = u16, 4 = u32 */
    const void* index_data;
    uint32_t index_type;
    uint32_t count;             /* vertex_count (draw) or index_count (draw_indexed) */
    uint32_t first;             /* first_vertex (draw) or first_index (draw_indexed) */
    int32_t  vertex_offset;     /* draw_indexed only */
    /* pipeline state (pipeline.rs:645-698) */
    uint32_t program;
    uint32_t cull_mode, front_face;
    uint32_t depth_test, depth_write, depth_compare;
    /* dynamic state (renderer.rs:504-518) */
    float    viewport[6];       /* x, y, width, height, min_depth, max_depth */
    int32_t  scissor[4];        /* x, y, width, height */
    /* uniforms: HLSL layouts (model.hlsl:5-19, lights.hlsli:17-55, model_full.hlsl:34-41) */
    const void* camera;         /* CameraData 208 B */
    const void* object;         /* ObjectData 128 B */
    const void* light_ubo;      /* LightUBO 48 B */
    const void* material;       /* MaterialData 32 B */
    const void* point_lights;   /* PointLight[NumPointLights], 32 B each */
    const void* spot_lights;    /* SpotLight[NumSpotLights], 48 B each */
    oracle_texture albedo_map, normal_map;
    /* MODEL_PBR only (pixel/model_pbr.hlsl:62-95): t2 metallic-roughness, t3 occlusion, t4 emissive; material is the 80 B block :36-59 */
    oracle_texture metallic_roughness_map, occlusion_map, emissive_map;
    /* colour blending (ColorBlendAttachment, crates/rhi/src/pipeline.rs:478-531; factors :411-448, ops :452-476).
     * blend_enable 0: opaque overwrite.  Fragments are blended in primitive order, in float, against the pixel's
     * current colour.  Factor / op values are the reference's enum order. */
    uint32_t blend_enable;
    uint32_t src_color_factor, dst_color_factor, color_op, src_alpha_factor, dst_alpha_factor, alpha_op, color_write_mask;
} oracle_draw;

typedef struct {
    uint32_t width, height;     /* colour attachment extent */
    float    clear_color[4];    /* rendering.rs:102-115 default (0,0,0,1) */
    float    clear_depth;       /* rendering.rs:356-370 default 1.0 */
    uint32_t num_draws;
    const oracle_draw* draws;
    /* optional tile-row band: render only rows [row_begin,row_end) (multi-GPU split); 0,0 = all */
    uint32_t row_begin, row_end;
} oracle_pass;

#define ORACLE_NO_PRIM 0xFFFFFFFFu

/* Renders the pass with nthreads row bands (1 = scalar).  Any output pointer may be NULL.
 *   out_rgba   : width*height*4 floats, linear (pre-quantisation) colour
 *   out_prim   : width*height winning global primitive id (ORACLE_NO_PRIM = clear)
 *   out_depth  : width*height stored depth
 *   out_bgra8  : width*height*4 bytes, B8G8R8A8_SRGB encoding (swapchain.rs:561-570)
 * returns 0 on success. */
int oracle_render(const oracle_pass* pass, int nthreads, float* out_rgba, uint32_t* out_prim,
                  float* out_depth, uint8_t* out_bgra8);

/* sRGB OETF + UNORM8 quantisation of one linear channel */
uint8_t oracle_srgb8(float linear);

/* lights.hlsli helpers exposed for the K3 known-answer tests */
float oracle_attenuation(float distance, float radius);       /* lights.hlsli:63-73 */
float oracle_roughness_to_shininess(float roughness);         /* lights.hlsli:152-159 */
void  oracle_blinn_phong(const float L[3], const float V[3], const float N[3], const float light_color[3],
                         const float albedo[3], float shininess, float out[3]); /* lights.hlsli:95-117 */

/* pbr.hlsli helpers exposed for known-answer tests */
float oracle_distribution_ggx(float n_dot_h, float roughness);   /* pbr.hlsli:55-69 */
float oracle_geometry_schlick_ggx(float n_dot_v, float roughness); /* pbr.hlsli:83-93 */

/* glam 0.30.9 restatements (column-major float[16]); call sites cited in SURVEY 8c */
void oracle_glam_perspective_rh(float fovy, float aspect, float z_near, float z_far, float out[16]);
void oracle_glam_orthographic_rh(float l, float r, float b, float t, float n, float f, float out[16]);
void oracle_glam_look_at_rh(const float eye[3], const float center[3], const float up[3], float out[16]);
void oracle_glam_mat4_mul(const float a[16], const float b[16], float out[16]);
void oracle_glam_from_scale_rotation_translation(const float s[3], const float q[4], const float t[3], float out[16]);
float oracle_glam_determinant(const float m[16]);
void oracle_glam_inverse(const float m[16], float out[16]);
void oracle_glam_transpose(const float m[16], float out[16]);
void oracle_glam_quat_mul_vec3(const float q[4], const float v[3], float out[3]);
/* crates/scene/src/camera.rs:110-142 */
void oracle_camera_view_matrix(const float position[3], const float rotation[4], float out[16]);
void oracle_camera_projection_perspective(float fovy, float aspect, float z_near, float z_far, float out[16]);
/* crates/resources/src/ubo.rs:243-259, crates/scene/src/transform.rs:163-179 */
void oracle_normal_matrix(const float model[16], float out[16]);

#ifdef __cplusplus
}
#endif
#endif
