/*
 * mirhi.h -- C ABI of the MI355X-native compute rasterizer (libmirhi.so).
 *
 * Drop-in boundary for the draw path of itsakeyfut/renderer-rs: every entry point below
 * replaces one public method of the reference's `crates/rhi` object surface (the reference has
 * no trait / plugin seam, SURVEY.md section 0.2, so the seam is the rhi structs' methods) or one step
 * of `crates/renderer`'s draw-submit loop.  Plain pointers, sizes and #[repr(C)]-compatible
 * structs only: a Rust `mirhi-sys` crate binds this header 1:1.
 *
 * Conventions
 *   - every function returns mirhi_result; MIRHI_OK == 0; on failure
 *     mirhi_last_error_message() returns a thread-local description whose text mirrors the
 *     reference's RhiError payload (crates/rhi/src/error.rs:6-50).
 *   - never aborts, never throws across the ABI.
 *   - threading contract = the reference's: a device is shared freely (device.rs:379-380),
 *     command buffers / recording are externally synchronised (command.rs:48-51).
 *   - there is NO CPU fallback: creating a device without a HIP GPU fails with
 *     MIRHI_ERR_NO_SUITABLE_GPU.
 */
#ifndef MIRHI_H
#define MIRHI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIRHI_ABI_VERSION 5u     /* 3: mirhi_pipeline_desc.fragment_discard_enable; 4: mirhi_device_set_submit_thread; 5: mirhi_device_set_native_dispatch,
                                    mirhi_device_dispatch_path, mirhi_device_measure_roundtrip, mirhi_build_id, mirhi_device_set_tile_split_layout,
                                    mirhi_device_split_rows, mirhi_device_stats grew four words */

/* ---- errors: one code per RhiError variant (crates/rhi/src/error.rs:6-50) ------------------------ */
typedef int32_t mirhi_result;
enum {
    MIRHI_OK = 0,
    MIRHI_ERR_DEVICE = 1,           /* RhiError::VulkanError  -> any HIP runtime failure */
    MIRHI_ERR_LOADING = 2,          /* RhiError::LoadingError -> HIP runtime / code object unavailable */
    MIRHI_ERR_ALLOCATOR = 3,        /* RhiError::AllocatorError -> hipMalloc failure */
    MIRHI_ERR_NO_SUITABLE_GPU = 4,  /* RhiError::NoSuitableGpu */
    MIRHI_ERR_SHADER = 5,           /* RhiError::ShaderError  -> unknown mirhi_program */
    MIRHI_ERR_SURFACE = 6,          /* RhiError::SurfaceError (unused: offscreen only) */
    MIRHI_ERR_SWAPCHAIN = 7,        /* RhiError::SwapchainError (unused: offscreen only) */
    MIRHI_ERR_INVALID_HANDLE = 8,   /* RhiError::InvalidHandle */
    MIRHI_ERR_PIPELINE = 9,         /* RhiError::PipelineError */
    MIRHI_ERR_LOCK_POISONED = 10,   /* RhiError::LockPoisoned */
    MIRHI_TIMEOUT = 11,             /* RhiError::VulkanError(vk::Result::TIMEOUT) from Fence::wait */
    MIRHI_NOT_READY = 12            /* vk::Result::NOT_READY from a fence status query */
};
const char* mirhi_last_error_message(void);
const char* mirhi_result_name(mirhi_result r);
uint32_t    mirhi_abi_version(void);

/* ---- opaque handles ------------------------------------------------------------------------------- */
typedef struct mirhi_device   mirhi_device;    /* rhi::Device        crates/rhi/src/device.rs:61-77 */
typedef struct mirhi_buffer   mirhi_buffer;    /* rhi::Buffer        crates/rhi/src/buffer.rs:124-135 */
typedef struct mirhi_image    mirhi_image;     /* swapchain image / DepthBuffer / texture (image.rs is a stub) */
typedef struct mirhi_pipeline mirhi_pipeline;  /* rhi::Pipeline      crates/rhi/src/pipeline.rs:161-168 */
typedef struct mirhi_cmd      mirhi_cmd;       /* rhi::CommandBuffer crates/rhi/src/command.rs:279-284 */
typedef struct mirhi_fence    mirhi_fence;     /* rhi::Fence         crates/rhi/src/sync.rs:134-137 */

/* ---- device: Device::new / wait_idle / Drop (device.rs:120-233,290-293,356-372) ------------------- */
mirhi_result mirhi_device_count(int32_t* out_count);                      /* physical_device.rs:202-254 */
mirhi_result mirhi_device_create(int32_t hip_ordinal, mirhi_device** out);
/* same, but all work is issued on an existing HIP stream (e.g. torch.cuda.current_stream().cuda_stream) */
mirhi_result mirhi_device_create_on_stream(int32_t hip_ordinal, void* hip_stream, mirhi_device** out);
/* Native dispatch (csrc/mirhi_native.h): a plain submit leaves the library as AQL packets on a ROCr queue of the lane's own, NOT on a HIP stream.  A device
 * made by mirhi_device_create does that on every lane.  A device made on the caller's stream keeps the promise above for queue lane 0 -- submits to lane 0
 * are HIP launches on `hip_stream`, ordered against whatever else the caller put there -- and dispatches natively only on the lanes the library created
 * (mirhi_device_set_queue_lanes), which were never ordered against that stream.  enable = 1 opts lane 0 in as well: the caller then orders its own stream
 * against the frames with fences / wait_idle (hipStreamSynchronize knows nothing about the device's queues).  enable = 0: back to the default. */
mirhi_result mirhi_device_set_native_dispatch(mirhi_device* dev, uint32_t enable);
mirhi_result mirhi_device_wait_idle(mirhi_device* dev);                   /* Device::wait_idle :290-293; also reports (once) the device-side
                                                                             status of frames submitted without a fence, as mirhi_fence_wait does */
mirhi_result mirhi_device_destroy(mirhi_device* dev);                     /* fails if children are alive */
mirhi_result mirhi_device_name(mirhi_device* dev, char* out, uint32_t out_len);
/* screen-tile-row split: this device rasterizes only the tile rows owned by `rank` of `world` (which ones: the
 * layout below); rank 0 / world 1 = whole frame.  Gathering the rows is mirhi_comm_all_gather_bands, or the caller's collective. */
mirhi_result mirhi_device_set_tile_split(mirhi_device* dev, uint32_t rank, uint32_t world);
/* Which tile rows a rank gets.  BANDS: one contiguous band of ceil(tile rows / world) rows per rank (the last rank's may be short).  INTERLEAVED (the
 * default; MIRHI_SPLIT=bands|interleaved in the environment sets another default): rank r owns tile rows r, r + world, r + 2 world, ... -- every rank then
 * holds the same share of every part of the frame, so the slowest rank is the average one whatever the scene puts where ("interleave bands").
 * Every rank of a communicator must use the same layout; set it before mirhi_device_set_tile_split / mirhi_comm_create. */
typedef enum { MIRHI_SPLIT_BANDS = 0, MIRHI_SPLIT_INTERLEAVED = 1 } mirhi_split_layout;
mirhi_result mirhi_device_set_tile_split_layout(mirhi_device* dev, mirhi_split_layout layout);
/* the tile rows (32 pixel rows each, the last one of a frame possibly short) this device rasterizes of a frame `height` pixels high:
 * rows first_tile_row + k * tile_row_step, k = 0 .. tile_rows - 1 */
mirhi_result mirhi_device_split_rows(mirhi_device* dev, uint32_t height, uint32_t* first_tile_row, uint32_t* tile_row_step, uint32_t* tile_rows);
/* frames in flight (crates/renderer/src/lib.rs:43 MAX_FRAMES_IN_FLIGHT): command buffers are assigned round-robin to
 * `lanes` submit streams at creation, so independent frames (own command buffer, own target) overlap on the GPU the way
 * the reference's per-frame command buffers do between their semaphores.  Default 1 = strict submission order.  Set before
 * creating command buffers.  Work submitted on different lanes is unordered unless a fence is waited. */
mirhi_result mirhi_device_set_queue_lanes(mirhi_device* dev, uint32_t lanes);
/* Submit thread (default off).  vkQueueSubmit hands its work to the driver and returns (renderer.rs:407-424); with the thread on,
 * mirhi_queue_submit validates the submission, queues it and returns, and a thread of the device makes the kernel launches (HIP takes
 * 2.3 - 3 us of host time per launch whatever the entry point: 5 - 7 us per frame the render thread can spend recording the next
 * frame instead).  Everything else keeps its meaning: a fence waits for its submission, wait_idle and every call that reads or writes
 * a resource first wait until all queued submissions have been issued; an error the launches raise is reported by the submission's
 * fence (or by wait_idle if it has none).  Submissions are issued in the order they were made. */
mirhi_result mirhi_device_set_submit_thread(mirhi_device* dev, uint32_t enable);
/* first/last+1 pixel row of the band rendered by this device for a target of `height` rows (MIRHI_SPLIT_BANDS; with interleaved rows there is no single band:
 * InvalidHandle -- use mirhi_device_split_rows) */
mirhi_result mirhi_device_band_rows(mirhi_device* dev, uint32_t height, uint32_t* row_begin, uint32_t* row_end);

/* ---- buffers: BufferUsage + Buffer (buffer.rs:47-112,149-293,345-417) ------------------------------ */
typedef enum {
    MIRHI_BUFFER_VERTEX = 0, MIRHI_BUFFER_INDEX = 1, MIRHI_BUFFER_UNIFORM = 2,
    MIRHI_BUFFER_STORAGE = 3, MIRHI_BUFFER_STAGING = 4, MIRHI_BUFFER_INDIRECT = 5
} mirhi_buffer_usage;
mirhi_result mirhi_buffer_create(mirhi_device* dev, mirhi_buffer_usage usage, uint64_t size, mirhi_buffer** out); /* Buffer::new :149 (size 0 -> InvalidHandle) */
mirhi_result mirhi_buffer_create_with_data(mirhi_device* dev, mirhi_buffer_usage usage, const void* data, uint64_t len, mirhi_buffer** out); /* Buffer::new_with_data :227 */
mirhi_result mirhi_buffer_write(mirhi_buffer* buf, uint64_t offset, const void* data, uint64_t len);   /* Buffer::write_data :247 (bounds-checked; Storage/Indirect are "not mapped") */
mirhi_result mirhi_buffer_upload(mirhi_buffer* buf, const void* data, uint64_t len);                    /* Buffer::upload :291 */
mirhi_result mirhi_buffer_upload_via_staging(mirhi_buffer* buf, const void* data, uint64_t len);        /* Buffer::upload_via_staging :345 */
/* wrap memory that is already resident in HBM (e.g. a torch tensor); not freed on destroy */
mirhi_result mirhi_buffer_wrap_device_memory(mirhi_device* dev, mirhi_buffer_usage usage, void* device_ptr, uint64_t size, mirhi_buffer** out);
mirhi_result mirhi_buffer_read(mirhi_buffer* buf, uint64_t offset, void* dst, uint64_t len);           /* added: reference has no readback */
uint64_t     mirhi_buffer_size(const mirhi_buffer* buf);                                                /* Buffer::size :409 */
int32_t      mirhi_buffer_usage_of(const mirhi_buffer* buf);                                            /* Buffer::usage :415 */
void*        mirhi_buffer_device_ptr(const mirhi_buffer* buf);                                          /* Buffer::handle :403 */
mirhi_result mirhi_buffer_destroy(mirhi_buffer* buf);

/* ---- images: colour targets, DepthBuffer, textures -------------------------------------------------
 * formats: swapchain.rs:561-570 (B8G8R8A8_SRGB), depth_buffer.rs:48 (D32_SFLOAT); RGBA32F is the
 * parity target (linear, pre-quantisation); RGBA8_UNORM is the sampled-texture format. */
typedef enum {
    MIRHI_FORMAT_UNDEFINED = 0,
    MIRHI_FORMAT_B8G8R8A8_SRGB = 1,
    MIRHI_FORMAT_R32G32B32A32_SFLOAT = 2,
    MIRHI_FORMAT_D32_SFLOAT = 3,
    MIRHI_FORMAT_R8G8B8A8_UNORM = 4,
    MIRHI_FORMAT_R32_UINT = 5,      /* debug/parity: winning primitive id per pixel */
    MIRHI_FORMAT_R8G8B8A8_SRGB = 6  /* sampled colour textures: RGB decoded to linear on sampling */
} mirhi_format;
mirhi_result mirhi_image_create(mirhi_device* dev, uint32_t width, uint32_t height, mirhi_format format, mirhi_image** out); /* DepthBuffer::new depth_buffer.rs:117-127 (0 size -> error) */
mirhi_result mirhi_image_wrap_device_memory(mirhi_device* dev, uint32_t width, uint32_t height, mirhi_format format, void* device_ptr, mirhi_image** out);
mirhi_result mirhi_image_upload(mirhi_image* img, const void* src, uint64_t len);
mirhi_result mirhi_image_read(mirhi_image* img, void* dst, uint64_t len);   /* added: swapchain images have no readback (swapchain.rs:255) */
/* Texture fidelity (image.rs / sampler.rs / texture.rs are stubs in the reference, the shaders assume
 * `SamplerState` filtering, model_full.hlsl:44-46): builds the full mip chain of an owned R8G8B8A8 image from its level 0
 * (2x2 box filter on the stored bytes, round half up, edge clamp for odd sizes).  A texture with a chain is sampled
 * trilinearly (LOD from the analytic screen-space UV derivatives), one without bilinearly.  Call again after an upload. */
mirhi_result mirhi_image_generate_mips(mirhi_image* img);
uint32_t     mirhi_image_mip_levels(const mirhi_image* img);
/* Sampler state of a texture (sampler.rs is a stub; Device::new enables `sampler_anisotropy`, device.rs:161-165):
 * max_anisotropy in [1, 16], 1 = plain trilinear (the default).  Takes effect on textures with a mip chain, for draws recorded
 * afterwards: N = min(ceil(Pmax / Pmin), max_anisotropy) trilinear taps along the longer axis of the pixel's footprint at
 * lambda = log2(Pmax / N), averaged (the example filter of the Vulkan specification, "Texel Anisotropic Filtering"). */
mirhi_result mirhi_image_set_max_anisotropy(mirhi_image* img, uint32_t max_anisotropy);
uint32_t     mirhi_image_max_anisotropy(const mirhi_image* img);
uint32_t     mirhi_image_width(const mirhi_image* img);
uint32_t     mirhi_image_height(const mirhi_image* img);
int32_t      mirhi_image_format(const mirhi_image* img);
uint64_t     mirhi_image_size_bytes(const mirhi_image* img);
void*        mirhi_image_device_ptr(const mirhi_image* img);
mirhi_result mirhi_image_destroy(mirhi_image* img);

/* ---- pipeline: GraphicsPipelineBuilder (pipeline.rs:590-1059) -------------------------------------- */
typedef enum {   /* replaces Shader::from_spirv_file (shader.rs:244-330): precompiled .hip programs */
    MIRHI_PROGRAM_NONE = -1,
    MIRHI_PROGRAM_TRIANGLE = 0,     /* vertex/triangle.hlsl + pixel/triangle.hlsl */
    MIRHI_PROGRAM_MODEL = 1,        /* vertex/model.hlsl + pixel/model.hlsl (hard-coded fallback light/material) */
    MIRHI_PROGRAM_MODEL_FULL = 2,   /* vertex/model.hlsl + pixel/model_full.hlsl + lights.hlsli */
    MIRHI_PROGRAM_MODEL_PBR = 3     /* vertex/model.hlsl + pixel/model_pbr.hlsl + pbr.hlsli (Cook-Torrance GGX; no shadow pass: shadow = 1) */
} mirhi_program;
typedef enum { MIRHI_TOPOLOGY_POINT_LIST = 0, MIRHI_TOPOLOGY_LINE_LIST = 1, MIRHI_TOPOLOGY_LINE_STRIP = 2,
               MIRHI_TOPOLOGY_TRIANGLE_LIST = 3, MIRHI_TOPOLOGY_TRIANGLE_STRIP = 4, MIRHI_TOPOLOGY_TRIANGLE_FAN = 5 } mirhi_topology;   /* pipeline.rs:274-300 */
typedef enum { MIRHI_POLYGON_FILL = 0, MIRHI_POLYGON_LINE = 1, MIRHI_POLYGON_POINT = 2 } mirhi_polygon_mode;                          /* :306-325 */
typedef enum { MIRHI_CULL_NONE = 0, MIRHI_CULL_FRONT = 1, MIRHI_CULL_BACK = 2, MIRHI_CULL_FRONT_AND_BACK = 3 } mirhi_cull_mode;        /* :329-351 */
typedef enum { MIRHI_FRONT_FACE_COUNTER_CLOCKWISE = 0, MIRHI_FRONT_FACE_CLOCKWISE = 1 } mirhi_front_face;                              /* :355-371 */
typedef enum { MIRHI_COMPARE_NEVER = 0, MIRHI_COMPARE_LESS = 1, MIRHI_COMPARE_EQUAL = 2, MIRHI_COMPARE_LESS_OR_EQUAL = 3,
               MIRHI_COMPARE_GREATER = 4, MIRHI_COMPARE_NOT_EQUAL = 5, MIRHI_COMPARE_GREATER_OR_EQUAL = 6, MIRHI_COMPARE_ALWAYS = 7 } mirhi_compare_op; /* :375-409 */

/* ColorBlendAttachment (pipeline.rs:478-531): BlendFactor :411-448, BlendOp :452-476, in the reference's enum order.  The
 * CONSTANT_* factors need blend constants, which the reference's command buffer has no call for: refused at pipeline create. */
typedef enum { MIRHI_BLEND_ZERO = 0, MIRHI_BLEND_ONE, MIRHI_BLEND_SRC_COLOR, MIRHI_BLEND_ONE_MINUS_SRC_COLOR, MIRHI_BLEND_DST_COLOR,
               MIRHI_BLEND_ONE_MINUS_DST_COLOR, MIRHI_BLEND_SRC_ALPHA, MIRHI_BLEND_ONE_MINUS_SRC_ALPHA, MIRHI_BLEND_DST_ALPHA,
               MIRHI_BLEND_ONE_MINUS_DST_ALPHA, MIRHI_BLEND_CONSTANT_COLOR, MIRHI_BLEND_ONE_MINUS_CONSTANT_COLOR, MIRHI_BLEND_CONSTANT_ALPHA,
               MIRHI_BLEND_ONE_MINUS_CONSTANT_ALPHA, MIRHI_BLEND_SRC_ALPHA_SATURATE } mirhi_blend_factor;
typedef enum { MIRHI_BLEND_OP_ADD = 0, MIRHI_BLEND_OP_SUBTRACT, MIRHI_BLEND_OP_REVERSE_SUBTRACT, MIRHI_BLEND_OP_MIN, MIRHI_BLEND_OP_MAX } mirhi_blend_op;

typedef struct {
    int32_t  vertex_program;            /* builder.vertex_shader();   MIRHI_PROGRAM_NONE -> "Vertex shader is required" */
    int32_t  fragment_program;          /* builder.fragment_shader(); MIRHI_PROGRAM_NONE -> "Fragment shader is required" */
    uint32_t vertex_stride;             /* vertex_binding(): 24 TriangleVertex / 48 Vertex (vertex.rs:35-41,130-136) */
    uint32_t attribute_count;           /* vertex_attributes(): byte offsets by location (vertex.rs:44-61,139-170) */
    uint32_t attribute_offsets[4];
    int32_t  topology;                  /* default TRIANGLE_LIST   pipeline.rs:655 */
    int32_t  polygon_mode;              /* default FILL            :659 */
    int32_t  cull_mode;                 /* default BACK            :660 */
    int32_t  front_face;                /* default COUNTER_CLOCKWISE :661 */
    uint32_t depth_clamp_enable;        /* default 0               :662 */
    uint32_t rasterizer_discard_enable; /* default 0               :663 */
    uint32_t depth_bias_enable;         /* default 0               :664 */
    uint32_t rasterization_samples;     /* default 1               :671 */
    uint32_t depth_test_enable;         /* default 1               :676 */
    uint32_t depth_write_enable;        /* default 1               :677 */
    int32_t  depth_compare_op;          /* default LESS            :678 */
    uint32_t blend_enable;              /* default 0               :499-512 */
    uint32_t blend_attachment_count;    /* default 0 = one default attachment per colour format :1007-1018 */
    uint32_t color_attachment_count;    /* default 0 -> "At least one color attachment format is required" */
    int32_t  color_attachment_formats[4];
    int32_t  depth_attachment_format;   /* default UNDEFINED (None) :690 */
    /* the colour attachment's ColorBlendAttachment, used when blend_enable != 0 (defaults :499-512: One, Zero, Add, One, Zero, Add,
     * RGBA).  Blended draws are resolved fragment by fragment in primitive order (DESIGN.md "Ordered segments"). */
    int32_t  src_color_blend_factor, dst_color_blend_factor, color_blend_op;
    int32_t  src_alpha_blend_factor, dst_alpha_blend_factor, alpha_blend_op;
    uint32_t color_write_mask;          /* bit 0 R, 1 G, 2 B, 3 A */
    /* default 0.  The MODEL_PBR fragment program ends fragments whose base-colour alpha is below the material's alphaCutoff
     * (`discard`, pixel/model_pbr.hlsl:176-179).  With a base colour texture that is a decision per fragment, taken before the depth
     * write: pipelines for alpha-masked materials (glTF alphaMode MASK) set this.  Their draws form a rendering-scope segment of their
     * own whose raster kernel tests alpha per covered pixel in front of the depth key (blended or predicate-depth-state ones are resolved
     * fragment by fragment in primitive order instead, DESIGN.md "Ordered segments").  Without it such a draw is refused loudly at the
     * fence ("alpha cutoff"); draws whose alpha cannot cross the cutoff (no texture, or cutoff <= 0) never need it. */
    uint32_t fragment_discard_enable;
} mirhi_pipeline_desc;
void         mirhi_pipeline_desc_default(mirhi_pipeline_desc* desc);                      /* GraphicsPipelineBuilder::new :645-698 */
mirhi_result mirhi_pipeline_create(mirhi_device* dev, const mirhi_pipeline_desc* desc, mirhi_pipeline** out); /* build :918-1057 */
mirhi_result mirhi_pipeline_destroy(mirhi_pipeline* p);

/* ---- command recording: CommandBuffer (command.rs:297-628) ------------------------------------------ */
typedef enum { MIRHI_LOAD_OP_LOAD = 0, MIRHI_LOAD_OP_CLEAR = 1, MIRHI_LOAD_OP_DONT_CARE = 2 } mirhi_load_op;
typedef enum { MIRHI_STORE_OP_STORE = 0, MIRHI_STORE_OP_DONT_CARE = 1 } mirhi_store_op;
typedef enum { MIRHI_INDEX_UINT16 = 0, MIRHI_INDEX_UINT32 = 1 } mirhi_index_type;

typedef struct {     /* RenderingConfig / ColorAttachment / DepthAttachment (rendering.rs:65-115,319-370,680-726) */
    mirhi_image* color_image;        /* required */
    int32_t      color_load_op;      /* default CLEAR  rendering.rs:106 */
    int32_t      color_store_op;     /* default STORE  :107 */
    float        clear_color[4];     /* default (0,0,0,1) :108-112 */
    mirhi_image* depth_image;        /* optional; NULL + a depth-testing pipeline keeps depth on chip only */
    int32_t      depth_load_op;      /* default CLEAR  :360 */
    int32_t      depth_store_op;     /* default DONT_CARE :361 */
    float        clear_depth;        /* default 1.0    :362-366 */
    int32_t      render_area[4];     /* x, y, width, height; width==0 -> full extent (rendering.rs:713-726) */
    mirhi_image* prim_id_image;      /* optional R32_UINT: winning global primitive id (parity instrumentation) */
} mirhi_rendering_info;
void mirhi_rendering_info_default(mirhi_rendering_info* info);

typedef struct { float x, y, width, height, min_depth, max_depth; } mirhi_viewport;  /* vk::Viewport, renderer.rs:504-512 */
typedef struct { int32_t x, y; uint32_t width, height; } mirhi_rect2d;              /* vk::Rect2D,   renderer.rs:514-518 */

/* descriptor stand-in: register slots of shaders/hlsl (model.hlsl:5-19, model_full.hlsl:27-50) */
typedef enum {
    MIRHI_SLOT_CAMERA = 0,        /* b0 CameraData 208 B */
    MIRHI_SLOT_OBJECT = 1,        /* b1 ObjectData 128 B */
    MIRHI_SLOT_LIGHTS = 2,        /* b2 LightUBO 48 B */
    MIRHI_SLOT_MATERIAL = 3,      /* b3 MaterialData 32 B (model_full.hlsl:34-41); 80 B for MODEL_PBR (model_pbr.hlsl:36-59) */
    MIRHI_SLOT_POINT_LIGHTS = 4,  /* t0,space1 StructuredBuffer<PointLight> */
    MIRHI_SLOT_SPOT_LIGHTS = 5,   /* t1,space1 StructuredBuffer<SpotLight> */
    MIRHI_SLOT_COUNT = 6
} mirhi_uniform_slot;
typedef enum { MIRHI_TEXTURE_ALBEDO = 0 /* t0 */, MIRHI_TEXTURE_NORMAL = 1 /* t1 */,
               MIRHI_TEXTURE_METALLIC_ROUGHNESS = 2 /* t2 */, MIRHI_TEXTURE_OCCLUSION = 3 /* t3 */, MIRHI_TEXTURE_EMISSIVE = 4 /* t4 (model_pbr.hlsl:62-95) */,
               MIRHI_TEXTURE_COUNT = 5 } mirhi_texture_slot;

mirhi_result mirhi_cmd_create(mirhi_device* dev, mirhi_cmd** out);                 /* CommandPool::new + CommandBuffer::new :89,:297 */
mirhi_result mirhi_cmd_destroy(mirhi_cmd* cmd);
/* Queue lane (mirhi_device_set_queue_lanes) this command buffer is submitted on; by default command buffers take the lanes round
 * robin in creation order.  A submit of several command buffers that are each one plain rendering scope of the same shape (the
 * frames of a frame loop) runs as ONE batch of launches on the first one's lane (vkQueueSubmit with several command buffers,
 * renderer.rs:407-424: no ordering between them is promised without a barrier). */
mirhi_result mirhi_cmd_set_queue_lane(mirhi_cmd* cmd, uint32_t lane);
mirhi_result mirhi_cmd_begin(mirhi_cmd* cmd);                                       /* begin :333 (ONE_TIME_SUBMIT) */
mirhi_result mirhi_cmd_begin_reusable(mirhi_cmd* cmd);                              /* begin_reusable :353 */
mirhi_result mirhi_cmd_end(mirhi_cmd* cmd);                                         /* end :372 */
mirhi_result mirhi_cmd_reset(mirhi_cmd* cmd);                                       /* reset :387 */
mirhi_result mirhi_cmd_begin_rendering(mirhi_cmd* cmd, const mirhi_rendering_info* info);  /* begin_rendering :408 */
mirhi_result mirhi_cmd_end_rendering(mirhi_cmd* cmd);                               /* end_rendering :417 */
mirhi_result mirhi_cmd_bind_pipeline(mirhi_cmd* cmd, mirhi_pipeline* pipeline);     /* bind_pipeline :433 */
mirhi_result mirhi_cmd_bind_vertex_buffers(mirhi_cmd* cmd, uint32_t first_binding, uint32_t count, mirhi_buffer* const* buffers, const uint64_t* offsets); /* :448 */
mirhi_result mirhi_cmd_bind_index_buffer(mirhi_cmd* cmd, mirhi_buffer* buffer, uint64_t offset, mirhi_index_type type); /* :471 */
mirhi_result mirhi_cmd_bind_uniform(mirhi_cmd* cmd, mirhi_uniform_slot slot, mirhi_buffer* buffer, uint64_t offset, uint64_t range); /* bind_descriptor_sets :493 + descriptor.rs:390-409 buffer_info */
mirhi_result mirhi_cmd_bind_texture(mirhi_cmd* cmd, mirhi_texture_slot slot, mirhi_image* image);   /* descriptor.rs:411-420 image_info */
mirhi_result mirhi_cmd_set_viewport(mirhi_cmd* cmd, const mirhi_viewport* viewport);  /* set_viewport :522 */
mirhi_result mirhi_cmd_set_scissor(mirhi_cmd* cmd, const mirhi_rect2d* scissor);      /* set_scissor :549 */
/* instance_count > 1 (at most 4096): the path has no instance-rate input (binding 0 is per-vertex, vertex.rs:35-41,130-136; no program
 * reads SV_InstanceID), so instance i draws the same primitives again behind instance i - 1; first_instance has nothing to offset. */
mirhi_result mirhi_cmd_draw(mirhi_cmd* cmd, uint32_t vertex_count, uint32_t instance_count, uint32_t first_vertex, uint32_t first_instance); /* draw :583 */
mirhi_result mirhi_cmd_draw_indexed(mirhi_cmd* cmd, uint32_t index_count, uint32_t instance_count, uint32_t first_index, int32_t vertex_offset, uint32_t first_instance); /* draw_indexed :610 */

/* draw_indirect :630 / draw_indexed_indirect :646 (VkDrawIndirectCommand: 4 x u32; VkDrawIndexedIndirectCommand: 4 x u32 + i32 vertexOffset at
 * word 3).  The arguments live in a device buffer; this build READS THEM WHEN THE COMMAND IS RECORDED (one synchronous device -> host copy of
 * draw_count x stride bytes) and records the equivalent direct draws -- Vulkan reads them when the command executes, so a command buffer
 * whose indirect arguments change afterwards must be recorded again.  stride: multiple of 4, >= 16 / 20 when draw_count > 1. */
mirhi_result mirhi_cmd_draw_indirect(mirhi_cmd* cmd, mirhi_buffer* buffer, uint64_t offset, uint32_t draw_count, uint32_t stride);
mirhi_result mirhi_cmd_draw_indexed_indirect(mirhi_cmd* cmd, mirhi_buffer* buffer, uint64_t offset, uint32_t draw_count, uint32_t stride);
/* push_constants / push_constants_bytes :732-769.  Validated as Vulkan does (offset and length multiples of 4, offset + length <= 128) and kept
 * with the command buffer; the programs on this path read none (no HLSL file of the reference declares a push-constant block). */
mirhi_result mirhi_cmd_push_constants(mirhi_cmd* cmd, uint32_t stage_flags, uint32_t offset, const void* data, uint32_t len);

/* ---- submit + sync: vkQueueSubmit (renderer.rs:407-424, frame_manager.rs:439-462), Fence (sync.rs:168-298) */
mirhi_result mirhi_queue_submit(mirhi_device* dev, uint32_t cmd_count, mirhi_cmd* const* cmds, mirhi_fence* fence /* may be NULL */);
mirhi_result mirhi_fence_create(mirhi_device* dev, uint32_t signaled, mirhi_fence** out);   /* Fence::new :168 */
mirhi_result mirhi_fence_wait(mirhi_fence* fence, uint64_t timeout_ns);                     /* Fence::wait :228 (UINT64_MAX = forever) */
mirhi_result mirhi_fence_reset(mirhi_fence* fence);                                         /* Fence::reset :264 */
mirhi_result mirhi_fence_status(mirhi_fence* fence);    /* MIRHI_OK = signaled, MIRHI_NOT_READY = unsignaled; Fence::is_signaled :294 */
mirhi_result mirhi_fence_destroy(mirhi_fence* fence);

/* ---- measurement: per-dispatch device time, fragment statistics ---------------------------------- */
typedef enum { MIRHI_KERNEL_GEOMETRY = 0, MIRHI_KERNEL_RASTER = 1, MIRHI_KERNEL_VERTEX = 2, MIRHI_KERNEL_FRAGMENT_COUNT = 3,
               MIRHI_KERNEL_COUNT = 4 } mirhi_kernel_id;
/* enable: 0 = off, or a mask of
 *   MIRHI_PROFILE_TIMING     every kernel dispatch carries its own event pair (hipExtLaunchKernelGGL start / stop events): the
 *                            duration is the dispatch's begin -> end on the GPU clock, as rocprofv3 --kernel-trace reports it;
 *                            no event-record commands enter the stream and nothing is subtracted
 *   MIRHI_PROFILE_FRAGMENTS  fragment statistics (below): one extra counting kernel per scope and a few instructions in the
 *                            resolve -- never combine with a throughput measurement */
enum { MIRHI_PROFILE_TIMING = 1, MIRHI_PROFILE_FRAGMENTS = 2 };
/* MIRHI_PROFILE_TIMING | MIRHI_PROFILE_ONE_LANE(k): only the dispatches of queue lane k are timed; the other lanes run as in an
 * untimed frame loop (a timed dispatch is completed through its own signal and overlaps its neighbours less than an untimed one) */
#define MIRHI_PROFILE_ONE_LANE(k) ((((uint32_t)(k) + 1u) & 0xFFu) << 8)
mirhi_result mirhi_device_set_profiling(mirhi_device* dev, uint32_t enable);
/* accumulated since the last reset; waits for outstanding dispatches */
mirhi_result mirhi_device_kernel_time(mirhi_device* dev, mirhi_kernel_id kernel, double* total_ms, uint64_t* launches);
/* every timed dispatch since the last reset, in submission order: begin / end in microseconds on one GPU time axis that starts
 * at the begin of the first of them -- overlap between the queue lanes' kernels and the gaps between dependent launches can be
 * read off directly.  Writes min(capacity, *count) records; `out` may be NULL to query the count. */
typedef struct { uint32_t kernel /* mirhi_kernel_id */, lane; double begin_us, end_us; } mirhi_dispatch_time;
mirhi_result mirhi_device_timeline(mirhi_device* dev, mirhi_dispatch_time* out, uint32_t capacity, uint32_t* count);
/* SURVEY 8d "shaded Mpix/s ... report overdraw separately", summed over the scopes rendered with MIRHI_PROFILE_FRAGMENTS since
 * the last reset: shaded_pixels = pixels whose fragment program ran (the winners of the depth resolve: this design shades
 * visible pixels only), covered_fragments = pixel centres covered by a triangle before any depth test (what a GPU's
 * rasterizer emits); overdraw = covered_fragments / shaded_pixels.  Blended (ordered) segments are not counted. */
mirhi_result mirhi_device_fragment_stats(mirhi_device* dev, uint64_t* shaded_pixels, uint64_t* covered_fragments, uint64_t* scopes);
mirhi_result mirhi_device_reset_kernel_times(mirhi_device* dev);   /* also clears the timeline and the fragment statistics */
typedef struct {
    uint64_t frames_submitted;      /* rendering scopes executed */
    uint64_t triangles_submitted;   /* input triangles over those scopes */
    uint64_t workspace_bytes;       /* HBM held for bins / records */
    uint32_t last_big_list;         /* triangles that took the large/overflow list in the last finished scope */
    uint32_t last_status;           /* device status word of the last finished scope (0 = ok; bit 2: the bin pool ran out and is grown) */
    uint32_t last_bin_pages;        /* 2 KB bin pages the last finished scope took from the pool (beyond each tile's fixed first page) */
    uint32_t native_dispatches;     /* kernels this device dispatched as AQL packets on its own ROCr queues (csrc/mirhi_native.h) instead of through
                                       HIP launches (low 32 bits of the count); 0 on a device whose native dispatcher could not start */
    uint32_t dispatch_path;         /* how a plain submit leaves the library: 0 HIP launches (mirhi_device_dispatch_path says why), 1 AQL packets on the
                                       device's own hardware queues, 2 AQL packets on a queue a tool intercepts (MIRHI_NATIVE_DISPATCH=2 only) */
    uint32_t device_lost;           /* 1: a wait on one of the device's queues ran into its deadline; every later submit and wait fails with VulkanError */
    uint32_t reserved;
} mirhi_device_stats;
mirhi_result mirhi_device_get_stats(mirhi_device* dev, mirhi_device_stats* out);
/* "native: ..." or "hip: <why the native dispatcher is not used>", NUL-terminated into out[0 .. out_len) */
mirhi_result mirhi_device_dispatch_path(mirhi_device* dev, char* out, uint32_t out_len);
/* Measurement: round trip of ONE dispatch on queue lane `lane`, host store of the packet -> host sees its completion signal, averaged over `reps` dispatches
 * one at a time: [0] an empty one-wave kernel (doorbell -> packet processor -> wave -> end-of-kernel release -> signal -> host), [1] a barrier packet (no
 * wave).  What a fence-gated frame pays around its kernels.  Fails with LoadingError on a device without native dispatch. */
mirhi_result mirhi_device_measure_roundtrip(mirhi_device* dev, uint32_t lane, uint32_t reps, double* out_us /* [2] */);
/* the build this library is: first 16 hex digits of the sha256 over csrc/ and include/mirhi.h (renderer-rs_amd/build.py::source_hash).  The code object the
 * native dispatcher loads (libmirhi_kernels.hsaco) carries the same id and is refused if it differs. */
const char* mirhi_build_id(void);

/* ---- multi-GPU: screen-tile-row split + exchange of the finished RGBA bands over RCCL / xGMI ------------------
 * The reference drives one VkDevice (crates/rhi/src/device.rs:61-77) and has nothing to mirror here; BASELINE.json's north_star
 * defines the split.  One process per GPU: every rank creates its device, calls mirhi_device_set_tile_split(rank, world)
 * (mirhi_comm_create does it), renders -- only its band of 32-pixel tile rows is rasterized -- and calls
 * mirhi_comm_all_gather_bands on the frame: afterwards every rank holds the whole image.  librccl is loaded with dlopen the
 * first time one of these functions runs: a single-GPU host never loads or initialises RCCL. */
typedef struct mirhi_comm mirhi_comm;
#define MIRHI_COMM_ID_BYTES 128                     /* = NCCL_UNIQUE_ID_BYTES */
/* rank 0: fills `id` (ncclGetUniqueId); ship the 128 bytes to the other ranks over any host channel (the launcher's) */
mirhi_result mirhi_comm_unique_id(uint8_t* id /* [MIRHI_COMM_ID_BYTES] */);
/* collective over all `world` ranks (ncclCommInitRank); also applies mirhi_device_set_tile_split(dev, rank, world) */
mirhi_result mirhi_comm_create(mirhi_device* dev, const uint8_t* id, uint32_t rank, uint32_t world, mirhi_comm** out);
uint32_t mirhi_comm_world(const mirhi_comm* comm);  /* ranks RCCL counts in the communicator (ncclCommCount) */
uint32_t mirhi_comm_rank(const mirhi_comm* comm);
typedef enum {
    MIRHI_GATHER_DIRECT = 0,      /* one grouped batch of ncclSend / ncclRecv: every rank sends its band straight to every peer --
                                     xGMI is a full mesh of point-to-point links, so the 7 transfers of a rank run in parallel
                                     (4K BGRA8 on 8 GPUs: 4.15 MB per link) where a ring would pass 7 hops one after another */
    MIRHI_GATHER_BROADCAST = 1    /* one grouped batch of `world` in-place ncclBroadcast, one band each; RCCL picks the algorithm */
} mirhi_gather_algo;
/* In place on `frame` (a colour image every rank created with the same extent and format): the rows this rank rendered -- its band, or with interleaved
 * rows one 32-row piece per tile row it owns (mirhi_device_split_rows) -- are sent, the other ranks' rows received, all pieces in ONE RCCL group.  Shares may
 * differ in size (the last band / tile row is short when the rows do not divide).  Enqueued on the queue lane of `after` (the command buffer that rendered the
 * frame; NULL = lane 0), so it runs behind that frame's raster kernel; completion through mirhi_device_wait_idle or a later
 * submit on the same lane. */
mirhi_result mirhi_comm_all_gather_bands(mirhi_comm* comm, mirhi_image* frame, mirhi_cmd* after, mirhi_gather_algo algo);
mirhi_result mirhi_comm_destroy(mirhi_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* MIRHI_H */
