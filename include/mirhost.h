/*
 * mirhost.h -- C ABI of the host-side frame loop (libmirhost.so; C++ in renderer-rs_amd/host/frame_loop.cpp over host/mirhi.hpp).
 *
 * The reference's draw-submit loop is compiled host code (Rust): Renderer::render_frame, crates/renderer/src/renderer.rs:367-449,
 * with FrameManager (crates/renderer/src/frame_manager.rs:299-539) and record_commands (renderer.rs:452-557).  Rust is not in this
 * image, so the same loop is C++ on top of include/mirhi.h; this header lets a caller that holds mirhi handles (the Python tests,
 * bench.py) hand a frame description over and have the loop run natively -- per frame:
 *     wait_for_fence(in_flight[current]) -> acquire_next_image -> reset_fence -> command_buffer.reset() -> begin()
 *     -> begin_rendering -> set_viewport -> set_scissor -> bind_pipeline -> bind_vertex_buffers [-> uniforms, textures, index buffer]
 *     -> draw / draw_indexed -> end_rendering -> end() -> queue_submit(fence) -> present -> current = (current + 1) % frames_in_flight
 * Every frame is RE-RECORDED, as the reference does; nothing is replayed.  Resources (pipelines, buffers, images) stay the caller's.
 */
#ifndef MIRHOST_H
#define MIRHOST_H

#include "mirhi.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {           /* one draw of the frame: what record_commands binds and draws (renderer.rs:504-548) */
    mirhi_pipeline* pipeline;
    mirhi_buffer*   vertex_buffer;  uint64_t vertex_offset_bytes;
    mirhi_buffer*   index_buffer;   uint64_t index_offset_bytes;  int32_t index_type;   /* index_buffer NULL: draw(); else mirhi_index_type */
    struct { mirhi_buffer* buffer; uint64_t offset, range; } uniforms[MIRHI_SLOT_COUNT];   /* buffer NULL: slot not bound */
    mirhi_image*    textures[MIRHI_TEXTURE_COUNT];
    mirhi_viewport  viewport;
    mirhi_rect2d    scissor;
    uint32_t        count, instance_count, first;   /* vertex_count / index_count, instances, first_vertex / first_index */
    int32_t         vertex_offset;
} mirhost_draw;

typedef struct {
    uint32_t frames_in_flight;            /* crates/renderer/src/lib.rs:43 MAX_FRAMES_IN_FLIGHT = 2 */
    uint32_t image_count;                 /* swapchain images: frames_in_flight + 1 in the reference (swapchain.rs:228-236) */
    mirhi_image* const* images;           /* colour targets, cycled by acquire_next_image */
    mirhi_image* depth;                   /* optional DepthBuffer (depth_buffer.rs); NULL + a depth-testing pipeline keeps depth on chip */
    float    clear_color[4];              /* renderer.rs:479-488 */
    float    clear_depth;
    uint32_t draw_count;
    const mirhost_draw* draws;
    /* frame f draws `count - 3 * (f % vary_triangles)` vertices / indices of draw 0 (0 or 1: every frame the same): a frame loop whose
     * triangle count changes from frame to frame -- the re-recorded command buffer then has another shape every time */
    uint32_t vary_triangles;
    uint32_t submit_thread;               /* 1: mirhi_device_set_submit_thread(dev, 1) while the loop exists (the device's previous setting is restored at destroy) */
    /* s + 1: uniform slot s of draw 0 gets one buffer PER FRAME IN FLIGHT (copies of the caller's, made at create), and every frame
     * rewrites its slot's copy before recording -- Buffer::write_data (crates/rhi/src/buffer.rs:247-279) on the frame's own uniform
     * buffer, the per-frame update of a camera / object block.  0: the caller's buffers are bound as they are, nothing is written */
    uint32_t per_frame_uniform;
    uint32_t reserved;
} mirhost_frame_desc;

typedef struct mirhost_frame_loop mirhost_frame_loop;
mirhi_result mirhost_frame_loop_create(mirhi_device* dev, const mirhost_frame_desc* desc, mirhost_frame_loop** out);
/* renders `frames` frames (Renderer::render_frame each), then waits for all frames in flight; *seconds = wall time of the whole call */
mirhi_result mirhost_frame_loop_run(mirhost_frame_loop* loop, uint64_t frames, double* seconds);
/* index into desc.images of the frame rendered last, and how many frames this loop has rendered */
mirhi_result mirhost_frame_loop_last_image(const mirhost_frame_loop* loop, uint32_t* image_index, uint64_t* frames_rendered);
/* host seconds per phase since the last call -- [0] fence wait, [1] recording (reset .. end_rendering), [2] end(), [3] submit -- and
 * switches the accounting on / off (four clock reads per frame while it is on) */
mirhi_result mirhost_frame_loop_phase_seconds(mirhost_frame_loop* loop, int32_t enable, double* out4 /* may be NULL */);
mirhi_result mirhost_frame_loop_destroy(mirhost_frame_loop* loop);
const char*  mirhost_last_error_message(void);

#ifdef __cplusplus
}
#endif
#endif /* MIRHOST_H */
