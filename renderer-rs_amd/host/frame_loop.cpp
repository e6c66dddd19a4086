// frame_loop.cpp -- the reference's draw-submit loop, natively (include/mirhost.h -> libmirhost.so).
//
// Renderer::render_frame (crates/renderer/src/renderer.rs:367-449) over FrameManager (frame_manager.rs:299-539), with
// record_commands (renderer.rs:452-557) generalised from the one hard-coded triangle to a list of draws the caller describes.
// Host code only: everything here goes through the C ABI of include/mirhi.h, exactly as the Rust crates would.
#include <chrono>
#include <string>
#include <vector>

#include "../../include/mirhost.h"
#include "mirhi.hpp"

namespace {
thread_local std::string g_error;
mirhi_result fail(mirhi_result code, const std::string& msg) { g_error = msg; return code; }
}  // namespace

struct mirhost_frame_loop {
    std::shared_ptr<mirhi::Device> device;
    mirhi::CommandPool pool;
    mirhi::FrameManager frames;
    mirhost_frame_desc desc;
    std::vector<mirhi_image*> images;
    std::vector<mirhost_draw> draws;
    uint64_t frame_number = 0;
    uint32_t last_image = 0;
    // desc.per_frame_uniform: one copy of draw 0's uniform buffer per frame in flight, rewritten by its frame
    std::vector<mirhi_buffer*> frame_uniforms;
    std::vector<uint8_t> uniform_bytes;

    mirhost_frame_loop(mirhi_device* dev, const mirhost_frame_desc& d)
        : device(mirhi::Device::borrow(dev)), pool(device, 0), frames(device, pool, d.frames_in_flight), desc(d),
          images(d.images, d.images + d.image_count), draws(d.draws, d.draws + d.draw_count) {
        desc.images = images.data(); desc.draws = draws.data();
        if (d.per_frame_uniform && !draws.empty()) {
            const auto& u = draws[0].uniforms[d.per_frame_uniform - 1];
            const uint64_t bytes = u.range ? u.range : mirhi_buffer_size(u.buffer) - u.offset;
            uniform_bytes.resize(bytes);
            mirhi::check(mirhi_buffer_read(u.buffer, u.offset, uniform_bytes.data(), bytes));
            for (uint32_t k = 0; k < d.frames_in_flight; k++) {
                mirhi_buffer* b = nullptr;
                mirhi::check(mirhi_buffer_create_with_data(dev, MIRHI_BUFFER_UNIFORM, uniform_bytes.data(), bytes, &b));
                frame_uniforms.push_back(b);
            }
        }
    }
    ~mirhost_frame_loop() { for (mirhi_buffer* b : frame_uniforms) (void)mirhi_buffer_destroy(b); }

    // renderer.rs:452-557
    void record_commands(const mirhi::CommandBuffer& cmd, uint32_t image_index, uint32_t slot) const {
        mirhi_rendering_info info;
        mirhi_rendering_info_default(&info);
        info.color_image = images[image_index];                            // :479-488 ColorAttachment, CLEAR / STORE
        for (int k = 0; k < 4; k++) info.clear_color[k] = desc.clear_color[k];
        if (desc.depth) { info.depth_image = desc.depth; info.clear_depth = desc.clear_depth; }
        else info.clear_depth = desc.clear_depth;
        cmd.begin_rendering(info);                                          // :498
        mirhi_cmd* h = cmd.handle();
        for (size_t di = 0; di < draws.size(); di++) {
            const mirhost_draw& d = draws[di];
            mirhi::check(mirhi_cmd_set_viewport(h, &d.viewport));           // :504-512
            mirhi::check(mirhi_cmd_set_scissor(h, &d.scissor));             // :514-518
            mirhi::check(mirhi_cmd_bind_pipeline(h, d.pipeline));           // :521-527
            mirhi_buffer* vbs[1] = {d.vertex_buffer}; const uint64_t offs[1] = {d.vertex_offset_bytes};
            mirhi::check(mirhi_cmd_bind_vertex_buffers(h, 0, 1, vbs, offs));   // :530-534
            for (int s = 0; s < MIRHI_SLOT_COUNT; s++) {
                if (!d.uniforms[s].buffer) continue;
                if (di == 0 && !frame_uniforms.empty() && (uint32_t)s + 1 == desc.per_frame_uniform)
                    mirhi::check(mirhi_cmd_bind_uniform(h, (mirhi_uniform_slot)s, frame_uniforms[slot], 0, uniform_bytes.size()));
                else
                    mirhi::check(mirhi_cmd_bind_uniform(h, (mirhi_uniform_slot)s, d.uniforms[s].buffer, d.uniforms[s].offset, d.uniforms[s].range));
            }
            for (int t = 0; t < MIRHI_TEXTURE_COUNT; t++)
                if (d.textures[t] || t < 2) mirhi::check(mirhi_cmd_bind_texture(h, (mirhi_texture_slot)t, d.textures[t]));
            uint32_t count = d.count;
            if (di == 0 && desc.vary_triangles > 1) {
                const uint32_t less = 3u * (uint32_t)(frame_number % desc.vary_triangles);
                count = count > less ? count - less : count;
            }
            if (d.index_buffer) {
                mirhi::check(mirhi_cmd_bind_index_buffer(h, d.index_buffer, d.index_offset_bytes, (mirhi_index_type)d.index_type));
                mirhi::check(mirhi_cmd_draw_indexed(h, count, d.instance_count, d.first, d.vertex_offset, 0));
            } else {
                mirhi::check(mirhi_cmd_draw(h, count, d.instance_count, d.first, 0));   // :541-548
            }
        }
        cmd.end_rendering();                                                // :551
    }

    // host time per phase, accumulated when the caller asked for it (mirhost_frame_loop_phase_seconds): fence wait, recording
    // (reset .. end_rendering), end() (the launch plan), submit
    bool timed = false;
    double phase[4] = {0, 0, 0, 0};
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

    // renderer.rs:367-449
    void render_frame() {
        const double t0 = timed ? now() : 0.0;
        frames.wait_for_frame();                                            // :371-374  wait_for_fence(in_flight_fences[current_frame])
        const double t1 = timed ? now() : 0.0;
        frames.acquire_next_image((uint32_t)images.size());                 // :377-390
        const uint32_t slot = (uint32_t)frames.current_frame_index();
        if (!frame_uniforms.empty())                                        // this frame's uniform block: its previous frame has been waited for above
            mirhi::check(mirhi_buffer_write(frame_uniforms[slot], 0, uniform_bytes.data(), uniform_bytes.size()));
        frames.begin_frame();                                               // :393-397  reset_fence; :457-467 command_buffer.reset(), begin()
        record_commands(frames.current_frame().command_buffer, frames.image_index(), slot);
        const double t2 = timed ? now() : 0.0;
        frames.end_frame();                                                 // :555 command_buffer.end()
        const double t3 = timed ? now() : 0.0;
        frames.submit();                                                    // :407-424 queue_submit(..., in_flight_fence)
        frames.present();                                                   // :427-443
        last_image = frames.image_index();
        frames.next_frame();                                                // :446 current_frame = (current_frame + 1) % MAX_FRAMES_IN_FLIGHT
        frame_number++;
        if (timed) { const double t4 = now(); phase[0] += t1 - t0; phase[1] += t2 - t1; phase[2] += t3 - t2; phase[3] += t4 - t3; }
    }
};

extern "C" const char* mirhost_last_error_message(void) { return g_error.c_str(); }

extern "C" mirhi_result mirhost_frame_loop_create(mirhi_device* dev, const mirhost_frame_desc* desc, mirhost_frame_loop** out) {
    if (!dev || !desc || !out) return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: null argument");
    *out = nullptr;
    if (desc->frames_in_flight == 0 || desc->frames_in_flight > 16 || desc->image_count == 0 || !desc->images || (desc->draw_count && !desc->draws))
        return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: frame description needs 1..16 frames in flight, at least one image, and its draws");
    if (desc->per_frame_uniform > MIRHI_SLOT_COUNT || (desc->per_frame_uniform && (!desc->draw_count || !desc->draws[0].uniforms[desc->per_frame_uniform - 1].buffer)))
        return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: per_frame_uniform names a slot draw 0 does not bind");
    for (uint32_t i = 0; i < desc->image_count; i++) if (!desc->images[i]) return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: images[i] is null");
    for (uint32_t i = 0; i < desc->draw_count; i++)
        if (!desc->draws[i].pipeline || !desc->draws[i].vertex_buffer) return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: a draw needs a pipeline and a vertex buffer");
    try {
        *out = new mirhost_frame_loop(dev, *desc);
        if (desc->submit_thread) (*out)->device->set_submit_thread(true);
    }
    catch (const mirhi::RhiError& e) { return fail(e.code, e.what()); }
    catch (const std::exception& e) { return fail(MIRHI_ERR_ALLOCATOR, e.what()); }
    return MIRHI_OK;
}

extern "C" mirhi_result mirhost_frame_loop_run(mirhost_frame_loop* loop, uint64_t frames, double* seconds) {
    if (!loop) return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: loop is null");
    const auto t0 = std::chrono::steady_clock::now();
    try {
        for (uint64_t f = 0; f < frames; f++) loop->render_frame();
        loop->frames.wait_for_all_frames();                                 // frame_manager.rs wait_for_all_frames
    } catch (const mirhi::RhiError& e) { return fail(e.code, e.what()); }
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return MIRHI_OK;
}

extern "C" mirhi_result mirhost_frame_loop_last_image(const mirhost_frame_loop* loop, uint32_t* image_index, uint64_t* frames_rendered) {
    if (!loop) return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: loop is null");
    if (image_index) *image_index = loop->last_image;
    if (frames_rendered) *frames_rendered = loop->frame_number;
    return MIRHI_OK;
}

extern "C" mirhi_result mirhost_frame_loop_phase_seconds(mirhost_frame_loop* loop, int32_t enable, double* out4) {
    if (!loop) return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: loop is null");
    if (out4) for (int k = 0; k < 4; k++) out4[k] = loop->phase[k];
    for (double& p : loop->phase) p = 0.0;
    loop->timed = enable != 0;
    return MIRHI_OK;
}

extern "C" mirhi_result mirhost_frame_loop_destroy(mirhost_frame_loop* loop) {
    if (!loop) return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: loop is null");
    try { loop->frames.wait_for_all_frames(); } catch (...) {}
    if (loop->desc.submit_thread) { try { loop->device->set_submit_thread(false); } catch (...) {} }
    delete loop;
    return MIRHI_OK;
}
