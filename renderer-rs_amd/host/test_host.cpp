// test_host.cpp -- tests of the C++ host mirror (mirhi.hpp).  Without arguments: CPU-only checks that restate
// the reference's own unit tests (camera.rs:551-569, transform.rs:211-453, ubo.rs:421-596, vertex.rs:177-319,
// pipeline.rs:1177-1254, rendering.rs:1027-1073).  With --gpu: Renderer::render_frame / FrameManager on a MI355X.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "mirhi.hpp"
#include "gltf.hpp"

using namespace mirhi;

static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); failures++; } } while (0)
static bool near(float a, float b, float eps = 1e-5f) { return std::fabs(a - b) <= eps; }
static bool near3(Vec3 a, Vec3 b, float eps = 1e-5f) { return near(a.x, b.x, eps) && near(a.y, b.y, eps) && near(a.z, b.z, eps); }

static void cpu_tests() {
    // vertex.rs:177-319
    CHECK(sizeof(TriangleVertex) == 24 && offsetof(TriangleVertex, color) == 12);
    CHECK(sizeof(Vertex) == 48 && offsetof(Vertex, normal) == 12 && offsetof(Vertex, tex_coord) == 24 && offsetof(Vertex, tangent) == 32);
    CHECK(Vertex::binding_description().stride == 48 && TriangleVertex::attribute_descriptions()[1].offset == 12);
    CHECK(Vertex::attribute_descriptions()[3].offset == 32 && Vertex::attribute_descriptions()[2].location == 2);
    // camera.rs:551-569
    Camera cam;
    CHECK(near3(cam.view_matrix().transform_point3(Vec3::ZERO()), {0.0f, 0.0f, -5.0f}));
    CHECK(cam.projection_matrix().at(1, 1) < 0.0f);
    CHECK(near(cam.projection_matrix().at(0, 0), 1.35799513f, 1e-6f) && near(cam.projection_matrix().at(1, 1), -2.41421356f, 1e-6f));
    CHECK(near(cam.projection_matrix().at(2, 2), -1.00010001f, 1e-6f) && near(cam.projection_matrix().at(2, 3), -0.100010001f, 1e-7f));
    CHECK(cam.view_projection_matrix() == cam.projection_matrix() * cam.view_matrix());
    cam.set_rotation(0.0f, 3.14159265f * 0.5f);                 // yaw 90 deg: forward = -X
    CHECK(near3(cam.forward(), {-1.0f, 0.0f, 0.0f}, 1e-6f));
    cam.set_rotation(10.0f, 0.0f);                              // pitch clamps to 89 deg
    CHECK(cam.forward().y < 0.99999f && cam.forward().y > 0.999f);
    // ubo.rs:421-523
    CHECK(sizeof(CameraUbo) == 208 && offsetof(CameraUbo, projection) == 64 && offsetof(CameraUbo, view_projection) == 128 && offsetof(CameraUbo, camera_position) == 192);
    CHECK(sizeof(ObjectUbo) == 128 && offsetof(ObjectUbo, normal_matrix) == 64);
    const Mat4 view = Mat4::look_at_rh({0, 2, 5}, {0, 0, 0}, {0, 1, 0});
    const Mat4 proj = Mat4::perspective_rh(45.0f * 3.14159265f / 180.0f, 16.0f / 9.0f, 0.1f, 100.0f);
    CameraUbo cu(view, proj, {0, 2, 5});
    CHECK(cu.view_projection == proj * view);
    const Mat4 model = Mat4::from_scale_rotation_translation({2, 3, 4}, Quat::IDENTITY(), {1, 2, 3});
    ObjectUbo ou(model);
    CHECK(ou.normal_matrix == model.inverse().transpose());
    CHECK(near(ou.normal_matrix.at(0, 0), 0.5f) && near(ou.normal_matrix.at(1, 1), 1.0f / 3.0f) && near(ou.normal_matrix.at(2, 2), 0.25f));
    CHECK(ObjectUbo(Mat4::from_scale_rotation_translation({0, 1, 1}, Quat::IDENTITY(), {})).normal_matrix == Mat4::IDENTITY());
    CHECK(near(DirectionalLightUbo({0, -2, 0}, {1, 1, 1}, 1).direction.y, -1.0f) && DirectionalLightUbo({0, 0, 0}, {1, 1, 1}, 1).direction.length() == 0.0f);
    // transform.rs:129-146,231-267,315-442
    Transform parent; parent.with_position({10, 0, 0});
    Transform child; child.with_position({0, 5, 0}).with_parent(parent);
    CHECK(near3(child.world_matrix().transform_point3(Vec3::ZERO()), {10, 5, 0}, 1e-3f));
    CHECK(child.has_parent());
    child.clear_parent();
    CHECK(!child.has_parent() && child.world_matrix() == child.local_matrix());
    Transform t; t.with_scale({2, 2, 2}).with_rotation(Quat::from_axis_angle(Vec3::Y(), 3.14159265f * 0.5f)).with_position({1, 0, 0});
    CHECK(near3(t.world_matrix().transform_point3({1, 0, 0}), {1, 0, -2}, 1e-5f));
    Transform zero; zero.with_scale({0, 1, 1});
    CHECK(zero.normal_matrix() == Mat4::IDENTITY());
    // lights: Rust layout -> HLSL layout
    SpotLight sl;
    HlslSpotLight hs = to_hlsl(sl);
    CHECK(hs.inner_cone_cos == 0.9f && hs.outer_cone_cos == 0.8f && hs.intensity == 1.0f);
    // pipeline.rs:1177-1189,1230-1254 defaults; rendering.rs:1027-1073
    GraphicsPipelineBuilder b;
    CHECK(b.desc().topology == MIRHI_TOPOLOGY_TRIANGLE_LIST && b.desc().cull_mode == MIRHI_CULL_BACK && b.desc().front_face == MIRHI_FRONT_FACE_COUNTER_CLOCKWISE);
    CHECK(b.desc().depth_test_enable == 1 && b.desc().depth_write_enable == 1 && b.desc().depth_compare_op == MIRHI_COMPARE_LESS);
    CHECK(b.desc().blend_enable == 0 && b.desc().src_color_blend_factor == MIRHI_BLEND_ONE && b.desc().dst_color_blend_factor == MIRHI_BLEND_ZERO && b.desc().color_write_mask == 0xF);
    GraphicsPipelineBuilder ab; ab.color_blend_attachment(ColorBlendAttachment::alpha_blend());      // pipeline.rs:518-529
    CHECK(ab.desc().blend_enable == 1 && ab.desc().src_color_blend_factor == MIRHI_BLEND_SRC_ALPHA && ab.desc().dst_color_blend_factor == MIRHI_BLEND_ONE_MINUS_SRC_ALPHA &&
          ab.desc().src_alpha_blend_factor == MIRHI_BLEND_ONE && ab.desc().dst_alpha_blend_factor == MIRHI_BLEND_ZERO && ab.desc().alpha_blend_op == MIRHI_BLEND_OP_ADD);
    mirhi_rendering_info ri; mirhi_rendering_info_default(&ri);
    CHECK(ri.color_load_op == MIRHI_LOAD_OP_CLEAR && ri.color_store_op == MIRHI_STORE_OP_STORE && ri.clear_color[3] == 1.0f && ri.clear_color[0] == 0.0f);
    CHECK(ri.depth_load_op == MIRHI_LOAD_OP_CLEAR && ri.depth_store_op == MIRHI_STORE_OP_DONT_CARE && ri.clear_depth == 1.0f);
    CHECK(MAX_FRAMES_IN_FLIGHT == 2 && DEFAULT_DEPTH_FORMAT == Format::D32_SFLOAT);
    // model.rs defaults through the interleave helper
    Mesh mesh; mesh.positions = {{0, 0, 0}, {1, 0, 0}};
    auto vs = mesh.interleave();
    CHECK(vs.size() == 2 && vs[1].normal.y == 1.0f && vs[0].tangent.x == 1.0f && vs[0].tangent.w == 1.0f && vs[1].tex_coord.x == 0.0f);
    // Model::load (model.rs:111-270) on the dancer fixture: K5 counts and the loader's defaults
    if (const char* gltf = std::getenv("MIRHI_TEST_GLTF")) {
        const resources::Model m = resources::Model::load(gltf);
        CHECK(m.meshes.size() == 1 && m.total_vertices() == 11865 && m.total_triangles() == 17210);
        CHECK(m.materials.size() == 1 && m.meshes[0].material_index.has_value() && *m.meshes[0].material_index == 0);
        CHECK(near(m.aabb_min.x, -0.95f) && near(m.aabb_max.y, 0.769373f) && near(m.aabb_min.z, -0.92852f));
        for (auto& me : m.meshes) CHECK(me.positions.size() == me.normals.size() && me.positions.size() == me.tex_coords.size() && me.positions.size() == me.tangents.size());
        CHECK(m.aabb_min.x < m.aabb_max.x && m.aabb_min.y < m.aabb_max.y && m.aabb_min.z < m.aabb_max.z);
        uint32_t max_index = 0;
        for (uint32_t i : m.meshes[0].indices) max_index = i > max_index ? i : max_index;
        CHECK(max_index + 1 == m.meshes[0].positions.size());
        const auto vs0 = m.meshes[0].interleave();
        CHECK(vs0.size() == m.meshes[0].positions.size() && sizeof(vs0[0]) == 48);
        bool threw = false;
        try { resources::Model::load("/nonexistent/file.gltf"); } catch (const resources::ResourceError& e) { threw = std::strstr(e.what(), "File not found") != nullptr; }
        CHECK(threw);
        // images kept (ImagePolicy::Decode): the fixture ships the normal map only; the two absent files are reported, not fatal
        const resources::Model mt = resources::Model::load(gltf, resources::ImagePolicy::Decode);
        CHECK(m.images.empty() && m.material_textures.empty());                    // default = the reference's behaviour
        CHECK(mt.images.size() == 3 && !mt.images[0] && !mt.images[1] && mt.images[2].has_value() && mt.missing_images.size() == 2);
        CHECK(mt.images[2]->width == 1024 && mt.images[2]->height == 1024 && mt.images[2]->rgba.size() == 1024u * 1024u * 4u && mt.images[2]->source_channels == 3);
        uint64_t sum[4] = {0, 0, 0, 0};
        for (size_t i = 0; i < mt.images[2]->rgba.size(); i++) sum[i & 3] += mt.images[2]->rgba[i];
        CHECK(sum[3] == 255ull * 1024 * 1024 && sum[2] > sum[0] && sum[2] > sum[1]);   // opaque, and blue-dominant like any tangent-space normal map
        CHECK(mt.material_textures.size() == 1 && *mt.material_textures[0].normal == 2 && *mt.material_textures[0].base_color == 0 &&
              *mt.material_textures[0].metallic_roughness == 1 && !mt.material_textures[0].occlusion && mt.material_textures[0].double_sided);
    }
    // no GPU -> NoSuitableGpu, never a fallback
    int32_t n = 0; mirhi_device_count(&n);
    if (n == 0) {
        bool threw = false;
        try { Device::create(0); } catch (const RhiError& e) { threw = e.kind == RhiErrorKind::NoSuitableGpu; }
        CHECK(threw);
    }
}

static void gpu_tests() {
    Renderer r(256, 256);
    for (int i = 0; i < 5; i++) r.render_frame();             // > MAX_FRAMES_IN_FLIGHT: fences are waited and reused
    r.wait_idle();
    std::vector<uint8_t> px(256 * 256 * 4);
    r.last_image().read(px.data());
    int covered = 0;
    for (int i = 0; i < 256 * 256; i++) if (!(px[4 * i] == 108 && px[4 * i + 1] == 89 && px[4 * i + 2] == 89)) covered++;
    CHECK(covered == 8192);                                     // K2
    CHECK(px[0] == 108 && px[1] == 89 && px[2] == 89 && px[3] == 255);   // BGRA sRGB8 of (0.1, 0.1, 0.15, 1)
    const uint8_t* c = &px[4 * (149 * 256 + 128)];
    CHECK(std::abs((int)c[0] - 156) <= 2 && std::abs((int)c[1] - 156) <= 2 && std::abs((int)c[2] - 156) <= 2);
    r.resize(320, 200);
    r.render_frame();
    r.wait_idle();
    CHECK(r.width() == 320 && r.last_image().height() == 200);
    // reference error behaviour through the C++ types
    bool threw = false;
    try { Buffer bad(r.device(), BufferUsage::Vertex, 0); } catch (const RhiError& e) { threw = e.kind == RhiErrorKind::InvalidHandle && std::strstr(e.what(), "Buffer size must be greater than 0"); }
    CHECK(threw);
    threw = false;
    try { GraphicsPipelineBuilder().build(r.device()); } catch (const RhiError& e) { threw = e.kind == RhiErrorKind::PipelineError && std::strstr(e.what(), "Vertex shader is required"); }
    CHECK(threw);
    Fence f(r.device(), true);
    CHECK(f.is_signaled());
    f.reset();
    CHECK(!f.is_signaled());
}

int main(int argc, char** argv) {
    if (argc > 2 && std::strcmp(argv[1], "--load-gltf") == 0) {      // exit 0 = loaded, 3 = ResourceError (message on stdout)
        try {
            const resources::Model m = resources::Model::load(argv[2]);
            std::printf("loaded: %zu meshes, %zu vertices, %zu triangles\n", m.meshes.size(), (size_t)m.total_vertices(), (size_t)m.total_triangles());
            return 0;
        } catch (const resources::ResourceError& e) { std::printf("ResourceError: %s\n", e.what()); return 3; }
    }
    cpu_tests();
    if (argc > 1 && std::strcmp(argv[1], "--gpu") == 0) {
        try { gpu_tests(); } catch (const RhiError& e) { std::printf("FAIL: RhiError %s\n", e.what()); failures++; }
    }
    std::printf(failures ? "host tests: %d FAILED\n" : "host tests: ok\n", failures);
    return failures ? 1 : 0;
}
