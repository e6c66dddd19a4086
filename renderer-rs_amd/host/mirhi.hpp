// mirhi.hpp -- C++ host mirror of the reference's Rust API for the draw path, on top of the C ABI
// (include/mirhi.h).  Same type and method names, argument meaning and error behaviour as
//   crates/rhi        Device, BufferUsage, Buffer, TriangleVertex, Vertex, GraphicsPipelineBuilder, Pipeline,
//                     CommandPool, CommandBuffer, ColorAttachment, DepthAttachment, RenderingConfig,
//                     Semaphore, Fence, FrameSync, RhiError               (lib.rs:12-34)
//   crates/renderer   Renderer, FrameManager, DepthBuffer, MAX_FRAMES_IN_FLIGHT   (lib.rs:43)
//   crates/scene      Camera, Projection, Transform, DirectionalLight, PointLight, SpotLight
//   crates/resources  CameraUbo, ObjectUbo, DirectionalLightUbo, SceneUbo, Material, Mesh
// so that a test written against the reference reads the same here.  Rust is not available in this image;
// the 1:1 Rust binding a maintainer would write is shown in INTEGRATION.md.
#pragma once

#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mirhi.h"

namespace mirhi {

// ------------------------------------------------------------------------------------------------
// crates/rhi/src/error.rs:6-50
// ------------------------------------------------------------------------------------------------
enum class RhiErrorKind { VulkanError = 1, LoadingError, AllocatorError, NoSuitableGpu, ShaderError, SurfaceError,
                          SwapchainError, InvalidHandle, PipelineError, LockPoisoned, Timeout, NotReady };

class RhiError : public std::runtime_error {
public:
    RhiError(mirhi_result code, std::string msg) : std::runtime_error(msg), kind(static_cast<RhiErrorKind>(code)), code(code) {}
    RhiErrorKind kind;
    mirhi_result code;
};

inline void check(mirhi_result r) {
    if (r != MIRHI_OK) throw RhiError(r, mirhi_last_error_message());
}

constexpr size_t MAX_FRAMES_IN_FLIGHT = 2;   // crates/renderer/src/lib.rs:43, crates/rhi/src/sync.rs:314

// ------------------------------------------------------------------------------------------------
// glam 0.30.9 subset used by the scene / resources crates (column-major Mat4)
// ------------------------------------------------------------------------------------------------
struct Vec2 { float x = 0, y = 0; };
struct Vec3 {
    float x = 0, y = 0, z = 0;
    static constexpr Vec3 ZERO() { return {0, 0, 0}; }
    static constexpr Vec3 ONE() { return {1, 1, 1}; }
    static constexpr Vec3 X() { return {1, 0, 0}; }
    static constexpr Vec3 Y() { return {0, 1, 0}; }
    static constexpr Vec3 NEG_Z() { return {0, 0, -1}; }
    Vec3 operator+(Vec3 o) const { return {x + o.x, y + o.y, z + o.z}; }
    Vec3 operator-(Vec3 o) const { return {x - o.x, y - o.y, z - o.z}; }
    Vec3 operator*(float s) const { return {x * s, y * s, z * s}; }
    float dot(Vec3 o) const { return x * o.x + y * o.y + z * o.z; }
    Vec3 cross(Vec3 o) const { return {y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x}; }
    float length() const { return std::sqrt(dot(*this)); }
    float length_squared() const { return dot(*this); }
    Vec3 normalize() const { float l = 1.0f / length(); return {x * l, y * l, z * l}; }
    Vec3 normalize_or_zero() const { float l = length(); return l > 0.0f && std::isfinite(1.0f / l) ? *this * (1.0f / l) : Vec3{}; }
};
struct Vec4 { float x = 0, y = 0, z = 0, w = 0; };

struct Quat {
    float x = 0, y = 0, z = 0, w = 1;
    static Quat IDENTITY() { return {}; }
    static Quat from_axis_angle(Vec3 axis, float angle) {
        const float s = std::sin(angle * 0.5f), c = std::cos(angle * 0.5f);
        return {axis.x * s, axis.y * s, axis.z * s, c};
    }
    Quat operator*(Quat r) const {   // Hamilton product
        return {w * r.x + x * r.w + y * r.z - z * r.y, w * r.y - x * r.z + y * r.w + z * r.x,
                w * r.z + x * r.y - y * r.x + z * r.w, w * r.w - x * r.x - y * r.y - z * r.z};
    }
    // glam EulerRot::YXZ: yaw about Y, then pitch about X, then roll about Z (camera.rs:179)
    static Quat from_euler_yxz(float yaw, float pitch, float roll) {
        return from_axis_angle(Vec3::Y(), yaw) * from_axis_angle(Vec3::X(), pitch) * from_axis_angle({0, 0, 1}, roll);
    }
    Vec3 operator*(Vec3 v) const {   // glam Quat * Vec3 (scalar path)
        const Vec3 b{x, y, z};
        const float b2 = b.dot(b);
        return v * (w * w - b2) + b * (v.dot(b) * 2.0f) + b.cross(v) * (w * 2.0f);
    }
};

struct Mat4 {
    float m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};   // m[4*col + row]
    static Mat4 IDENTITY() { return {}; }
    float& at(int row, int col) { return m[4 * col + row]; }
    float at(int row, int col) const { return m[4 * col + row]; }
    Vec4 mul_vec4(Vec4 v) const {
        return {((m[0] * v.x + m[4] * v.y) + m[8] * v.z) + m[12] * v.w, ((m[1] * v.x + m[5] * v.y) + m[9] * v.z) + m[13] * v.w,
                ((m[2] * v.x + m[6] * v.y) + m[10] * v.z) + m[14] * v.w, ((m[3] * v.x + m[7] * v.y) + m[11] * v.z) + m[15] * v.w};
    }
    Vec3 transform_point3(Vec3 p) const { Vec4 r = mul_vec4({p.x, p.y, p.z, 1.0f}); return {r.x, r.y, r.z}; }
    Mat4 operator*(const Mat4& b) const {
        Mat4 r;
        for (int c = 0; c < 4; c++) {
            Vec4 o = mul_vec4({b.m[4 * c], b.m[4 * c + 1], b.m[4 * c + 2], b.m[4 * c + 3]});
            r.m[4 * c] = o.x; r.m[4 * c + 1] = o.y; r.m[4 * c + 2] = o.z; r.m[4 * c + 3] = o.w;
        }
        return r;
    }
    bool operator==(const Mat4& o) const { return std::memcmp(m, o.m, sizeof m) == 0; }
    static Mat4 perspective_rh(float fov_y, float aspect, float z_near, float z_far) {
        const float s = std::sin(0.5f * fov_y), c = std::cos(0.5f * fov_y);
        const float h = c / s, w = h / aspect, r = z_far / (z_near - z_far);
        Mat4 o; std::memset(o.m, 0, sizeof o.m);
        o.m[0] = w; o.m[5] = h; o.m[10] = r; o.m[11] = -1.0f; o.m[14] = r * z_near;
        return o;
    }
    static Mat4 orthographic_rh(float l, float r, float b, float t, float n, float f) {
        const float rw = 1.0f / (r - l), rh = 1.0f / (t - b), rd = 1.0f / (n - f);
        Mat4 o; std::memset(o.m, 0, sizeof o.m);
        o.m[0] = rw + rw; o.m[5] = rh + rh; o.m[10] = rd;
        o.m[12] = -(l + r) * rw; o.m[13] = -(t + b) * rh; o.m[14] = rd * n; o.m[15] = 1.0f;
        return o;
    }
    static Mat4 look_at_rh(Vec3 eye, Vec3 center, Vec3 up) {
        const Vec3 f = (center - eye).normalize(), s = f.cross(up).normalize(), u = s.cross(f);
        Mat4 o;
        o.m[0] = s.x; o.m[1] = u.x; o.m[2] = -f.x; o.m[3] = 0;
        o.m[4] = s.y; o.m[5] = u.y; o.m[6] = -f.y; o.m[7] = 0;
        o.m[8] = s.z; o.m[9] = u.z; o.m[10] = -f.z; o.m[11] = 0;
        o.m[12] = -eye.dot(s); o.m[13] = -eye.dot(u); o.m[14] = eye.dot(f); o.m[15] = 1;
        return o;
    }
    static Mat4 from_scale_rotation_translation(Vec3 s, Quat q, Vec3 t) {
        const float x2 = q.x + q.x, y2 = q.y + q.y, z2 = q.z + q.z;
        const float xx = q.x * x2, xy = q.x * y2, xz = q.x * z2, yy = q.y * y2, yz = q.y * z2, zz = q.z * z2;
        const float wx = q.w * x2, wy = q.w * y2, wz = q.w * z2;
        Mat4 o;
        o.m[0] = (1 - (yy + zz)) * s.x; o.m[1] = (xy + wz) * s.x; o.m[2] = (xz - wy) * s.x; o.m[3] = 0;
        o.m[4] = (xy - wz) * s.y; o.m[5] = (1 - (xx + zz)) * s.y; o.m[6] = (yz + wx) * s.y; o.m[7] = 0;
        o.m[8] = (xz + wy) * s.z; o.m[9] = (yz - wx) * s.z; o.m[10] = (1 - (xx + yy)) * s.z; o.m[11] = 0;
        o.m[12] = t.x; o.m[13] = t.y; o.m[14] = t.z; o.m[15] = 1;
        return o;
    }
    Mat4 transpose() const { Mat4 r; for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r.m[4 * i + j] = m[4 * j + i]; return r; }
    float minor3(int r0, int r1, int r2, int c0, int c1, int c2) const {
        return at(r0, c0) * (at(r1, c1) * at(r2, c2) - at(r2, c1) * at(r1, c2)) - at(r0, c1) * (at(r1, c0) * at(r2, c2) - at(r2, c0) * at(r1, c2)) +
               at(r0, c2) * (at(r1, c0) * at(r2, c1) - at(r2, c0) * at(r1, c1));
    }
    float determinant() const {
        return at(0, 0) * minor3(1, 2, 3, 1, 2, 3) - at(0, 1) * minor3(1, 2, 3, 0, 2, 3) + at(0, 2) * minor3(1, 2, 3, 0, 1, 3) -
               at(0, 3) * minor3(1, 2, 3, 0, 1, 2);
    }
    Mat4 inverse() const {
        float cof[16];
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) {
                int rr[3], cc[3], a = 0, b = 0;
                for (int i = 0; i < 4; i++) { if (i != r) rr[a++] = i; if (i != c) cc[b++] = i; }
                const float mn = minor3(rr[0], rr[1], rr[2], cc[0], cc[1], cc[2]);
                cof[4 * c + r] = ((r + c) & 1) ? -mn : mn;
            }
        const float inv_det = 1.0f / (at(0, 0) * cof[0] + at(0, 1) * cof[4] + at(0, 2) * cof[8] + at(0, 3) * cof[12]);
        Mat4 o;
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) o.m[4 * c + r] = cof[4 * r + c] * inv_det;
        return o;
    }
};

// ------------------------------------------------------------------------------------------------
// crates/rhi/src/vertex.rs
// ------------------------------------------------------------------------------------------------
struct VertexInputBindingDescription { uint32_t binding, stride; };
struct VertexInputAttributeDescription { uint32_t binding, location, offset; };

struct TriangleVertex {   // vertex.rs:20-61
    Vec3 position, color;
    static VertexInputBindingDescription binding_description() { return {0, (uint32_t)sizeof(TriangleVertex)}; }
    static std::array<VertexInputAttributeDescription, 2> attribute_descriptions() { return {{{0, 0, 0}, {0, 1, 12}}}; }
};
static_assert(sizeof(TriangleVertex) == 24, "vertex.rs:177-190");

struct Vertex {           // vertex.rs:88-170
    Vec3 position, normal;
    Vec2 tex_coord;
    Vec4 tangent;
    static constexpr size_t size() { return 48; }
    static VertexInputBindingDescription binding_description() { return {0, 48}; }
    static std::array<VertexInputAttributeDescription, 4> attribute_descriptions() { return {{{0, 0, 0}, {0, 1, 12}, {0, 2, 24}, {0, 3, 32}}}; }
};
static_assert(sizeof(Vertex) == 48, "vertex.rs:230-245");

// ------------------------------------------------------------------------------------------------
// crates/rhi: Device, Buffer, Image (DepthBuffer / colour target)
// ------------------------------------------------------------------------------------------------
class Device {   // device.rs:61-77; shared as Arc<Device>
public:
    static std::shared_ptr<Device> create(int ordinal = 0) {
        // the structs of include/mirhi.h this translation unit was compiled against are one ABI's
        if (mirhi_abi_version() != MIRHI_ABI_VERSION)
            throw RhiError(MIRHI_ERR_LOADING, "Loading error: libmirhi.so has ABI " + std::to_string(mirhi_abi_version()) + ", built against ABI " + std::to_string(MIRHI_ABI_VERSION));
        mirhi_device* h = nullptr;
        check(mirhi_device_create(ordinal, &h));
        return std::shared_ptr<Device>(new Device(h));
    }
    // a device some other owner created and will destroy (the frame loop of include/mirhost.h runs on its caller's device)
    static std::shared_ptr<Device> borrow(mirhi_device* h) { auto d = std::shared_ptr<Device>(new Device(h)); d->owned_ = false; return d; }
    ~Device() { if (h_ && owned_) mirhi_device_destroy(h_); }
    mirhi_device* handle() const { return h_; }
    void wait_idle() const { check(mirhi_device_wait_idle(h_)); }   // device.rs:290-293
    void set_queue_lanes(uint32_t lanes) const { check(mirhi_device_set_queue_lanes(h_, lanes)); }
    void set_submit_thread(bool enable) const { check(mirhi_device_set_submit_thread(h_, enable ? 1u : 0u)); }   // vkQueueSubmit returns at once (include/mirhi.h)
private:
    explicit Device(mirhi_device* h) : h_(h) {}
    mirhi_device* h_;
    bool owned_ = true;
};

enum class BufferUsage { Vertex = 0, Index, Uniform, Storage, Staging, Indirect };   // buffer.rs:47-60
inline const char* buffer_usage_name(BufferUsage u) {                                 // buffer.rs:102-111
    static const char* n[] = {"vertex", "index", "uniform", "storage", "staging", "indirect"};
    return n[(int)u];
}

class Buffer {   // buffer.rs:124-436
public:
    Buffer(std::shared_ptr<Device> device, BufferUsage usage, uint64_t size) : device_(std::move(device)) {
        check(mirhi_buffer_create(device_->handle(), (mirhi_buffer_usage)usage, size, &h_));
    }
    static Buffer new_with_data(std::shared_ptr<Device> device, BufferUsage usage, const void* data, uint64_t len) {
        Buffer b(std::move(device), usage, len);
        b.write_data(0, data, len);
        return b;
    }
    Buffer(Buffer&& o) noexcept : device_(std::move(o.device_)), h_(o.h_) { o.h_ = nullptr; }
    Buffer(const Buffer&) = delete;
    ~Buffer() { if (h_) mirhi_buffer_destroy(h_); }
    void write_data(uint64_t offset, const void* data, uint64_t len) const { check(mirhi_buffer_write(h_, offset, data, len)); }
    void upload(const void* data, uint64_t len) const { write_data(0, data, len); }
    void upload_via_staging(const void* data, uint64_t len) const { check(mirhi_buffer_upload_via_staging(h_, data, len)); }
    mirhi_buffer* handle() const { return h_; }
    uint64_t size() const { return mirhi_buffer_size(h_); }
    BufferUsage usage() const { return (BufferUsage)mirhi_buffer_usage_of(h_); }
private:
    std::shared_ptr<Device> device_;
    mirhi_buffer* h_ = nullptr;
};

enum class Format { Undefined = 0, B8G8R8A8_SRGB, R32G32B32A32_SFLOAT, D32_SFLOAT, R8G8B8A8_UNORM, R32_UINT, R8G8B8A8_SRGB };

class Image {
public:
    Image(std::shared_ptr<Device> device, uint32_t w, uint32_t h, Format f) : device_(std::move(device)), w_(w), h_px_(h), f_(f) {
        check(mirhi_image_create(device_->handle(), w, h, (mirhi_format)f, &h_));
    }
    Image(Image&& o) noexcept : device_(std::move(o.device_)), h_(o.h_), w_(o.w_), h_px_(o.h_px_), f_(o.f_) { o.h_ = nullptr; }
    Image(const Image&) = delete;
    ~Image() { if (h_) mirhi_image_destroy(h_); }
    mirhi_image* handle() const { return h_; }
    uint32_t width() const { return w_; }
    uint32_t height() const { return h_px_; }
    Format format() const { return f_; }
    void read(void* dst) const { check(mirhi_image_read(h_, dst, mirhi_image_size_bytes(h_))); }
    void upload(const void* src) const { check(mirhi_image_upload(h_, src, mirhi_image_size_bytes(h_))); }
    void generate_mips() const { check(mirhi_image_generate_mips(h_)); }     // texture fidelity
    uint32_t mip_levels() const { return mirhi_image_mip_levels(h_); }
    void set_max_anisotropy(uint32_t n) { check(mirhi_image_set_max_anisotropy(h_, n)); }          // sampler state (device.rs:161-165)
    uint32_t max_anisotropy() const { return mirhi_image_max_anisotropy(h_); }
private:
    std::shared_ptr<Device> device_;
    mirhi_image* h_ = nullptr;
    uint32_t w_, h_px_;
    Format f_;
};

constexpr Format DEFAULT_DEPTH_FORMAT = Format::D32_SFLOAT;   // depth_buffer.rs:48
class DepthBuffer {   // crates/renderer/src/depth_buffer.rs:67-280
public:
    DepthBuffer(std::shared_ptr<Device> device, uint32_t w, uint32_t h, Format f) : image_(std::move(device), w, h, f) {}
    static DepthBuffer with_default_format(std::shared_ptr<Device> device, uint32_t w, uint32_t h) { return DepthBuffer(std::move(device), w, h, DEFAULT_DEPTH_FORMAT); }
    const Image& image() const { return image_; }
    Format format() const { return image_.format(); }
    uint32_t width() const { return image_.width(); }
    uint32_t height() const { return image_.height(); }
private:
    Image image_;
};

// ------------------------------------------------------------------------------------------------
// crates/rhi/src/pipeline.rs
// ------------------------------------------------------------------------------------------------
enum class ShaderProgram { None = -1, Triangle = 0, Model = 1, ModelFull = 2, ModelPbr = 3 };   // replaces Shader::from_spirv_file
enum class PrimitiveTopology { PointList = 0, LineList, LineStrip, TriangleList, TriangleStrip, TriangleFan };
enum class PolygonMode { Fill = 0, Line, Point };
enum class CullMode { None = 0, Front, Back, FrontAndBack };
enum class FrontFace { CounterClockwise = 0, Clockwise };
enum class CompareOp { Never = 0, Less, Equal, LessOrEqual, Greater, NotEqual, GreaterOrEqual, Always };
enum class BlendFactor { Zero = 0, One, SrcColor, OneMinusSrcColor, DstColor, OneMinusDstColor, SrcAlpha, OneMinusSrcAlpha, DstAlpha, OneMinusDstAlpha,
                         ConstantColor, OneMinusConstantColor, ConstantAlpha, OneMinusConstantAlpha, SrcAlphaSaturate };   // pipeline.rs:411-448
enum class BlendOp { Add = 0, Subtract, ReverseSubtract, Min, Max };                                                           // pipeline.rs:452-476
struct ColorBlendAttachment {   // pipeline.rs:478-531
    bool blend_enable = false;
    BlendFactor src_color_blend_factor = BlendFactor::One, dst_color_blend_factor = BlendFactor::Zero; BlendOp color_blend_op = BlendOp::Add;
    BlendFactor src_alpha_blend_factor = BlendFactor::One, dst_alpha_blend_factor = BlendFactor::Zero; BlendOp alpha_blend_op = BlendOp::Add;
    uint32_t color_write_mask = 0xF;
    static ColorBlendAttachment alpha_blend() {      // :518-529  src * src_alpha + dst * (1 - src_alpha)
        ColorBlendAttachment a; a.blend_enable = true; a.src_color_blend_factor = BlendFactor::SrcAlpha; a.dst_color_blend_factor = BlendFactor::OneMinusSrcAlpha;
        return a;
    }
};

class Pipeline {
public:
    Pipeline(std::shared_ptr<Device> d, mirhi_pipeline* h) : device_(std::move(d)), h_(h) {}
    Pipeline(Pipeline&& o) noexcept : device_(std::move(o.device_)), h_(o.h_) { o.h_ = nullptr; }
    Pipeline(const Pipeline&) = delete;
    ~Pipeline() { if (h_) mirhi_pipeline_destroy(h_); }
    mirhi_pipeline* handle() const { return h_; }
private:
    std::shared_ptr<Device> device_;
    mirhi_pipeline* h_;
};

class GraphicsPipelineBuilder {   // pipeline.rs:590-1059
public:
    GraphicsPipelineBuilder() { mirhi_pipeline_desc_default(&d_); }
    GraphicsPipelineBuilder& vertex_shader(ShaderProgram p) { d_.vertex_program = (int32_t)p; return *this; }
    GraphicsPipelineBuilder& fragment_shader(ShaderProgram p) { d_.fragment_program = (int32_t)p; return *this; }
    GraphicsPipelineBuilder& vertex_binding(VertexInputBindingDescription b) { d_.vertex_stride = b.stride; return *this; }
    template <size_t N> GraphicsPipelineBuilder& vertex_attributes(const std::array<VertexInputAttributeDescription, N>& a) {
        d_.attribute_count = (uint32_t)N;
        for (size_t i = 0; i < N && i < 4; i++) d_.attribute_offsets[i] = a[i].offset;
        return *this;
    }
    GraphicsPipelineBuilder& topology(PrimitiveTopology t) { d_.topology = (int32_t)t; return *this; }
    GraphicsPipelineBuilder& polygon_mode(PolygonMode m) { d_.polygon_mode = (int32_t)m; return *this; }
    GraphicsPipelineBuilder& cull_mode(CullMode m) { d_.cull_mode = (int32_t)m; return *this; }
    GraphicsPipelineBuilder& front_face(FrontFace f) { d_.front_face = (int32_t)f; return *this; }
    GraphicsPipelineBuilder& depth_test_enable(bool e) { d_.depth_test_enable = e; return *this; }
    GraphicsPipelineBuilder& depth_write_enable(bool e) { d_.depth_write_enable = e; return *this; }
    GraphicsPipelineBuilder& depth_compare_op(CompareOp op) { d_.depth_compare_op = (int32_t)op; return *this; }
    GraphicsPipelineBuilder& fragment_discard_enable(bool e) { d_.fragment_discard_enable = e; return *this; }   // alpha-masked MODEL_PBR materials (model_pbr.hlsl:176-179)
    GraphicsPipelineBuilder& color_blend_attachment(const ColorBlendAttachment& a) {      // pipeline.rs color_blend_attachments
        d_.blend_enable = a.blend_enable ? 1u : 0u;
        d_.src_color_blend_factor = (int32_t)a.src_color_blend_factor; d_.dst_color_blend_factor = (int32_t)a.dst_color_blend_factor; d_.color_blend_op = (int32_t)a.color_blend_op;
        d_.src_alpha_blend_factor = (int32_t)a.src_alpha_blend_factor; d_.dst_alpha_blend_factor = (int32_t)a.dst_alpha_blend_factor; d_.alpha_blend_op = (int32_t)a.alpha_blend_op;
        d_.color_write_mask = a.color_write_mask;
        return *this;
    }
    GraphicsPipelineBuilder& color_attachment_format(Format f) { d_.color_attachment_formats[d_.color_attachment_count++ & 3] = (int32_t)f; return *this; }
    GraphicsPipelineBuilder& depth_attachment_format(Format f) { d_.depth_attachment_format = (int32_t)f; return *this; }
    const mirhi_pipeline_desc& desc() const { return d_; }
    Pipeline build(std::shared_ptr<Device> device) const {
        mirhi_pipeline* h = nullptr;
        check(mirhi_pipeline_create(device->handle(), &d_, &h));
        return Pipeline(std::move(device), h);
    }
private:
    mirhi_pipeline_desc d_;
};

// ------------------------------------------------------------------------------------------------
// crates/rhi/src/rendering.rs: ColorAttachment, DepthAttachment, RenderingConfig
// ------------------------------------------------------------------------------------------------
enum class AttachmentLoadOp { Load = 0, Clear, DontCare };
enum class AttachmentStoreOp { Store = 0, DontCare };

struct ColorAttachment {   // rendering.rs:65-115
    const Image* image;
    AttachmentLoadOp load_op = AttachmentLoadOp::Clear;
    AttachmentStoreOp store_op = AttachmentStoreOp::Store;
    std::array<float, 4> clear_color{0.0f, 0.0f, 0.0f, 1.0f};
    explicit ColorAttachment(const Image& img) : image(&img) {}
    ColorAttachment& with_clear_color(std::array<float, 4> c) { clear_color = c; return *this; }
    ColorAttachment& with_load_op(AttachmentLoadOp op) { load_op = op; return *this; }
    ColorAttachment& load() { load_op = AttachmentLoadOp::Load; return *this; }
};
struct DepthAttachment {   // rendering.rs:319-370
    const Image* image;
    AttachmentLoadOp load_op = AttachmentLoadOp::Clear;
    AttachmentStoreOp store_op = AttachmentStoreOp::DontCare;
    float clear_depth = 1.0f;
    explicit DepthAttachment(const Image& img) : image(&img) {}
    DepthAttachment& with_clear_depth(float d) { clear_depth = d; return *this; }
    DepthAttachment& load() { load_op = AttachmentLoadOp::Load; return *this; }
    DepthAttachment& store() { store_op = AttachmentStoreOp::Store; return *this; }
};
class RenderingConfig {    // rendering.rs:680-726
public:
    RenderingConfig(uint32_t w, uint32_t h) : w_(w), h_(h) { mirhi_rendering_info_default(&info_); }
    RenderingConfig& with_color_attachment(const ColorAttachment& a) {
        info_.color_image = a.image->handle(); info_.color_load_op = (int32_t)a.load_op; info_.color_store_op = (int32_t)a.store_op;
        std::memcpy(info_.clear_color, a.clear_color.data(), 16);
        return *this;
    }
    RenderingConfig& with_depth_attachment(const DepthAttachment& a) {
        info_.depth_image = a.image->handle(); info_.depth_load_op = (int32_t)a.load_op; info_.depth_store_op = (int32_t)a.store_op;
        info_.clear_depth = a.clear_depth;
        return *this;
    }
    uint32_t width() const { return w_; }
    uint32_t height() const { return h_; }
    const mirhi_rendering_info& build() const { return info_; }
private:
    uint32_t w_, h_;
    mirhi_rendering_info info_;
};

// ------------------------------------------------------------------------------------------------
// crates/rhi/src/command.rs, sync.rs
// ------------------------------------------------------------------------------------------------
class CommandPool {   // command.rs:52-237 (HIP needs no pool object; kept for call-site parity)
public:
    CommandPool(std::shared_ptr<Device> device, uint32_t queue_family_index) : device_(std::move(device)), family_(queue_family_index) {}
    const std::shared_ptr<Device>& device() const { return device_; }
    uint32_t queue_family_index() const { return family_; }
private:
    std::shared_ptr<Device> device_;
    uint32_t family_;
};

enum class IndexType { Uint16 = 0, Uint32 = 1 };
struct Viewport { float x, y, width, height, min_depth, max_depth; };
struct Rect2D { int32_t x, y; uint32_t width, height; };

class CommandBuffer {   // command.rs:279-628
public:
    CommandBuffer(std::shared_ptr<Device> device, const CommandPool&) : device_(std::move(device)) { check(mirhi_cmd_create(device_->handle(), &h_)); }
    CommandBuffer(CommandBuffer&& o) noexcept : device_(std::move(o.device_)), h_(o.h_) { o.h_ = nullptr; }
    CommandBuffer(const CommandBuffer&) = delete;
    ~CommandBuffer() { if (h_) mirhi_cmd_destroy(h_); }
    mirhi_cmd* handle() const { return h_; }
    void begin() const { check(mirhi_cmd_begin(h_)); }
    void begin_reusable() const { check(mirhi_cmd_begin_reusable(h_)); }
    void end() const { check(mirhi_cmd_end(h_)); }
    void reset() const { check(mirhi_cmd_reset(h_)); }
    void begin_rendering(const mirhi_rendering_info& info) const { check(mirhi_cmd_begin_rendering(h_, &info)); }
    void end_rendering() const { check(mirhi_cmd_end_rendering(h_)); }
    void bind_pipeline(const Pipeline& p) const { check(mirhi_cmd_bind_pipeline(h_, p.handle())); }
    void bind_vertex_buffers(uint32_t first_binding, const Buffer& b, uint64_t offset) const {
        mirhi_buffer* bufs[1] = {b.handle()}; uint64_t offs[1] = {offset};
        check(mirhi_cmd_bind_vertex_buffers(h_, first_binding, 1, bufs, offs));
    }
    void bind_index_buffer(const Buffer& b, uint64_t offset, IndexType t) const { check(mirhi_cmd_bind_index_buffer(h_, b.handle(), offset, (mirhi_index_type)t)); }
    void bind_uniform(mirhi_uniform_slot slot, const Buffer& b, uint64_t offset = 0, uint64_t range = 0) const { check(mirhi_cmd_bind_uniform(h_, slot, b.handle(), offset, range)); }
    void bind_texture(mirhi_texture_slot slot, const Image* img) const { check(mirhi_cmd_bind_texture(h_, slot, img ? img->handle() : nullptr)); }
    void set_viewport(const Viewport& v) const { mirhi_viewport vp{v.x, v.y, v.width, v.height, v.min_depth, v.max_depth}; check(mirhi_cmd_set_viewport(h_, &vp)); }
    void set_scissor(const Rect2D& r) const { mirhi_rect2d sc{r.x, r.y, r.width, r.height}; check(mirhi_cmd_set_scissor(h_, &sc)); }
    void draw(uint32_t vertex_count, uint32_t instance_count, uint32_t first_vertex, uint32_t first_instance) const { check(mirhi_cmd_draw(h_, vertex_count, instance_count, first_vertex, first_instance)); }
    void draw_indexed(uint32_t index_count, uint32_t instance_count, uint32_t first_index, int32_t vertex_offset, uint32_t first_instance) const {
        check(mirhi_cmd_draw_indexed(h_, index_count, instance_count, first_index, vertex_offset, first_instance));
    }
private:
    std::shared_ptr<Device> device_;
    mirhi_cmd* h_ = nullptr;
};

class Semaphore {   // sync.rs:62-111: GPU-GPU ordering collapses to HIP stream order
public:
    explicit Semaphore(std::shared_ptr<Device> device) : device_(std::move(device)) {}
private:
    std::shared_ptr<Device> device_;
};

class Fence {       // sync.rs:134-298
public:
    Fence(std::shared_ptr<Device> device, bool signaled) : device_(std::move(device)) { check(mirhi_fence_create(device_->handle(), signaled, &h_)); }
    Fence(Fence&& o) noexcept : device_(std::move(o.device_)), h_(o.h_) { o.h_ = nullptr; }
    Fence(const Fence&) = delete;
    ~Fence() { if (h_) mirhi_fence_destroy(h_); }
    mirhi_fence* handle() const { return h_; }
    void wait(uint64_t timeout) const { check(mirhi_fence_wait(h_, timeout)); }
    void reset() const { check(mirhi_fence_reset(h_)); }
    bool is_signaled() const { return mirhi_fence_status(h_) == MIRHI_OK; }
private:
    std::shared_ptr<Device> device_;
    mirhi_fence* h_ = nullptr;
};

struct FrameSync {  // sync.rs:366-460
    Semaphore image_available, render_finished;
    Fence in_flight;
    explicit FrameSync(std::shared_ptr<Device> d) : image_available(d), render_finished(d), in_flight(d, true) {}
};

// ------------------------------------------------------------------------------------------------
// crates/scene: camera.rs, transform.rs, light.rs
// ------------------------------------------------------------------------------------------------
struct Projection {
    enum Kind { Perspective, Orthographic } kind = Perspective;
    float fov_y = 45.0f * 3.14159265358979323846f / 180.0f, aspect = 16.0f / 9.0f, near_ = 0.1f, far_ = 1000.0f;   // camera.rs:43-56
    float left = -1, right = 1, bottom = -1, top = 1;
};

struct Camera {   // camera.rs:110-190
    Vec3 position{0.0f, 0.0f, 5.0f};
    Quat rotation = Quat::IDENTITY();
    Projection projection;
    Mat4 view_matrix() const { const Vec3 forward = rotation * Vec3::NEG_Z(); return Mat4::look_at_rh(position, position + forward, Vec3::Y()); }
    Mat4 projection_matrix() const {
        Mat4 proj = projection.kind == Projection::Perspective
                        ? Mat4::perspective_rh(projection.fov_y, projection.aspect, projection.near_, projection.far_)
                        : Mat4::orthographic_rh(projection.left, projection.right, projection.bottom, projection.top, projection.near_, projection.far_);
        proj.m[5] *= -1.0f;   // camera.rs:135: flip Y for Vulkan
        return proj;
    }
    Mat4 view_projection_matrix() const { return projection_matrix() * view_matrix(); }
    Vec3 forward() const { return rotation * Vec3::NEG_Z(); }
    Vec3 right() const { return rotation * Vec3::X(); }
    Vec3 up() const { return rotation * Vec3::Y(); }
    void set_rotation(float pitch, float yaw) {   // camera.rs:173-180
        const float max_pitch = 89.0f * 3.14159265358979323846f / 180.0f;
        const float p = pitch < -max_pitch ? -max_pitch : (pitch > max_pitch ? max_pitch : pitch);
        rotation = Quat::from_euler_yxz(yaw, p, 0.0f);
    }
    void translate(Vec3 offset) { position = position + offset; }
    void move_forward(float d) { position = position + forward() * d; }
    void move_right(float d) { position = position + right() * d; }
    void move_up(float d) { position = position + up() * d; }
};

struct Transform {   // transform.rs:119-179
    Vec3 position{}, scale{1, 1, 1};
    Quat rotation = Quat::IDENTITY();
    std::shared_ptr<Transform> parent;
    Transform& with_position(Vec3 p) { position = p; return *this; }
    Transform& with_rotation(Quat q) { rotation = q; return *this; }
    Transform& with_scale(Vec3 s) { scale = s; return *this; }
    Transform& with_parent(const Transform& p) { parent = std::make_shared<Transform>(p); return *this; }
    void clear_parent() { parent.reset(); }
    bool has_parent() const { return (bool)parent; }
    Mat4 local_matrix() const { return Mat4::from_scale_rotation_translation(scale, rotation, position); }
    Mat4 world_matrix() const { const Mat4 local = local_matrix(); return parent ? parent->world_matrix() * local : local; }
    Mat4 normal_matrix() const {
        const Mat4 model = world_matrix();
        return std::fabs(model.determinant()) < 1e-6f ? Mat4::IDENTITY() : model.inverse().transpose();
    }
};

// Rust-side light layouts (light.rs:7-74) ...
struct DirectionalLight { Vec3 direction{0, -1, 0}; float _pad0 = 0; Vec3 color{1, 1, 1}; float intensity = 1; };
struct PointLight { Vec3 position{}; float radius = 10; Vec3 color{1, 1, 1}; float intensity = 1; };
struct SpotLight { Vec3 position{}; float _pad0 = 0; Vec3 direction{0, -1, 0}; float _pad1 = 0; Vec3 color{1, 1, 1}; float intensity = 1;
                   float inner_cutoff = 0.9f, outer_cutoff = 0.8f; float _pad2[2] = {0, 0}; };
static_assert(sizeof(DirectionalLight) == 32 && sizeof(PointLight) == 32 && sizeof(SpotLight) == 64, "light.rs layouts");
// ... and the HLSL layouts the GPU programs read (lights.hlsli:17-55).  The two disagree in the reference
// (SURVEY 0.7); the HLSL layout is GPU truth and the conversion happens here, at the boundary.
struct HlslDirectionalLight { Vec3 direction; float intensity; Vec3 color; float padding; };
struct HlslSpotLight { Vec3 position; float inner_cone_cos; Vec3 direction; float outer_cone_cos; Vec3 color; float intensity; };
struct HlslLightUbo { HlslDirectionalLight directional; uint32_t num_point_lights, num_spot_lights; float padding[2]; };
static_assert(sizeof(HlslDirectionalLight) == 32 && sizeof(HlslSpotLight) == 48 && sizeof(HlslLightUbo) == 48, "lights.hlsli layouts");
inline HlslDirectionalLight to_hlsl(const DirectionalLight& l) { return {l.direction, l.intensity, l.color, 0.0f}; }
inline HlslSpotLight to_hlsl(const SpotLight& l) { return {l.position, l.inner_cutoff, l.direction, l.outer_cutoff, l.color, l.intensity}; }

// ------------------------------------------------------------------------------------------------
// crates/resources: ubo.rs, material.rs, model.rs (Mesh SoA + the interleave helper the reference lacks)
// ------------------------------------------------------------------------------------------------
struct CameraUbo {   // ubo.rs:64-117
    Mat4 view, projection, view_projection;
    Vec3 camera_position;
    float _padding = 0;
    CameraUbo() = default;
    CameraUbo(const Mat4& v, const Mat4& p, Vec3 eye) : view(v), projection(p), view_projection(p * v), camera_position(eye) {}
    void update_view(const Mat4& v) { view = v; view_projection = projection * view; }
    void update_projection(const Mat4& p) { projection = p; view_projection = projection * view; }
};
static_assert(sizeof(CameraUbo) == 208, "ubo.rs:421-434");
struct ObjectUbo {   // ubo.rs:174-259
    Mat4 model, normal_matrix;
    ObjectUbo() = default;
    explicit ObjectUbo(const Mat4& m) : model(m), normal_matrix(compute_normal_matrix(m)) {}
    void update_model(const Mat4& m) { model = m; normal_matrix = compute_normal_matrix(m); }
    static Mat4 compute_normal_matrix(const Mat4& m) { return std::fabs(m.determinant()) < 1e-6f ? Mat4::IDENTITY() : m.inverse().transpose(); }
};
static_assert(sizeof(ObjectUbo) == 128, "ubo.rs:466-477");
struct DirectionalLightUbo {   // ubo.rs:287-330
    Vec3 direction; float _padding1 = 0; Vec3 color; float intensity = 0;
    DirectionalLightUbo() = default;
    DirectionalLightUbo(Vec3 d, Vec3 c, float i) : direction(d.normalize_or_zero()), color(c), intensity(i) {}
    static constexpr size_t size() { return 32; }
};
struct SceneUbo { Vec3 ambient_color; float time = 0, delta_time = 0; float _padding[3] = {0, 0, 0}; };   // ubo.rs:355-395
static_assert(sizeof(DirectionalLightUbo) == 32 && sizeof(SceneUbo) == 32, "ubo.rs sizes");
struct Material { Vec4 base_color{1, 1, 1, 1}; float metallic = 0, roughness = 0.5f, ao = 1; Vec4 emissive{}; };   // material.rs:6-30
struct MaterialData { Vec4 base_color; float metallic, roughness, ambient_occlusion, padding; };                      // model_full.hlsl:34-41
inline MaterialData to_hlsl(const Material& m) { return {m.base_color, m.metallic, m.roughness, m.ao, 0.0f}; }

struct Mesh {   // model.rs:31-44 (SoA as the glTF loader produces it)
    std::vector<Vec3> positions, normals;
    std::vector<Vec2> tex_coords;
    std::vector<Vec4> tangents;
    std::vector<uint32_t> indices;
    // model.rs:111-270 defaults for missing attributes: normal +Y, uv 0, tangent (1,0,0,1)
    std::vector<Vertex> interleave() const {
        std::vector<Vertex> out(positions.size());
        for (size_t i = 0; i < positions.size(); i++) {
            out[i].position = positions[i];
            out[i].normal = i < normals.size() ? normals[i] : Vec3{0, 1, 0};
            out[i].tex_coord = i < tex_coords.size() ? tex_coords[i] : Vec2{};
            out[i].tangent = i < tangents.size() ? tangents[i] : Vec4{1, 0, 0, 1};
        }
        return out;
    }
};

// ------------------------------------------------------------------------------------------------
// crates/renderer: FrameManager (frame_manager.rs:111-601) and Renderer (renderer.rs:55-683), offscreen
// ------------------------------------------------------------------------------------------------
struct FrameData {
    CommandBuffer command_buffer;
    Semaphore image_available_semaphore, render_finished_semaphore;
    Fence in_flight_fence;
    FrameData(std::shared_ptr<Device> d, const CommandPool& pool)
        : command_buffer(d, pool), image_available_semaphore(d), render_finished_semaphore(d), in_flight_fence(d, true) {}
};

class FrameManager {
public:
    // frames_in_flight: the reference's constant (lib.rs:43) unless a measurement asks for another depth
    FrameManager(std::shared_ptr<Device> device, const CommandPool& pool, size_t frames_in_flight = MAX_FRAMES_IN_FLIGHT) : device_(device) {
        frames_.reserve(frames_in_flight ? frames_in_flight : 1);
        for (size_t i = 0; i < (frames_in_flight ? frames_in_flight : 1); i++) frames_.emplace_back(device, pool);
    }
    void wait_for_frame() const { frames_[current_].in_flight_fence.wait(UINT64_MAX); }             // :299-304
    bool acquire_next_image(uint32_t image_count) { image_index_ = (image_index_ + 1) % image_count; return false; }   // :341-355
    void begin_frame() const {                                                                       // :380-386
        const FrameData& f = frames_[current_];
        f.in_flight_fence.reset(); f.command_buffer.reset(); f.command_buffer.begin();
    }
    void end_frame() const { frames_[current_].command_buffer.end(); }                               // :410-413
    void submit() const {                                                                            // :439-462
        mirhi_cmd* cmds[1] = {frames_[current_].command_buffer.handle()};
        check(mirhi_queue_submit(device_->handle(), 1, cmds, frames_[current_].in_flight_fence.handle()));
    }
    bool present() const { return false; }                                                           // :499-518 (offscreen: nothing to present)
    void next_frame() { current_ = (current_ + 1) % frames_.size(); }                                // :537-539
    void wait_for_all_frames() const { for (auto& f : frames_) if (!f.in_flight_fence.is_signaled()) f.in_flight_fence.wait(UINT64_MAX); }
    const FrameData& current_frame() const { return frames_[current_]; }
    size_t current_frame_index() const { return current_; }
    uint32_t image_index() const { return image_index_; }
private:
    std::shared_ptr<Device> device_;
    std::vector<FrameData> frames_;
    size_t current_ = 0;
    uint32_t image_index_ = 0;
};

// The hello-triangle renderer: same resources, same per-frame call order as renderer.rs:205-260,367-557;
// the swapchain is replaced by MAX_FRAMES_IN_FLIGHT offscreen B8G8R8A8_SRGB images that can be read back.
class Renderer {
public:
    Renderer(uint32_t width, uint32_t height, int ordinal = 0, Format swapchain_format = Format::B8G8R8A8_SRGB)
        : device_(Device::create(ordinal)), pool_(device_, 0), width_(width), height_(height), format_(swapchain_format),
          frames_(device_, pool_), pipeline_(create_triangle_pipeline()), vertex_buffer_(create_triangle_vertices()) {
        create_swapchain_images();
    }
    void resize(uint32_t width, uint32_t height) { if (width && height) { new_w_ = width; new_h_ = height; framebuffer_resized_ = true; } }   // renderer.rs:281-291
    void render_frame() {                                                                                                                      // renderer.rs:367-449
        if (framebuffer_resized_) recreate_swapchain();
        frames_.wait_for_frame();
        frames_.acquire_next_image((uint32_t)images_.size());
        frames_.begin_frame();
        record_commands(frames_.current_frame().command_buffer, frames_.image_index());
        frames_.end_frame();
        frames_.submit();
        frames_.present();
        last_image_ = frames_.image_index();
        frames_.next_frame();
    }
    void wait_idle() const { frames_.wait_for_all_frames(); device_->wait_idle(); }
    const Image& last_image() const { return images_[last_image_]; }
    uint32_t width() const { return width_; }
    uint32_t height() const { return height_; }
    const std::shared_ptr<Device>& device() const { return device_; }
private:
    Pipeline create_triangle_pipeline() const {   // renderer.rs:228-237
        return GraphicsPipelineBuilder().vertex_shader(ShaderProgram::Triangle).fragment_shader(ShaderProgram::Triangle)
            .vertex_binding(TriangleVertex::binding_description()).vertex_attributes(TriangleVertex::attribute_descriptions())
            .color_attachment_format(format_).cull_mode(CullMode::None).depth_test_enable(false).depth_write_enable(false).build(device_);
    }
    Buffer create_triangle_vertices() const {     // renderer.rs:242-250
        const TriangleVertex v[3] = {{{0.0f, -0.5f, 0.0f}, {1.0f, 0.0f, 0.0f}}, {{-0.5f, 0.5f, 0.0f}, {0.0f, 1.0f, 0.0f}}, {{0.5f, 0.5f, 0.0f}, {0.0f, 0.0f, 1.0f}}};
        return Buffer::new_with_data(device_, BufferUsage::Vertex, v, sizeof v);
    }
    void create_swapchain_images() {
        images_.clear();
        for (size_t i = 0; i < MAX_FRAMES_IN_FLIGHT + 1; i++) images_.emplace_back(device_, width_, height_, format_);   // min_image_count + 1 (swapchain.rs:228-236)
    }
    void recreate_swapchain() {                   // renderer.rs:301-320
        wait_idle();
        width_ = new_w_; height_ = new_h_;
        create_swapchain_images();
        framebuffer_resized_ = false;
    }
    void record_commands(const CommandBuffer& cmd, uint32_t image_index) const {   // renderer.rs:452-557
        ColorAttachment color(images_[image_index]);
        color.with_clear_color({0.1f, 0.1f, 0.15f, 1.0f});
        RenderingConfig cfg(width_, height_);
        cfg.with_color_attachment(color);
        cmd.begin_rendering(cfg.build());
        cmd.set_viewport({0.0f, 0.0f, (float)width_, (float)height_, 0.0f, 1.0f});
        cmd.set_scissor({0, 0, width_, height_});
        cmd.bind_pipeline(pipeline_);
        cmd.bind_vertex_buffers(0, vertex_buffer_, 0);
        cmd.draw(3, 1, 0, 0);
        cmd.end_rendering();
    }
    std::shared_ptr<Device> device_;
    CommandPool pool_;
    uint32_t width_, height_, new_w_ = 0, new_h_ = 0;
    Format format_;
    FrameManager frames_;
    Pipeline pipeline_;
    Buffer vertex_buffer_;
    std::vector<Image> images_;
    uint32_t last_image_ = 0;
    bool framebuffer_resized_ = false;
};

}  // namespace mirhi
