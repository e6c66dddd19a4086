//! crates/rhi/src/pipeline.rs:274-1059 and shader.rs:244-330.  Enum discriminants equal the C enum values of include/mirhi.h.
use crate::device::Device;
use crate::error::{check, RhiError, RhiResult};
use crate::image::Format;
use std::path::Path;
use std::sync::Arc;

#[repr(i32)] #[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum PrimitiveTopology { PointList = 0, LineList = 1, LineStrip = 2, TriangleList = 3, TriangleStrip = 4, TriangleFan = 5 }
#[repr(i32)] #[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum PolygonMode { Fill = 0, Line = 1, Point = 2 }
#[repr(i32)] #[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum CullMode { None = 0, Front = 1, Back = 2, FrontAndBack = 3 }
#[repr(i32)] #[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum FrontFace { CounterClockwise = 0, Clockwise = 1 }
#[repr(i32)] #[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum CompareOp { Never = 0, Less = 1, Equal = 2, LessOrEqual = 3, Greater = 4, NotEqual = 5, GreaterOrEqual = 6, Always = 7 }
#[repr(i32)] #[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum BlendFactor { Zero = 0, One = 1, SrcColor = 2, OneMinusSrcColor = 3, DstColor = 4, OneMinusDstColor = 5, SrcAlpha = 6,
                       OneMinusSrcAlpha = 7, DstAlpha = 8, OneMinusDstAlpha = 9 }
#[repr(i32)] #[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum BlendOp { Add = 0, Subtract = 1, ReverseSubtract = 2, Min = 3, Max = 4 }
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum ShaderStage { Vertex, Fragment }
#[repr(i32)] #[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum ShaderProgram { Triangle = 0, Model = 1, ModelFull = 2, ModelPbr = 3 }

/// pipeline.rs:499-529
#[derive(Clone, Copy, Debug)]
pub struct ColorBlendAttachment {
    pub blend_enable: bool,
    pub src_color: BlendFactor, pub dst_color: BlendFactor, pub color_op: BlendOp,
    pub src_alpha: BlendFactor, pub dst_alpha: BlendFactor, pub alpha_op: BlendOp,
    pub color_write_mask: u32,
}
impl Default for ColorBlendAttachment {
    fn default() -> Self {                                       // pipeline.rs:499-512: blending off, write mask RGBA
        Self { blend_enable: false, src_color: BlendFactor::One, dst_color: BlendFactor::Zero, color_op: BlendOp::Add,
               src_alpha: BlendFactor::One, dst_alpha: BlendFactor::Zero, alpha_op: BlendOp::Add, color_write_mask: 0xF }
    }
}
impl ColorBlendAttachment {
    pub fn alpha_blend() -> Self {                               // pipeline.rs:518-529
        Self { blend_enable: true, src_color: BlendFactor::SrcAlpha, dst_color: BlendFactor::OneMinusSrcAlpha, color_op: BlendOp::Add,
               src_alpha: BlendFactor::One, dst_alpha: BlendFactor::Zero, alpha_op: BlendOp::Add, color_write_mask: 0xF }
    }
}

/// shader.rs:244-330.  No SPIR-V at run time: the file name selects one of the precompiled programs.
pub struct Shader { program: ShaderProgram, stage: ShaderStage }
impl Shader {
    pub fn from_spirv_file(_device: Arc<Device>, path: impl AsRef<Path>, stage: ShaderStage, entry: &str) -> RhiResult<Self> {
        if entry != "main" {
            return Err(RhiError::ShaderError(format!("Shader error: entry point '{entry}' is not available (precompiled programs export 'main')")));
        }
        let name = path.as_ref().file_name().and_then(|s| s.to_str()).unwrap_or("").to_ascii_lowercase();
        let program = if name.starts_with("triangle") { ShaderProgram::Triangle }
                      else if name.starts_with("model_pbr") { ShaderProgram::ModelPbr }
                      else if name.starts_with("model_full") { ShaderProgram::ModelFull }
                      else if name.starts_with("model") { ShaderProgram::Model }
                      else { return Err(RhiError::ShaderError(format!("Shader error: no precompiled program for '{name}'"))); };
        Ok(Self { program, stage })
    }
    pub fn program(&self) -> ShaderProgram { self.program }
    pub fn stage(&self) -> ShaderStage { self.stage }
}

pub struct Pipeline {
    #[allow(dead_code)]
    device: Arc<Device>,
    pub(crate) raw: *mut mirhi_sys::mirhi_pipeline,
}
unsafe impl Send for Pipeline {}
impl Drop for Pipeline {
    fn drop(&mut self) { unsafe { mirhi_sys::mirhi_pipeline_destroy(self.raw) }; }
}

/// pipeline.rs:590-1059: same setters, same defaults (TriangleList, Fill, cull Back, CCW, depth test + write, Less; 645-698),
/// same validation failures from `build` (920-952) -- they are produced by `mirhi_pipeline_create`.
pub struct GraphicsPipelineBuilder { desc: mirhi_sys::mirhi_pipeline_desc }
impl Default for GraphicsPipelineBuilder { fn default() -> Self { Self::new() } }
impl GraphicsPipelineBuilder {
    pub fn new() -> Self {
        let mut d = std::mem::MaybeUninit::<mirhi_sys::mirhi_pipeline_desc>::zeroed();
        unsafe { mirhi_sys::mirhi_pipeline_desc_default(d.as_mut_ptr()) };
        Self { desc: unsafe { d.assume_init() } }
    }
    pub fn vertex_shader(mut self, s: &Shader) -> Self {
        // the vertex stage is TRIANGLE or MODEL: every model fragment program shares vertex/model.hlsl
        self.desc.vertex_program = if s.program() == ShaderProgram::Triangle { 0 } else { 1 };
        self
    }
    pub fn fragment_shader(mut self, s: &Shader) -> Self { self.desc.fragment_program = s.program() as i32; self }
    /// vertex.rs:35-61 / 130-170: one binding, attribute offsets in declaration order (TriangleVertex 24 B: 0, 12; Vertex 48 B: 0, 12, 24, 32)
    pub fn vertex_binding(mut self, stride: u32) -> Self { self.desc.vertex_stride = stride; self }
    pub fn vertex_attributes(mut self, offsets: &[u32]) -> Self {
        self.desc.attribute_count = offsets.len().min(4) as u32;
        for (i, o) in offsets.iter().take(4).enumerate() { self.desc.attribute_offsets[i] = *o; }
        self
    }
    pub fn topology(mut self, t: PrimitiveTopology) -> Self { self.desc.topology = t as i32; self }
    pub fn polygon_mode(mut self, m: PolygonMode) -> Self { self.desc.polygon_mode = m as i32; self }
    pub fn cull_mode(mut self, m: CullMode) -> Self { self.desc.cull_mode = m as i32; self }
    pub fn front_face(mut self, f: FrontFace) -> Self { self.desc.front_face = f as i32; self }
    pub fn depth_test_enable(mut self, on: bool) -> Self { self.desc.depth_test_enable = on as u32; self }
    pub fn depth_write_enable(mut self, on: bool) -> Self { self.desc.depth_write_enable = on as u32; self }
    pub fn depth_compare_op(mut self, op: CompareOp) -> Self { self.desc.depth_compare_op = op as i32; self }
    /// Alpha-masked MODEL_PBR materials (`discard`, model_pbr.hlsl:176-179): fragments are resolved one by one in primitive order.
    pub fn fragment_discard_enable(mut self, on: bool) -> Self { self.desc.fragment_discard_enable = on as u32; self }
    pub fn color_attachment_format(mut self, f: Format) -> Self {
        self.desc.color_attachment_count = 1; self.desc.color_attachment_formats[0] = f as i32; self
    }
    pub fn depth_attachment_format(mut self, f: Format) -> Self { self.desc.depth_attachment_format = f as i32; self }
    pub fn color_blend_attachment(mut self, a: ColorBlendAttachment) -> Self {
        self.desc.blend_attachment_count = 1;
        self.desc.blend_enable = a.blend_enable as u32;
        self.desc.src_color_blend_factor = a.src_color as i32; self.desc.dst_color_blend_factor = a.dst_color as i32; self.desc.color_blend_op = a.color_op as i32;
        self.desc.src_alpha_blend_factor = a.src_alpha as i32; self.desc.dst_alpha_blend_factor = a.dst_alpha as i32; self.desc.alpha_blend_op = a.alpha_op as i32;
        self.desc.color_write_mask = a.color_write_mask;
        self
    }
    pub fn build(self, device: Arc<Device>) -> RhiResult<Pipeline> {                                   // pipeline.rs:918-1057
        let mut raw = std::ptr::null_mut();
        check(unsafe { mirhi_sys::mirhi_pipeline_create(device.raw, &self.desc, &mut raw) })?;
        Ok(Pipeline { device, raw })
    }
}
