//! crates/rhi/src/command.rs:52-628 and rendering.rs:65-115,319-370: recording is host-only, one recorder per thread.
use crate::buffer::Buffer;
use crate::device::Device;
use crate::error::{check, RhiResult};
use crate::image::Image;
use crate::pipeline::Pipeline;
use std::sync::Arc;

#[repr(i32)] #[derive(Clone, Copy, Debug, PartialEq, Eq)] pub enum LoadOp { Load = 0, Clear = 1, DontCare = 2 }
#[repr(i32)] #[derive(Clone, Copy, Debug, PartialEq, Eq)] pub enum StoreOp { Store = 0, DontCare = 1 }
#[repr(i32)] #[derive(Clone, Copy, Debug, PartialEq, Eq)] pub enum IndexType { Uint16 = 0, Uint32 = 1 }
/// The registers of the HLSL the reference binds through descriptor sets (b0..b3, t0/t1 space1): INTEGRATION.md section 4.
#[repr(i32)] #[derive(Clone, Copy, Debug, PartialEq, Eq)] pub enum UniformSlot { Camera = 0, Object = 1, Lights = 2, Material = 3, PointLights = 4, SpotLights = 5 }
/// model_pbr.hlsl:62-95 slot order
#[repr(i32)] #[derive(Clone, Copy, Debug, PartialEq, Eq)] pub enum TextureSlot { Albedo = 0, Normal = 1, MetallicRoughness = 2, Occlusion = 3, Emissive = 4 }

#[derive(Clone, Copy, Debug)] pub struct Viewport { pub x: f32, pub y: f32, pub width: f32, pub height: f32, pub min_depth: f32, pub max_depth: f32 }
#[derive(Clone, Copy, Debug)] pub struct Rect2D { pub x: i32, pub y: i32, pub width: u32, pub height: u32 }
#[derive(Clone, Copy, Debug)] pub enum ClearValue { Color([f32; 4]), Depth(f32) }

/// What `RenderingConfig::build` (rendering.rs:908-995) hands to `begin_rendering`: one colour attachment, optional depth.
pub struct RenderingInfo<'a> {
    pub color: &'a Image, pub color_load_op: LoadOp, pub color_store_op: StoreOp, pub clear_color: [f32; 4],
    pub depth: Option<&'a Image>, pub depth_load_op: LoadOp, pub depth_store_op: StoreOp, pub clear_depth: f32,
    pub render_area: Rect2D,
}
impl<'a> RenderingInfo<'a> {
    /// rendering.rs defaults: colour CLEAR/STORE to (0,0,0,1); depth CLEAR/DONT_CARE to 1.0 (1027-1073)
    pub fn new(color: &'a Image) -> Self {
        Self { color, color_load_op: LoadOp::Clear, color_store_op: StoreOp::Store, clear_color: [0.0, 0.0, 0.0, 1.0],
               depth: None, depth_load_op: LoadOp::Clear, depth_store_op: StoreOp::DontCare, clear_depth: 1.0,
               render_area: Rect2D { x: 0, y: 0, width: color.width(), height: color.height() } }
    }
}

/// command.rs:52-106.  libmirhi keeps one workspace per command buffer; the pool only carries the device.
pub struct CommandPool { device: Arc<Device> }
impl CommandPool {
    pub fn new(device: Arc<Device>, _queue_family_index: u32) -> RhiResult<Self> { Ok(Self { device }) }
    pub fn device(&self) -> &Arc<Device> { &self.device }
}

pub struct CommandBuffer {
    #[allow(dead_code)]
    device: Arc<Device>,
    pub(crate) raw: *mut mirhi_sys::mirhi_cmd,
}
unsafe impl Send for CommandBuffer {}

impl CommandBuffer {
    pub fn new(device: Arc<Device>, _pool: &CommandPool) -> RhiResult<Self> {                           // command.rs:297
        let mut raw = std::ptr::null_mut();
        check(unsafe { mirhi_sys::mirhi_cmd_create(device.raw, &mut raw) })?;
        Ok(Self { device, raw })
    }
    pub fn begin(&self) -> RhiResult<()> { check(unsafe { mirhi_sys::mirhi_cmd_begin(self.raw) }) }                   // :333 one-time submit
    pub fn begin_reusable(&self) -> RhiResult<()> { check(unsafe { mirhi_sys::mirhi_cmd_begin_reusable(self.raw) }) } // :353
    pub fn end(&self) -> RhiResult<()> { check(unsafe { mirhi_sys::mirhi_cmd_end(self.raw) }) }                       // :372
    pub fn reset(&self) -> RhiResult<()> { check(unsafe { mirhi_sys::mirhi_cmd_reset(self.raw) }) }                   // :387

    // The reference's recording calls return (); libmirhi reports misuse (wrong state, missing binding) -- surfaced here.
    pub fn begin_rendering(&self, info: &RenderingInfo<'_>) -> RhiResult<()> {                                       // :408
        let mut ri = std::mem::MaybeUninit::<mirhi_sys::mirhi_rendering_info>::zeroed();
        unsafe { mirhi_sys::mirhi_rendering_info_default(ri.as_mut_ptr()) };
        let mut ri = unsafe { ri.assume_init() };
        ri.color_image = info.color.raw; ri.color_load_op = info.color_load_op as i32; ri.color_store_op = info.color_store_op as i32;
        ri.clear_color = info.clear_color;
        ri.depth_image = info.depth.map_or(std::ptr::null_mut(), |d| d.raw);
        ri.depth_load_op = info.depth_load_op as i32; ri.depth_store_op = info.depth_store_op as i32; ri.clear_depth = info.clear_depth;
        ri.render_area = [info.render_area.x, info.render_area.y, info.render_area.width as i32, info.render_area.height as i32];
        check(unsafe { mirhi_sys::mirhi_cmd_begin_rendering(self.raw, &ri) })
    }
    pub fn end_rendering(&self) -> RhiResult<()> { check(unsafe { mirhi_sys::mirhi_cmd_end_rendering(self.raw) }) }  // :417
    pub fn bind_pipeline(&self, pipeline: &Pipeline) -> RhiResult<()> {                                              // :433
        check(unsafe { mirhi_sys::mirhi_cmd_bind_pipeline(self.raw, pipeline.raw) })
    }
    pub fn bind_vertex_buffers(&self, first_binding: u32, buffers: &[&Buffer], offsets: &[u64]) -> RhiResult<()> {   // :448
        let raws: Vec<*mut mirhi_sys::mirhi_buffer> = buffers.iter().map(|b| b.raw).collect();
        check(unsafe { mirhi_sys::mirhi_cmd_bind_vertex_buffers(self.raw, first_binding, raws.len() as u32, raws.as_ptr(), offsets.as_ptr()) })
    }
    pub fn bind_index_buffer(&self, buffer: &Buffer, offset: u64, index_type: IndexType) -> RhiResult<()> {          // :471
        check(unsafe { mirhi_sys::mirhi_cmd_bind_index_buffer(self.raw, buffer.raw, offset, index_type as i32) })
    }
    /// Stands in for `bind_descriptor_sets` (:493): one call per register; `range` 0 = to the end of the buffer.
    pub fn bind_uniform(&self, slot: UniformSlot, buffer: &Buffer, offset: u64, range: u64) -> RhiResult<()> {
        check(unsafe { mirhi_sys::mirhi_cmd_bind_uniform(self.raw, slot as i32, buffer.raw, offset, range) })
    }
    pub fn bind_texture(&self, slot: TextureSlot, image: Option<&Image>) -> RhiResult<()> {
        check(unsafe { mirhi_sys::mirhi_cmd_bind_texture(self.raw, slot as i32, image.map_or(std::ptr::null_mut(), |i| i.raw)) })
    }
    pub fn set_viewport(&self, v: &Viewport) -> RhiResult<()> {                                                      // :522
        let raw = mirhi_sys::mirhi_viewport { x: v.x, y: v.y, width: v.width, height: v.height, min_depth: v.min_depth, max_depth: v.max_depth };
        check(unsafe { mirhi_sys::mirhi_cmd_set_viewport(self.raw, &raw) })
    }
    pub fn set_scissor(&self, s: &Rect2D) -> RhiResult<()> {                                                         // :549
        let raw = mirhi_sys::mirhi_rect2d { x: s.x, y: s.y, width: s.width, height: s.height };
        check(unsafe { mirhi_sys::mirhi_cmd_set_scissor(self.raw, &raw) })
    }
    pub fn draw(&self, vertex_count: u32, instance_count: u32, first_vertex: u32, first_instance: u32) -> RhiResult<()> {   // :583
        check(unsafe { mirhi_sys::mirhi_cmd_draw(self.raw, vertex_count, instance_count, first_vertex, first_instance) })
    }
    pub fn draw_indexed(&self, index_count: u32, instance_count: u32, first_index: u32, vertex_offset: i32, first_instance: u32) -> RhiResult<()> {   // :610
        check(unsafe { mirhi_sys::mirhi_cmd_draw_indexed(self.raw, index_count, instance_count, first_index, vertex_offset, first_instance) })
    }
    /// Arguments are read from `buffer` when this call records (libmirhi latches them; Vulkan reads them at execution).
    pub fn draw_indirect(&self, buffer: &Buffer, offset: u64, draw_count: u32, stride: u32) -> RhiResult<()> {              // :630
        check(unsafe { mirhi_sys::mirhi_cmd_draw_indirect(self.raw, buffer.raw, offset, draw_count, stride) })
    }
    pub fn draw_indexed_indirect(&self, buffer: &Buffer, offset: u64, draw_count: u32, stride: u32) -> RhiResult<()> {      // :646
        check(unsafe { mirhi_sys::mirhi_cmd_draw_indexed_indirect(self.raw, buffer.raw, offset, draw_count, stride) })
    }
    pub fn push_constants_bytes(&self, stages: u32, offset: u32, data: &[u8]) -> RhiResult<()> {                            // :752
        check(unsafe { mirhi_sys::mirhi_cmd_push_constants(self.raw, stages, offset, data.as_ptr() as *const std::os::raw::c_void, data.len() as u32) })
    }
    pub fn push_constants<T: Copy>(&self, stages: u32, offset: u32, data: &T) -> RhiResult<()> {                            // :732
        let bytes = unsafe { std::slice::from_raw_parts(data as *const T as *const u8, std::mem::size_of::<T>()) };
        self.push_constants_bytes(stages, offset, bytes)
    }
    /// Queue lane (`Device::set_queue_lanes`) this command buffer is submitted on; several command buffers handed to one
    /// `Queue::submit` that are frames of the same shape run as one batch of launches on the first one's lane.
    pub fn set_queue_lane(&self, lane: u32) -> RhiResult<()> {
        check(unsafe { mirhi_sys::mirhi_cmd_set_queue_lane(self.raw, lane) })
    }
}

impl Drop for CommandBuffer {
    fn drop(&mut self) { unsafe { mirhi_sys::mirhi_cmd_destroy(self.raw) }; }
}
