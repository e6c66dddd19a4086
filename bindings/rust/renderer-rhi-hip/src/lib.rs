//! `renderer-rhi-hip`: the draw-path surface of the reference's `crates/rhi` (lib.rs:12-34) over `libmirhi.so`.
//!
//! Same type and method names, same `Arc<Device>`-holding RAII objects, same `RhiError` variants; the `ash::vk` value
//! types the reference passes through (`vk::Viewport`, `vk::Rect2D`, `vk::RenderingInfo`, ...) are replaced by plain
//! structs of the same shape, and `Shader` maps a SPIR-V file name to a precompiled fragment / vertex program.
//! Out of scope, as in SURVEY.md section 8: instance / physical-device selection, surface, swapchain, descriptor pools.
//!
//! NOT COMPILED in the build image (no Rust toolchain there): written against the generated `mirhi-sys` crate and kept in
//! step with `include/mirhi.h` by review; `INTEGRATION.md` walks through the few lines of `crates/renderer` that change.

mod buffer;
mod command;
mod device;
mod error;
mod image;
mod pipeline;
mod sync;

pub use buffer::{Buffer, BufferUsage};
pub use command::{ClearValue, CommandBuffer, CommandPool, IndexType, LoadOp, Rect2D, RenderingInfo, StoreOp, TextureSlot, UniformSlot, Viewport};
pub use device::{Device, DeviceStats};
pub use error::{RhiError, RhiResult};
pub use image::{Format, Image};
pub use pipeline::{BlendFactor, BlendOp, ColorBlendAttachment, CompareOp, CullMode, FrontFace, GraphicsPipelineBuilder, Pipeline, PolygonMode, PrimitiveTopology, Shader, ShaderProgram, ShaderStage};
pub use sync::Fence;

/// `MAX_FRAMES_IN_FLIGHT` of crates/renderer/src/frame_manager.rs; libmirhi runs each frame in flight on its own queue lane.
pub const MAX_FRAMES_IN_FLIGHT: usize = 2;
