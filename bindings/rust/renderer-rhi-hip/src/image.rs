//! Attachments and textures: crates/renderer/src/depth_buffer.rs:117-127, crates/rhi/src/{image,texture}.rs (stubs in the reference).
use crate::device::Device;
use crate::error::{check, RhiResult};
use std::sync::Arc;

#[repr(i32)]
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum Format { Undefined = 0, B8G8R8A8Srgb = 1, R32G32B32A32Sfloat = 2, D32Sfloat = 3, R8G8B8A8Unorm = 4, R32Uint = 5, R8G8B8A8Srgb = 6 }

pub struct Image {
    #[allow(dead_code)]
    device: Arc<Device>,
    pub(crate) raw: *mut mirhi_sys::mirhi_image,
}
unsafe impl Send for Image {}

impl Image {
    pub fn new(device: Arc<Device>, width: u32, height: u32, format: Format) -> RhiResult<Self> {     // (zero size => error, depth_buffer.rs:117-127)
        let mut raw = std::ptr::null_mut();
        check(unsafe { mirhi_sys::mirhi_image_create(device.raw, width, height, format as i32, &mut raw) })?;
        Ok(Self { device, raw })
    }
    /// RGBA8 / float texels of level 0, row-major, top row first.
    pub fn upload(&self, texels: &[u8]) -> RhiResult<()> {
        check(unsafe { mirhi_sys::mirhi_image_upload(self.raw, texels.as_ptr().cast(), texels.len() as u64) })
    }
    /// Full mip chain behind level 0 (2x2 box on the stored bytes); the image is then sampled trilinearly.
    pub fn generate_mips(&self) -> RhiResult<()> { check(unsafe { mirhi_sys::mirhi_image_generate_mips(self.raw) }) }
    /// Readback of a finished target -- the reference presents instead (swapchain.rs:255) and has no such call.
    pub fn read(&self, dst: &mut [u8]) -> RhiResult<()> {
        check(unsafe { mirhi_sys::mirhi_image_read(self.raw, dst.as_mut_ptr().cast(), dst.len() as u64) })
    }
    pub fn width(&self) -> u32 { unsafe { mirhi_sys::mirhi_image_width(self.raw) } }
    pub fn height(&self) -> u32 { unsafe { mirhi_sys::mirhi_image_height(self.raw) } }
    pub fn size_bytes(&self) -> u64 { unsafe { mirhi_sys::mirhi_image_size_bytes(self.raw) } }
    pub fn mip_levels(&self) -> u32 { unsafe { mirhi_sys::mirhi_image_mip_levels(self.raw) } }
    /// Sampler state: 1 = trilinear, up to 16 = anisotropic (the device enables `sampler_anisotropy`, device.rs:161-165).
    pub fn set_max_anisotropy(&self, max_anisotropy: u32) -> RhiResult<()> {
        check(unsafe { mirhi_sys::mirhi_image_set_max_anisotropy(self.raw, max_anisotropy) })
    }
    pub fn max_anisotropy(&self) -> u32 { unsafe { mirhi_sys::mirhi_image_max_anisotropy(self.raw) } }
}

impl Drop for Image {
    fn drop(&mut self) { unsafe { mirhi_sys::mirhi_image_destroy(self.raw) }; }
}
