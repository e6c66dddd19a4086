//! crates/rhi/src/buffer.rs:47-436.
use crate::device::Device;
use crate::error::{check, RhiResult};
use std::sync::Arc;

#[repr(i32)]
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum BufferUsage { Vertex = 0, Index = 1, Uniform = 2, Storage = 3, Staging = 4, Indirect = 5 }   // buffer.rs:47-60

pub struct Buffer {
    #[allow(dead_code)]
    device: Arc<Device>,
    pub(crate) raw: *mut mirhi_sys::mirhi_buffer,
}
unsafe impl Send for Buffer {}

impl Buffer {
    pub fn new(device: Arc<Device>, usage: BufferUsage, size: u64) -> RhiResult<Self> {               // buffer.rs:149 (size 0 => InvalidHandle)
        let mut raw = std::ptr::null_mut();
        check(unsafe { mirhi_sys::mirhi_buffer_create(device.raw, usage as i32, size, &mut raw) })?;
        Ok(Self { device, raw })
    }
    pub fn new_with_data(device: Arc<Device>, usage: BufferUsage, data: &[u8]) -> RhiResult<Self> {   // buffer.rs:227
        let mut raw = std::ptr::null_mut();
        check(unsafe { mirhi_sys::mirhi_buffer_create_with_data(device.raw, usage as i32, data.as_ptr().cast(), data.len() as u64, &mut raw) })?;
        Ok(Self { device, raw })
    }
    pub fn write_data(&self, offset: u64, data: &[u8]) -> RhiResult<()> {                             // buffer.rs:247 (bounds-checked)
        check(unsafe { mirhi_sys::mirhi_buffer_write(self.raw, offset, data.as_ptr().cast(), data.len() as u64) })
    }
    pub fn upload(&self, data: &[u8]) -> RhiResult<()> {                                              // buffer.rs:291
        check(unsafe { mirhi_sys::mirhi_buffer_upload(self.raw, data.as_ptr().cast(), data.len() as u64) })
    }
    pub fn upload_via_staging(&self, data: &[u8]) -> RhiResult<()> {                                  // buffer.rs:345
        check(unsafe { mirhi_sys::mirhi_buffer_upload_via_staging(self.raw, data.as_ptr().cast(), data.len() as u64) })
    }
    pub fn size(&self) -> u64 { unsafe { mirhi_sys::mirhi_buffer_size(self.raw) } }
    pub fn usage(&self) -> BufferUsage {
        match unsafe { mirhi_sys::mirhi_buffer_usage_of(self.raw) } { 0 => BufferUsage::Vertex, 1 => BufferUsage::Index, 2 => BufferUsage::Uniform,
                                                                     3 => BufferUsage::Storage, 4 => BufferUsage::Staging, _ => BufferUsage::Indirect }
    }
}

impl Drop for Buffer {
    fn drop(&mut self) { unsafe { mirhi_sys::mirhi_buffer_destroy(self.raw) }; }                      // buffer.rs:420-436
}
