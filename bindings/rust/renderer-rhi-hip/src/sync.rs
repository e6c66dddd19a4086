//! crates/rhi/src/sync.rs:134-298.  Semaphores have no counterpart: submissions on one queue lane are ordered by the stream.
use crate::device::Device;
use crate::error::{check, RhiError, RhiResult};
use std::sync::Arc;

pub struct Fence {
    #[allow(dead_code)]
    device: Arc<Device>,
    pub(crate) raw: *mut mirhi_sys::mirhi_fence,
}
unsafe impl Send for Fence {}

impl Fence {
    pub fn new(device: Arc<Device>, signaled: bool) -> RhiResult<Self> {                   // sync.rs:168
        let mut raw = std::ptr::null_mut();
        check(unsafe { mirhi_sys::mirhi_fence_create(device.raw, signaled as u32, &mut raw) })?;
        Ok(Self { device, raw })
    }
    /// sync.rs:228: blocks up to `timeout` ns (u64::MAX = forever); device-side errors of the frame (a raster list that
    /// overflowed, an unsupported state combination) surface here, like a lost device would.
    pub fn wait(&self, timeout: u64) -> Result<(), RhiError> { check(unsafe { mirhi_sys::mirhi_fence_wait(self.raw, timeout) }) }
    pub fn reset(&self) -> Result<(), RhiError> { check(unsafe { mirhi_sys::mirhi_fence_reset(self.raw) }) }   // sync.rs:264
    pub fn is_signaled(&self) -> bool { unsafe { mirhi_sys::mirhi_fence_status(self.raw) == mirhi_sys::MIRHI_OK } }   // sync.rs:294
}

impl Drop for Fence {
    fn drop(&mut self) { unsafe { mirhi_sys::mirhi_fence_destroy(self.raw) }; }
}
