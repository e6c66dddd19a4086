//! crates/rhi/src/device.rs:61-380.  `Device::new` stands in for Instance + physical-device selection + logical device.
use crate::error::{check, RhiResult};
use std::sync::Arc;

pub struct Device {
    pub(crate) raw: *mut mirhi_sys::mirhi_device,
}
// device.rs:379-380: the reference's Device is Send + Sync; libmirhi guards its per-device state with a mutex
unsafe impl Send for Device {}
unsafe impl Sync for Device {}

#[derive(Clone, Copy, Debug, Default)]
pub struct DeviceStats {
    pub frames_submitted: u64,
    pub triangles_submitted: u64,
    pub workspace_bytes: u64,
    pub last_big_list: u32,
    pub last_status: u32,
}

impl Device {
    /// `hip_ordinal`: the GPU this process owns (one process per GPU; LOCAL_RANK under torchrun-style launchers).
    pub fn new(hip_ordinal: i32) -> RhiResult<Arc<Self>> {
        // the #[repr(C)] structs of mirhi-sys are one ABI's: a library of another one must not be driven with them
        let abi = unsafe { mirhi_sys::mirhi_abi_version() };
        if abi != mirhi_sys::MIRHI_ABI_VERSION {
            return Err(crate::error::RhiError::LoadingError(format!("libmirhi.so has ABI {abi}, mirhi-sys was generated for ABI {}", mirhi_sys::MIRHI_ABI_VERSION)));
        }
        let mut raw = std::ptr::null_mut();
        check(unsafe { mirhi_sys::mirhi_device_create(hip_ordinal, &mut raw) })?;
        Ok(Arc::new(Self { raw }))
    }
    pub fn count() -> RhiResult<i32> {
        let mut n = 0;
        check(unsafe { mirhi_sys::mirhi_device_count(&mut n) })?;
        Ok(n)
    }
    pub fn wait_idle(&self) -> RhiResult<()> {                       // device.rs:290-293
        check(unsafe { mirhi_sys::mirhi_device_wait_idle(self.raw) })
    }
    pub fn name(&self) -> String {
        let mut buf = [0 as std::os::raw::c_char; 256];
        unsafe { mirhi_sys::mirhi_device_name(self.raw, buf.as_mut_ptr(), buf.len() as u32) };
        unsafe { std::ffi::CStr::from_ptr(buf.as_ptr()) }.to_string_lossy().into_owned()
    }
    /// Frames in flight run on separate HIP streams ("queue lanes"); frame_manager.rs's MAX_FRAMES_IN_FLIGHT goes here.
    pub fn set_queue_lanes(&self, lanes: u32) -> RhiResult<()> {
        check(unsafe { mirhi_sys::mirhi_device_set_queue_lanes(self.raw, lanes) })
    }
    /// vkQueueSubmit semantics: `queue_submit` queues the work and returns, a thread of the device makes the kernel launches
    /// (renderer.rs:407-424 returns as soon as the driver has the submission).  Off by default.
    pub fn set_submit_thread(&self, enable: bool) -> RhiResult<()> {
        check(unsafe { mirhi_sys::mirhi_device_set_submit_thread(self.raw, enable as u32) })
    }
    /// Screen-tile-row split across the GPUs of a node: this process rasters band `rank` of `world`.
    pub fn set_tile_split(&self, rank: u32, world: u32) -> RhiResult<()> {
        check(unsafe { mirhi_sys::mirhi_device_set_tile_split(self.raw, rank, world) })
    }
    /// Which tile rows a rank of the split gets: `interleaved` (rows rank, rank + world, ...: the default) or one contiguous band.  Every rank
    /// the same layout, before `set_tile_split` / the communicator.
    pub fn set_tile_split_interleaved(&self, interleaved: bool) -> RhiResult<()> {
        let layout = if interleaved { mirhi_sys::MIRHI_SPLIT_INTERLEAVED } else { mirhi_sys::MIRHI_SPLIT_BANDS };
        check(unsafe { mirhi_sys::mirhi_device_set_tile_split_layout(self.raw, layout) })
    }
    /// A device made on the caller's HIP stream keeps queue lane 0 on that stream (HIP launches); `true` lets lane 0 leave the library as AQL
    /// packets on its own ROCr queue like the other lanes (the caller then orders its stream against the frames with fences / wait_idle).
    pub fn set_native_dispatch(&self, enable: bool) -> RhiResult<()> {
        check(unsafe { mirhi_sys::mirhi_device_set_native_dispatch(self.raw, enable as u32) })
    }
    /// Submit recorded command buffers in order; semaphores of `vkQueueSubmit` collapse to stream order (renderer.rs:407-424).
    pub fn submit(&self, cmds: &[&crate::CommandBuffer], fence: Option<&crate::Fence>) -> RhiResult<()> {
        let raws: Vec<*mut mirhi_sys::mirhi_cmd> = cmds.iter().map(|c| c.raw).collect();
        let f = fence.map_or(std::ptr::null_mut(), |f| f.raw);
        check(unsafe { mirhi_sys::mirhi_queue_submit(self.raw, raws.len() as u32, raws.as_ptr(), f) })
    }
    pub fn stats(&self) -> RhiResult<DeviceStats> {
        let mut s = std::mem::MaybeUninit::<mirhi_sys::mirhi_device_stats>::zeroed();
        check(unsafe { mirhi_sys::mirhi_device_get_stats(self.raw, s.as_mut_ptr()) })?;
        let s = unsafe { s.assume_init() };
        Ok(DeviceStats { frames_submitted: s.frames_submitted, triangles_submitted: s.triangles_submitted,
                         workspace_bytes: s.workspace_bytes, last_big_list: s.last_big_list, last_status: s.last_status })
    }
}

impl Drop for Device {
    fn drop(&mut self) {
        // children hold an Arc<Device>, so none is alive here (the C side would refuse otherwise: "device still has N live child objects")
        unsafe { mirhi_sys::mirhi_device_destroy(self.raw) };
    }
}
