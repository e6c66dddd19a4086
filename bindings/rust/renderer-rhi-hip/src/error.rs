//! crates/rhi/src/error.rs:6-50 -- one variant per `mirhi_result` code.
use std::ffi::CStr;
use std::fmt;

#[derive(Debug)]
pub enum RhiError {
    VulkanError(String),        // the slot HIP / device errors take (MIRHI_ERR_DEVICE)
    LoadingError(String),
    AllocatorError(String),
    NoSuitableGpu,
    ShaderError(String),
    SurfaceError(String),
    SwapchainError(String),
    InvalidHandle(String),
    PipelineError(String),
    LockPoisoned(String),
    Timeout,
    NotReady,
}

pub type RhiResult<T> = Result<T, RhiError>;

impl fmt::Display for RhiError {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        match self {
            RhiError::NoSuitableGpu => write!(f, "No suitable GPU found"),
            RhiError::Timeout => write!(f, "Vulkan error: TIMEOUT"),
            RhiError::NotReady => write!(f, "Vulkan error: NOT_READY"),
            RhiError::VulkanError(m) | RhiError::LoadingError(m) | RhiError::AllocatorError(m) | RhiError::ShaderError(m)
            | RhiError::SurfaceError(m) | RhiError::SwapchainError(m) | RhiError::InvalidHandle(m) | RhiError::PipelineError(m)
            | RhiError::LockPoisoned(m) => write!(f, "{m}"),
        }
    }
}
impl std::error::Error for RhiError {}

pub(crate) fn last_message() -> String {
    // the string is thread-local inside libmirhi and stays valid until the next failing call on this thread
    unsafe { CStr::from_ptr(mirhi_sys::mirhi_last_error_message()) }.to_string_lossy().into_owned()
}

pub(crate) fn check(code: mirhi_sys::mirhi_result) -> RhiResult<()> {
    use mirhi_sys::*;
    if code == MIRHI_OK {
        return Ok(());
    }
    let m = last_message();
    Err(match code {
        MIRHI_ERR_LOADING => RhiError::LoadingError(m),
        MIRHI_ERR_ALLOCATOR => RhiError::AllocatorError(m),
        MIRHI_ERR_NO_SUITABLE_GPU => RhiError::NoSuitableGpu,
        MIRHI_ERR_SHADER => RhiError::ShaderError(m),
        MIRHI_ERR_SURFACE => RhiError::SurfaceError(m),
        MIRHI_ERR_SWAPCHAIN => RhiError::SwapchainError(m),
        MIRHI_ERR_INVALID_HANDLE => RhiError::InvalidHandle(m),
        MIRHI_ERR_PIPELINE => RhiError::PipelineError(m),
        MIRHI_ERR_LOCK_POISONED => RhiError::LockPoisoned(m),
        MIRHI_TIMEOUT => RhiError::Timeout,
        MIRHI_NOT_READY => RhiError::NotReady,
        _ => RhiError::VulkanError(m),
    })
}
