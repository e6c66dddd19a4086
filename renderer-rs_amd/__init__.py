"""renderer_rs_amd -- Python host binding of libmirhi.so, the MI355X-native compute rasterizer.

The product is the C-ABI shared library (include/mirhi.h); this module is a thin ctypes mirror of
the reference's `crates/rhi` object names (Device, Buffer, BufferUsage, GraphicsPipelineBuilder,
CommandBuffer, Fence, ...) used by tests/ and bench.py.  There is no CPU fallback: if the library
cannot be loaded the import fails, and creating a Device without a gfx950 GPU raises RhiError.

The directory name contains a hyphen, so load it with `load_package()` from
`__graft_entry__.py` (registers the module as `renderer_rs_amd`).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

from . import build as _build
from . import scenes  # noqa: F401  (re-export)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, os.environ.get("MIRHI_LIB_NAME", "libmirhi.so"))   # MIRHI_LIB_NAME: diagnostic builds only
INCLUDE = os.path.join(os.path.dirname(HERE), "include", "mirhi.h")

# ---- enums (include/mirhi.h) ------------------------------------------------------------------------
OK, ERR_DEVICE, ERR_LOADING, ERR_ALLOCATOR, ERR_NO_SUITABLE_GPU, ERR_SHADER, ERR_SURFACE, ERR_SWAPCHAIN, \
    ERR_INVALID_HANDLE, ERR_PIPELINE, ERR_LOCK_POISONED, TIMEOUT, NOT_READY = range(13)


class BufferUsage:  # crates/rhi/src/buffer.rs:47-60
    Vertex, Index, Uniform, Storage, Staging, Indirect = range(6)


class Format:
    UNDEFINED, B8G8R8A8_SRGB, R32G32B32A32_SFLOAT, D32_SFLOAT, R8G8B8A8_UNORM, R32_UINT, R8G8B8A8_SRGB = range(7)


class Program:
    NONE, TRIANGLE, MODEL, MODEL_FULL, MODEL_PBR = -1, 0, 1, 2, 3


class PrimitiveTopology:  # pipeline.rs:274-282
    PointList, LineList, LineStrip, TriangleList, TriangleStrip, TriangleFan = range(6)


class PolygonMode:
    Fill, Line, Point = range(3)


class CullMode:  # pipeline.rs:329-336
    NONE, Front, Back, FrontAndBack = range(4)


class FrontFace:
    CounterClockwise, Clockwise = range(2)


class CompareOp:  # pipeline.rs:375-386
    Never, Less, Equal, LessOrEqual, Greater, NotEqual, GreaterOrEqual, Always = range(8)


class BlendFactor:  # pipeline.rs:411-448
    (Zero, One, SrcColor, OneMinusSrcColor, DstColor, OneMinusDstColor, SrcAlpha, OneMinusSrcAlpha, DstAlpha, OneMinusDstAlpha,
     ConstantColor, OneMinusConstantColor, ConstantAlpha, OneMinusConstantAlpha, SrcAlphaSaturate) = range(15)


class BlendOp:  # pipeline.rs:452-476
    Add, Subtract, ReverseSubtract, Min, Max = range(5)


class LoadOp:
    LOAD, CLEAR, DONT_CARE = range(3)


class StoreOp:
    STORE, DONT_CARE = range(2)


class IndexType:
    UINT16, UINT32 = range(2)


class Slot:
    CAMERA, OBJECT, LIGHTS, MATERIAL, POINT_LIGHTS, SPOT_LIGHTS = range(6)


class TextureSlot:
    ALBEDO, NORMAL, METALLIC_ROUGHNESS, OCCLUSION, EMISSIVE = range(5)


class Kernel:
    GEOMETRY, RASTER, VERTEX, FRAGMENT_COUNT = range(4)
    NAMES = ("geometry", "raster", "vertex", "fragment_count")


class Profile:
    TIMING, FRAGMENTS = 1, 2


class GatherAlgo:
    DIRECT, BROADCAST = range(2)


class SplitLayout:
    """mirhi_split_layout: which tile rows a rank of a tile split rasterizes"""
    BANDS, INTERLEAVED = range(2)
    NAMES = {"bands": 0, "interleaved": 1}


ABI_VERSION = 5                 # MIRHI_ABI_VERSION of include/mirhi.h this file mirrors
COMM_ID_BYTES = 128


MAX_FRAMES_IN_FLIGHT = 2  # crates/renderer/src/lib.rs:43


# ---- structs ------------------------------------------------------------------------------------------
class PipelineDesc(C.Structure):
    _fields_ = [
        ("vertex_program", C.c_int32), ("fragment_program", C.c_int32), ("vertex_stride", C.c_uint32),
        ("attribute_count", C.c_uint32), ("attribute_offsets", C.c_uint32 * 4),
        ("topology", C.c_int32), ("polygon_mode", C.c_int32), ("cull_mode", C.c_int32), ("front_face", C.c_int32),
        ("depth_clamp_enable", C.c_uint32), ("rasterizer_discard_enable", C.c_uint32), ("depth_bias_enable", C.c_uint32),
        ("rasterization_samples", C.c_uint32), ("depth_test_enable", C.c_uint32), ("depth_write_enable", C.c_uint32),
        ("depth_compare_op", C.c_int32), ("blend_enable", C.c_uint32), ("blend_attachment_count", C.c_uint32),
        ("color_attachment_count", C.c_uint32), ("color_attachment_formats", C.c_int32 * 4),
        ("depth_attachment_format", C.c_int32),
        ("src_color_blend_factor", C.c_int32), ("dst_color_blend_factor", C.c_int32), ("color_blend_op", C.c_int32),
        ("src_alpha_blend_factor", C.c_int32), ("dst_alpha_blend_factor", C.c_int32), ("alpha_blend_op", C.c_int32),
        ("color_write_mask", C.c_uint32),
        ("fragment_discard_enable", C.c_uint32),
    ]


class RenderingInfo(C.Structure):
    _fields_ = [
        ("color_image", C.c_void_p), ("color_load_op", C.c_int32), ("color_store_op", C.c_int32),
        ("clear_color", C.c_float * 4),
        ("depth_image", C.c_void_p), ("depth_load_op", C.c_int32), ("depth_store_op", C.c_int32),
        ("clear_depth", C.c_float), ("render_area", C.c_int32 * 4), ("prim_id_image", C.c_void_p),
    ]


class Viewport(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("width", C.c_float), ("height", C.c_float),
                ("min_depth", C.c_float), ("max_depth", C.c_float)]


class Rect2D(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("width", C.c_uint32), ("height", C.c_uint32)]


class DispatchTime(C.Structure):
    _fields_ = [("kernel", C.c_uint32), ("lane", C.c_uint32), ("begin_us", C.c_double), ("end_us", C.c_double)]


class DeviceStats(C.Structure):
    _fields_ = [("frames_submitted", C.c_uint64), ("triangles_submitted", C.c_uint64), ("workspace_bytes", C.c_uint64),
                ("last_big_list", C.c_uint32), ("last_status", C.c_uint32), ("last_bin_pages", C.c_uint32), ("native_dispatches", C.c_uint32),
                ("dispatch_path", C.c_uint32), ("device_lost", C.c_uint32), ("reserved", C.c_uint32)]


class RhiError(RuntimeError):
    """crates/rhi/src/error.rs:6-50; `.code` is the mirhi_result, `.variant` the RhiError variant name."""

    def __init__(self, code: int, message: str, variant: str):
        super().__init__(f"[{variant}] {message}")
        self.code, self.message, self.variant = code, message, variant


# ---- library loading ------------------------------------------------------------------------------------
_lib = None

_SIGNATURES = {
    "mirhi_last_error_message": (C.c_char_p, []),
    "mirhi_result_name": (C.c_char_p, [C.c_int32]),
    "mirhi_abi_version": (C.c_uint32, []),
    "mirhi_device_count": (C.c_int32, [C.POINTER(C.c_int32)]),
    "mirhi_device_create": (C.c_int32, [C.c_int32, C.POINTER(C.c_void_p)]),
    "mirhi_device_create_on_stream": (C.c_int32, [C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]),
    "mirhi_device_wait_idle": (C.c_int32, [C.c_void_p]),
    "mirhi_device_destroy": (C.c_int32, [C.c_void_p]),
    "mirhi_device_name": (C.c_int32, [C.c_void_p, C.c_char_p, C.c_uint32]),
    "mirhi_device_set_tile_split": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "mirhi_device_set_queue_lanes": (C.c_int32, [C.c_void_p, C.c_uint32]),
    "mirhi_device_set_submit_thread": (C.c_int32, [C.c_void_p, C.c_uint32]),
    "mirhi_device_band_rows": (C.c_int32, [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "mirhi_device_set_tile_split_layout": (C.c_int32, [C.c_void_p, C.c_int32]),
    "mirhi_device_split_rows": (C.c_int32, [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "mirhi_buffer_create": (C.c_int32, [C.c_void_p, C.c_int32, C.c_uint64, C.POINTER(C.c_void_p)]),
    "mirhi_buffer_create_with_data": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "mirhi_buffer_write": (C.c_int32, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]),
    "mirhi_buffer_upload": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "mirhi_buffer_upload_via_staging": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "mirhi_buffer_wrap_device_memory": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "mirhi_buffer_read": (C.c_int32, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]),
    "mirhi_buffer_size": (C.c_uint64, [C.c_void_p]),
    "mirhi_buffer_usage_of": (C.c_int32, [C.c_void_p]),
    "mirhi_buffer_device_ptr": (C.c_void_p, [C.c_void_p]),
    "mirhi_buffer_destroy": (C.c_int32, [C.c_void_p]),
    "mirhi_image_create": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int32, C.POINTER(C.c_void_p)]),
    "mirhi_image_wrap_device_memory": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]),
    "mirhi_image_upload": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "mirhi_image_read": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "mirhi_image_generate_mips": (C.c_int32, [C.c_void_p]),
    "mirhi_image_set_max_anisotropy": (C.c_int32, [C.c_void_p, C.c_uint32]),
    "mirhi_image_max_anisotropy": (C.c_uint32, [C.c_void_p]),
    "mirhi_image_mip_levels": (C.c_uint32, [C.c_void_p]),
    "mirhi_image_width": (C.c_uint32, [C.c_void_p]),
    "mirhi_image_height": (C.c_uint32, [C.c_void_p]),
    "mirhi_image_format": (C.c_int32, [C.c_void_p]),
    "mirhi_image_size_bytes": (C.c_uint64, [C.c_void_p]),
    "mirhi_image_device_ptr": (C.c_void_p, [C.c_void_p]),
    "mirhi_image_destroy": (C.c_int32, [C.c_void_p]),
    "mirhi_pipeline_desc_default": (None, [C.POINTER(PipelineDesc)]),
    "mirhi_pipeline_create": (C.c_int32, [C.c_void_p, C.POINTER(PipelineDesc), C.POINTER(C.c_void_p)]),
    "mirhi_pipeline_destroy": (C.c_int32, [C.c_void_p]),
    "mirhi_rendering_info_default": (None, [C.POINTER(RenderingInfo)]),
    "mirhi_cmd_create": (C.c_int32, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "mirhi_cmd_destroy": (C.c_int32, [C.c_void_p]),
    "mirhi_cmd_set_queue_lane": (C.c_int32, [C.c_void_p, C.c_uint32]),
    "mirhi_cmd_draw_indirect": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32]),
    "mirhi_cmd_draw_indexed_indirect": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32]),
    "mirhi_cmd_push_constants": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32]),
    "mirhi_cmd_begin": (C.c_int32, [C.c_void_p]),
    "mirhi_cmd_begin_reusable": (C.c_int32, [C.c_void_p]),
    "mirhi_cmd_end": (C.c_int32, [C.c_void_p]),
    "mirhi_cmd_reset": (C.c_int32, [C.c_void_p]),
    "mirhi_cmd_begin_rendering": (C.c_int32, [C.c_void_p, C.POINTER(RenderingInfo)]),
    "mirhi_cmd_end_rendering": (C.c_int32, [C.c_void_p]),
    "mirhi_cmd_bind_pipeline": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "mirhi_cmd_bind_vertex_buffers": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    "mirhi_cmd_bind_index_buffer": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32]),
    "mirhi_cmd_bind_uniform": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_uint64, C.c_uint64]),
    "mirhi_cmd_bind_texture": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p]),
    "mirhi_cmd_set_viewport": (C.c_int32, [C.c_void_p, C.POINTER(Viewport)]),
    "mirhi_cmd_set_scissor": (C.c_int32, [C.c_void_p, C.POINTER(Rect2D)]),
    "mirhi_cmd_draw": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "mirhi_cmd_draw_indexed": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, C.c_uint32]),
    "mirhi_queue_submit": (C.c_int32, [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.c_void_p]),
    "mirhi_fence_create": (C.c_int32, [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]),
    "mirhi_fence_wait": (C.c_int32, [C.c_void_p, C.c_uint64]),
    "mirhi_fence_reset": (C.c_int32, [C.c_void_p]),
    "mirhi_fence_status": (C.c_int32, [C.c_void_p]),
    "mirhi_fence_destroy": (C.c_int32, [C.c_void_p]),
    "mirhi_device_set_profiling": (C.c_int32, [C.c_void_p, C.c_uint32]),
    "mirhi_device_kernel_time": (C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "mirhi_device_reset_kernel_times": (C.c_int32, [C.c_void_p]),
    "mirhi_device_timeline": (C.c_int32, [C.c_void_p, C.POINTER(DispatchTime), C.c_uint32, C.POINTER(C.c_uint32)]),
    "mirhi_device_fragment_stats": (C.c_int32, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "mirhi_comm_unique_id": (C.c_int32, [C.c_void_p]),
    "mirhi_comm_create": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]),
    "mirhi_comm_world": (C.c_uint32, [C.c_void_p]),
    "mirhi_comm_rank": (C.c_uint32, [C.c_void_p]),
    "mirhi_comm_all_gather_bands": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),
    "mirhi_comm_destroy": (C.c_int32, [C.c_void_p]),
    "mirhi_device_get_stats": (C.c_int32, [C.c_void_p, C.POINTER(DeviceStats)]),
    "mirhi_device_dispatch_path": (C.c_int32, [C.c_void_p, C.c_char_p, C.c_uint32]),
    "mirhi_device_measure_roundtrip": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]),
    "mirhi_device_set_native_dispatch": (C.c_int32, [C.c_void_p, C.c_uint32]),
    "mirhi_build_id": (C.c_char_p, []),
}


def lib():
    """Loads libmirhi.so (building it first if the sources are newer). Raises if unavailable."""
    global _lib
    if _lib is None:
        if _build.needs_build() and LIB_PATH.endswith("libmirhi.so"):
            _build.build()
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: the HIP extension must be built (python __graft_entry__.py)")
        # (a variant build named by MIRHI_LIB_NAME is loaded globally, so that libmirhost.so -- linked against libmirhi.so -- binds to IT)
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL if os.environ.get("MIRHI_LIB_NAME") else C.DEFAULT_MODE)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError = header/library mismatch: fail loudly
            fn.restype, fn.argtypes = res, args
        if L.mirhi_abi_version() != ABI_VERSION:        # struct layouts (mirhi_pipeline_desc, ...) are this binding's: a stale library must not be driven with them
            raise ImportError(f"{LIB_PATH} has ABI {L.mirhi_abi_version()}, this binding is written for ABI {ABI_VERSION}: rebuild (python __graft_entry__.py)")
        _lib = L
    return _lib


def check(rc: int):
    if rc != OK:
        L = lib()
        raise RhiError(rc, L.mirhi_last_error_message().decode("utf-8", "replace"), L.mirhi_result_name(rc).decode())


def _as_bytes(data):
    if isinstance(data, (bytes, bytearray)):
        return np.frombuffer(bytes(data), dtype=np.uint8)
    return np.ascontiguousarray(data).view(np.uint8).reshape(-1)


# ---- object wrappers (names follow crates/rhi) -----------------------------------------------------------
class Device:
    """rhi::Device (crates/rhi/src/device.rs:120-233)."""

    def __init__(self, ordinal: int = 0, stream: Optional[int] = None):
        h = C.c_void_p()
        if stream is None:
            check(lib().mirhi_device_create(ordinal, C.byref(h)))
        else:
            check(lib().mirhi_device_create_on_stream(ordinal, C.c_void_p(stream), C.byref(h)))
        self.handle = h

    @staticmethod
    def count() -> int:
        n = C.c_int32()
        check(lib().mirhi_device_count(C.byref(n)))
        return n.value

    def wait_idle(self):
        check(lib().mirhi_device_wait_idle(self.handle))

    def name(self) -> str:
        buf = C.create_string_buffer(256)
        check(lib().mirhi_device_name(self.handle, buf, 256))
        return buf.value.decode()

    def set_tile_split(self, rank: int, world: int, layout=None):
        """layout: None = keep the device's (default interleaved, MIRHI_SPLIT in the environment), SplitLayout.* or 'bands' / 'interleaved'"""
        if layout is not None:
            check(lib().mirhi_device_set_tile_split_layout(self.handle, SplitLayout.NAMES[layout] if isinstance(layout, str) else int(layout)))
        check(lib().mirhi_device_set_tile_split(self.handle, rank, world))

    def set_split_layout(self, layout):
        """SplitLayout.* or 'bands' / 'interleaved' (mirhi_device_set_tile_split_layout): before set_tile_split / Comm"""
        check(lib().mirhi_device_set_tile_split_layout(self.handle, SplitLayout.NAMES[layout] if isinstance(layout, str) else int(layout)))

    def split_rows(self, height: int):
        """(first tile row, step, number of tile rows) this device rasterizes of a frame `height` pixels high"""
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        check(lib().mirhi_device_split_rows(self.handle, height, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def set_queue_lanes(self, lanes: int):
        check(lib().mirhi_device_set_queue_lanes(self.handle, lanes))

    def set_submit_thread(self, enable: bool = True):
        """vkQueueSubmit semantics: submit() queues the work and returns, a thread of the device makes the launches (include/mirhi.h)"""
        check(lib().mirhi_device_set_submit_thread(self.handle, int(enable)))

    def band_rows(self, height: int):
        a, b = C.c_uint32(), C.c_uint32()
        check(lib().mirhi_device_band_rows(self.handle, height, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_profiling(self, enable):
        """False / 0 = off, True = Profile.TIMING, or a mask of Profile.TIMING | Profile.FRAGMENTS."""
        check(lib().mirhi_device_set_profiling(self.handle, int(enable)))

    def kernel_time(self, kernel: int):
        ms, n = C.c_double(), C.c_uint64()
        check(lib().mirhi_device_kernel_time(self.handle, kernel, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def timeline(self):
        """Every timed dispatch since the last reset: list of (kernel, lane, begin_us, end_us) on one GPU time axis."""
        n = C.c_uint32(0)
        check(lib().mirhi_device_timeline(self.handle, None, 0, C.byref(n)))
        arr = (DispatchTime * max(1, n.value))()
        check(lib().mirhi_device_timeline(self.handle, arr, n.value, C.byref(n)))
        return [(arr[i].kernel, arr[i].lane, arr[i].begin_us, arr[i].end_us) for i in range(n.value)]

    def fragment_stats(self):
        """(shaded pixels, covered fragments, scopes) accumulated under Profile.FRAGMENTS since the last reset."""
        a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        check(lib().mirhi_device_fragment_stats(self.handle, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def reset_kernel_times(self):
        check(lib().mirhi_device_reset_kernel_times(self.handle))

    def stats(self) -> DeviceStats:
        s = DeviceStats()
        check(lib().mirhi_device_get_stats(self.handle, C.byref(s)))
        return s

    def dispatch_path(self) -> str:
        """'native: ...' or 'hip: <why>' (mirhi_device_dispatch_path)."""
        buf = C.create_string_buffer(512)
        check(lib().mirhi_device_dispatch_path(self.handle, buf, 512))
        return buf.value.decode()

    def measure_roundtrip(self, lane: int = 0, reps: int = 200):
        """(empty-kernel round trip, barrier-packet round trip) in microseconds on queue lane `lane` (mirhi_device_measure_roundtrip)."""
        out = (C.c_double * 2)()
        check(lib().mirhi_device_measure_roundtrip(self.handle, lane, reps, out))
        return out[0], out[1]

    def set_native_dispatch(self, enable: bool):
        check(lib().mirhi_device_set_native_dispatch(self.handle, 1 if enable else 0))

    def submit(self, cmds: Sequence["CommandBuffer"], fence: Optional["Fence"] = None):
        """vkQueueSubmit (crates/renderer/src/renderer.rs:407-424)."""
        arr = (C.c_void_p * len(cmds))(*[c.handle for c in cmds])
        check(lib().mirhi_queue_submit(self.handle, len(cmds), arr, fence.handle if fence else None))

    def destroy(self):
        if self.handle:
            check(lib().mirhi_device_destroy(self.handle))
            self.handle = None


class Buffer:
    """rhi::Buffer (crates/rhi/src/buffer.rs:149-417)."""

    def __init__(self, device: Device, usage: int, size: int):
        h = C.c_void_p()
        check(lib().mirhi_buffer_create(device.handle, usage, size, C.byref(h)))
        self.handle, self.device = h, device

    @classmethod
    def new_with_data(cls, device: Device, usage: int, data) -> "Buffer":
        arr = _as_bytes(data)
        self = cls.__new__(cls)
        h = C.c_void_p()
        check(lib().mirhi_buffer_create_with_data(device.handle, usage, arr.ctypes.data, arr.size, C.byref(h)))
        self.handle, self.device = h, device
        return self

    @classmethod
    def wrap(cls, device: Device, usage: int, device_ptr: int, size: int) -> "Buffer":
        self = cls.__new__(cls)
        h = C.c_void_p()
        check(lib().mirhi_buffer_wrap_device_memory(device.handle, usage, C.c_void_p(device_ptr), size, C.byref(h)))
        self.handle, self.device = h, device
        return self

    def write_data(self, offset: int, data):
        arr = _as_bytes(data)
        check(lib().mirhi_buffer_write(self.handle, offset, arr.ctypes.data if arr.size else None, arr.size))

    def upload(self, data):
        self.write_data(0, data)

    def upload_via_staging(self, data):
        arr = _as_bytes(data)
        check(lib().mirhi_buffer_upload_via_staging(self.handle, arr.ctypes.data if arr.size else None, arr.size))

    def read(self, offset: int, size: int) -> np.ndarray:
        out = np.empty(size, dtype=np.uint8)
        check(lib().mirhi_buffer_read(self.handle, offset, out.ctypes.data, size))
        return out

    def size(self) -> int:
        return lib().mirhi_buffer_size(self.handle)

    def usage(self) -> int:
        return lib().mirhi_buffer_usage_of(self.handle)

    def destroy(self):
        if self.handle:
            check(lib().mirhi_buffer_destroy(self.handle))
            self.handle = None


class Image:
    """Colour target / DepthBuffer (crates/renderer/src/depth_buffer.rs:117-243) / sampled texture."""

    def __init__(self, device: Device, width: int, height: int, fmt: int, device_ptr: Optional[int] = None):
        h = C.c_void_p()
        if device_ptr is None:
            check(lib().mirhi_image_create(device.handle, width, height, fmt, C.byref(h)))
        else:
            check(lib().mirhi_image_wrap_device_memory(device.handle, width, height, fmt, C.c_void_p(device_ptr), C.byref(h)))
        self.handle, self.device, self.width, self.height, self.format = h, device, width, height, fmt

    def upload(self, data):
        arr = _as_bytes(data)
        check(lib().mirhi_image_upload(self.handle, arr.ctypes.data, arr.size))

    def generate_mips(self):
        """Full mip chain from level 0 (include/mirhi.h); the image is sampled trilinearly from then on."""
        check(lib().mirhi_image_generate_mips(self.handle))

    @property
    def mip_levels(self) -> int:
        return int(lib().mirhi_image_mip_levels(self.handle))

    def set_max_anisotropy(self, max_anisotropy: int):
        """Sampler state of a texture with a mip chain (include/mirhi.h): 1 = trilinear, up to 16 = anisotropic."""
        check(lib().mirhi_image_set_max_anisotropy(self.handle, int(max_anisotropy)))

    @property
    def max_anisotropy(self) -> int:
        return int(lib().mirhi_image_max_anisotropy(self.handle))

    def read(self) -> np.ndarray:
        n = lib().mirhi_image_size_bytes(self.handle)
        raw = np.empty(n, dtype=np.uint8)
        check(lib().mirhi_image_read(self.handle, raw.ctypes.data, n))
        if self.format == Format.R32G32B32A32_SFLOAT:
            return raw.view(np.float32).reshape(self.height, self.width, 4)
        if self.format == Format.D32_SFLOAT:
            return raw.view(np.float32).reshape(self.height, self.width)
        if self.format == Format.R32_UINT:
            return raw.view(np.uint32).reshape(self.height, self.width)
        return raw.reshape(self.height, self.width, 4)

    def destroy(self):
        if self.handle:
            check(lib().mirhi_image_destroy(self.handle))
            self.handle = None


class GraphicsPipelineBuilder:
    """crates/rhi/src/pipeline.rs:590-1059 -- same setter names, same defaults, same build() errors."""

    def __init__(self):
        self.desc = PipelineDesc()
        lib().mirhi_pipeline_desc_default(C.byref(self.desc))

    def vertex_shader(self, program: int):
        self.desc.vertex_program = program
        return self

    def fragment_shader(self, program: int):
        self.desc.fragment_program = program
        return self

    def vertex_binding(self, stride: int):
        self.desc.vertex_stride = stride
        return self

    def vertex_attributes(self, offsets: Sequence[int]):
        self.desc.attribute_count = len(offsets)
        for i, o in enumerate(offsets[:4]):
            self.desc.attribute_offsets[i] = o
        return self

    def topology(self, t: int):
        self.desc.topology = t
        return self

    def polygon_mode(self, m: int):
        self.desc.polygon_mode = m
        return self

    def cull_mode(self, m: int):
        self.desc.cull_mode = m
        return self

    def front_face(self, f: int):
        self.desc.front_face = f
        return self

    def depth_test_enable(self, e: bool):
        self.desc.depth_test_enable = int(e)
        return self

    def depth_write_enable(self, e: bool):
        self.desc.depth_write_enable = int(e)
        return self

    def depth_compare_op(self, op: int):
        self.desc.depth_compare_op = op
        return self

    def blend_enable(self, e: bool):
        self.desc.blend_enable = int(e)
        return self

    def fragment_discard_enable(self, e: bool):
        """Pipelines of alpha-masked MODEL_PBR materials (model_pbr.hlsl:176-179 `discard`): per-fragment, ordered resolve."""
        self.desc.fragment_discard_enable = int(e)
        return self

    def color_blend_attachment(self, src_color, dst_color, color_op, src_alpha, dst_alpha, alpha_op, write_mask=0xF):
        """ColorBlendAttachment (pipeline.rs:478-531) with blending enabled."""
        d = self.desc
        d.blend_enable = 1
        d.src_color_blend_factor, d.dst_color_blend_factor, d.color_blend_op = src_color, dst_color, color_op
        d.src_alpha_blend_factor, d.dst_alpha_blend_factor, d.alpha_blend_op = src_alpha, dst_alpha, alpha_op
        d.color_write_mask = write_mask
        return self

    def alpha_blend(self):
        """ColorBlendAttachment::alpha_blend() (pipeline.rs:518-529): src * src_alpha + dst * (1 - src_alpha)."""
        return self.color_blend_attachment(BlendFactor.SrcAlpha, BlendFactor.OneMinusSrcAlpha, BlendOp.Add, BlendFactor.One, BlendFactor.Zero, BlendOp.Add)

    def color_attachment_format(self, fmt: int):
        n = self.desc.color_attachment_count
        self.desc.color_attachment_formats[n] = fmt
        self.desc.color_attachment_count = n + 1
        return self

    def depth_attachment_format(self, fmt: int):
        self.desc.depth_attachment_format = fmt
        return self

    def build(self, device: Device) -> "Pipeline":
        h = C.c_void_p()
        check(lib().mirhi_pipeline_create(device.handle, C.byref(self.desc), C.byref(h)))
        return Pipeline(h)


class Pipeline:
    def __init__(self, handle):
        self.handle = handle

    def destroy(self):
        if self.handle:
            check(lib().mirhi_pipeline_destroy(self.handle))
            self.handle = None


TRIANGLE_VERTEX_STRIDE, TRIANGLE_VERTEX_OFFSETS = 24, (0, 12)        # vertex.rs:35-61
VERTEX_STRIDE, VERTEX_OFFSETS = 48, (0, 12, 24, 32)                  # vertex.rs:130-170


class CommandBuffer:
    """rhi::CommandBuffer (crates/rhi/src/command.rs:297-628)."""

    def __init__(self, device: Device):
        h = C.c_void_p()
        check(lib().mirhi_cmd_create(device.handle, C.byref(h)))
        self.handle, self.device = h, device

    def begin(self):
        check(lib().mirhi_cmd_begin(self.handle))

    def begin_reusable(self):
        check(lib().mirhi_cmd_begin_reusable(self.handle))

    def end(self):
        check(lib().mirhi_cmd_end(self.handle))

    def reset(self):
        check(lib().mirhi_cmd_reset(self.handle))

    def begin_rendering(self, color: Image, clear_color=(0.0, 0.0, 0.0, 1.0), color_load_op=LoadOp.CLEAR,
                        depth: Optional[Image] = None, clear_depth: float = 1.0, depth_load_op=LoadOp.CLEAR,
                        depth_store_op=StoreOp.DONT_CARE, prim_id: Optional[Image] = None):
        info = RenderingInfo()
        lib().mirhi_rendering_info_default(C.byref(info))
        info.color_image = color.handle
        info.color_load_op = color_load_op
        info.clear_color = (C.c_float * 4)(*clear_color)
        if depth is not None:
            info.depth_image = depth.handle
        info.depth_load_op, info.depth_store_op, info.clear_depth = depth_load_op, depth_store_op, clear_depth
        if prim_id is not None:
            info.prim_id_image = prim_id.handle
        check(lib().mirhi_cmd_begin_rendering(self.handle, C.byref(info)))

    def end_rendering(self):
        check(lib().mirhi_cmd_end_rendering(self.handle))

    def bind_pipeline(self, p: Pipeline):
        check(lib().mirhi_cmd_bind_pipeline(self.handle, p.handle))

    def bind_vertex_buffers(self, first_binding: int, buffers: Sequence[Buffer], offsets: Sequence[int]):
        arr = (C.c_void_p * len(buffers))(*[b.handle for b in buffers])
        offs = (C.c_uint64 * len(offsets))(*offsets)
        check(lib().mirhi_cmd_bind_vertex_buffers(self.handle, first_binding, len(buffers), arr, offs))

    def bind_index_buffer(self, buffer: Buffer, offset: int, index_type: int):
        check(lib().mirhi_cmd_bind_index_buffer(self.handle, buffer.handle, offset, index_type))

    def bind_uniform(self, slot: int, buffer: Buffer, offset: int = 0, range_: int = 0):
        check(lib().mirhi_cmd_bind_uniform(self.handle, slot, buffer.handle, offset, range_))

    def bind_texture(self, slot: int, image: Optional[Image]):
        check(lib().mirhi_cmd_bind_texture(self.handle, slot, image.handle if image else None))

    def set_viewport(self, x, y, width, height, min_depth=0.0, max_depth=1.0):
        vp = Viewport(x, y, width, height, min_depth, max_depth)
        check(lib().mirhi_cmd_set_viewport(self.handle, C.byref(vp)))

    def set_scissor(self, x, y, width, height):
        sc = Rect2D(x, y, width, height)
        check(lib().mirhi_cmd_set_scissor(self.handle, C.byref(sc)))

    def draw_indirect(self, buffer: Buffer, offset: int, draw_count: int, stride: int):
        check(lib().mirhi_cmd_draw_indirect(self.handle, buffer.handle, offset, draw_count, stride))

    def draw_indexed_indirect(self, buffer: Buffer, offset: int, draw_count: int, stride: int):
        check(lib().mirhi_cmd_draw_indexed_indirect(self.handle, buffer.handle, offset, draw_count, stride))

    def push_constants(self, stage_flags: int, offset: int, data: bytes):
        buf = (C.c_uint8 * len(data)).from_buffer_copy(data) if data else None
        check(lib().mirhi_cmd_push_constants(self.handle, stage_flags, offset, buf, len(data)))

    def set_queue_lane(self, lane: int):
        check(lib().mirhi_cmd_set_queue_lane(self.handle, lane))

    def draw(self, vertex_count, instance_count=1, first_vertex=0, first_instance=0):
        check(lib().mirhi_cmd_draw(self.handle, vertex_count, instance_count, first_vertex, first_instance))

    def draw_indexed(self, index_count, instance_count=1, first_index=0, vertex_offset=0, first_instance=0):
        check(lib().mirhi_cmd_draw_indexed(self.handle, index_count, instance_count, first_index, vertex_offset, first_instance))

    def destroy(self):
        if self.handle:
            check(lib().mirhi_cmd_destroy(self.handle))
            self.handle = None


class Fence:
    """rhi::Fence (crates/rhi/src/sync.rs:168-298)."""

    def __init__(self, device: Device, signaled: bool = False):
        h = C.c_void_p()
        check(lib().mirhi_fence_create(device.handle, int(signaled), C.byref(h)))
        self.handle = h

    def wait(self, timeout_ns: int = 2 ** 64 - 1):
        check(lib().mirhi_fence_wait(self.handle, timeout_ns))

    def reset(self):
        check(lib().mirhi_fence_reset(self.handle))

    def is_signaled(self) -> bool:
        return lib().mirhi_fence_status(self.handle) == OK

    def destroy(self):
        if self.handle:
            check(lib().mirhi_fence_destroy(self.handle))
            self.handle = None


class Comm:
    """Exchange of the finished bands of a tile-row split over RCCL (include/mirhi.h; no reference counterpart:
    crates/rhi/src/device.rs:61-77 drives a single VkDevice)."""

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * COMM_ID_BYTES)()
        check(lib().mirhi_comm_unique_id(buf))
        return bytes(buf)

    def __init__(self, device: Device, unique_id: bytes, rank: int, world: int):
        assert len(unique_id) == COMM_ID_BYTES
        h = C.c_void_p()
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id)
        check(lib().mirhi_comm_create(device.handle, buf, rank, world, C.byref(h)))
        self.handle, self.device = h, device

    def world(self) -> int:
        return lib().mirhi_comm_world(self.handle)

    def rank(self) -> int:
        return lib().mirhi_comm_rank(self.handle)

    def all_gather_bands(self, frame: Image, after: Optional["CommandBuffer"] = None, algo: int = GatherAlgo.DIRECT):
        check(lib().mirhi_comm_all_gather_bands(self.handle, frame.handle, after.handle if after else None, algo))

    def destroy(self):
        if self.handle:
            check(lib().mirhi_comm_destroy(self.handle))
            self.handle = None


# ---- scene helper: records a scenes.Scene the way crates/renderer records a frame ---------------------------
class SceneResources:
    """Uploads a scenes.Scene once (vertex/index/uniform buffers, textures, pipelines) and records it into
    a reusable command buffer, mirroring Renderer::create_triangle_resources + record_commands
    (crates/renderer/src/renderer.rs:205-260,452-557)."""

    def __init__(self, device: Device, scene, color_format: int = Format.R32G32B32A32_SFLOAT, want_prim: bool = False,
                 want_depth: bool = False, color_image: Optional[Image] = None, wrap_buffers=None,
                 color_load_op: int = LoadOp.CLEAR):
        self.device, self.scene = device, scene
        self.color_load_op = color_load_op
        self.owns_color = color_image is None
        self.objs = []
        self.color = color_image or Image(device, scene.width, scene.height, color_format)
        self.color_format = self.color.format
        self.prim = Image(device, scene.width, scene.height, Format.R32_UINT) if want_prim else None
        self.depth = Image(device, scene.width, scene.height, Format.D32_SFLOAT) if want_depth else None
        self.cmd = CommandBuffer(device)
        self.draw_state = []
        cache = {}

        def buf(usage, data, key=None):
            if data is None:
                return None
            key = key if key is not None else id(data)
            if (usage, key) in cache:
                return cache[(usage, key)]
            arr = _as_bytes(data)
            if arr.size == 0:
                return None
            if wrap_buffers is not None:
                b = wrap_buffers(device, usage, arr)
            else:
                b = Buffer.new_with_data(device, usage, arr)
            cache[(usage, key)] = b
            self.objs.append(b)
            return b

        def tex(t):
            if t is None:
                return None
            if id(t) in cache:
                return cache[id(t)]
            img = Image(device, t.width, t.height, Format.R8G8B8A8_SRGB if t.srgb else Format.R8G8B8A8_UNORM)
            img.upload(np.ascontiguousarray(t.rgba8))
            if t.mips:
                img.generate_mips()
            if getattr(t, "max_anisotropy", 1) > 1:
                img.set_max_anisotropy(t.max_anisotropy)
            cache[id(t)] = img
            self.objs.append(img)
            return img

        for d in scene.draws:
            model = d.program != scenes.PROGRAM_TRIANGLE
            b = (GraphicsPipelineBuilder().vertex_shader(Program.MODEL if model else Program.TRIANGLE)
                 .fragment_shader(d.program)
                 .vertex_binding(d.stride)
                 .vertex_attributes(VERTEX_OFFSETS if model else TRIANGLE_VERTEX_OFFSETS)
                 .color_attachment_format(self.color_format)
                 .cull_mode(d.cull_mode).front_face(d.front_face)
                 .depth_test_enable(d.depth_test).depth_write_enable(d.depth_write).depth_compare_op(d.depth_compare))
            if d.depth_test or d.depth_write:
                b.depth_attachment_format(Format.D32_SFLOAT)
            if getattr(d, "blend", None) is not None:
                b.color_blend_attachment(*d.blend)
            if getattr(d, "alpha_test", False):
                b.fragment_discard_enable(True)
            pipe = b.build(device)
            self.objs.append(pipe)
            st = dict(pipe=pipe, vb=buf(BufferUsage.Vertex, d.vertices),
                      ib=buf(BufferUsage.Index, d.indices) if d.indices is not None else None,
                      camera=buf(BufferUsage.Uniform, d.camera, key=("u", d.camera)),
                      object=buf(BufferUsage.Uniform, d.object, key=("u", d.object)),
                      light=buf(BufferUsage.Uniform, d.light, key=("u", d.light)),
                      material=buf(BufferUsage.Uniform, d.material, key=("u", d.material)),
                      point=buf(BufferUsage.Storage if False else BufferUsage.Uniform, d.point_lights or None, key=("u", d.point_lights)),
                      spot=buf(BufferUsage.Uniform, d.spot_lights or None, key=("u", d.spot_lights)),
                      textures=[tex(t) for t in d.textures], draw=d)
            self.draw_state.append(st)
        self.record()

    def record(self):
        s, cmd = self.scene, self.cmd
        cmd.begin_reusable()
        cmd.begin_rendering(self.color, clear_color=s.clear_color, color_load_op=self.color_load_op, depth=self.depth, clear_depth=s.clear_depth,
                            depth_store_op=StoreOp.STORE if self.depth else StoreOp.DONT_CARE, prim_id=self.prim)
        for st in self.draw_state:
            d = st["draw"]
            vp = d.viewport or (0.0, 0.0, float(s.width), float(s.height), 0.0, 1.0)
            sc = d.scissor or (0, 0, s.width, s.height)
            cmd.set_viewport(*vp)
            cmd.set_scissor(*sc)
            cmd.bind_pipeline(st["pipe"])
            cmd.bind_vertex_buffers(0, [st["vb"]], [0])
            for slot, key in ((Slot.CAMERA, "camera"), (Slot.OBJECT, "object"), (Slot.LIGHTS, "light"),
                              (Slot.MATERIAL, "material"), (Slot.POINT_LIGHTS, "point"), (Slot.SPOT_LIGHTS, "spot")):
                if st[key] is not None:
                    cmd.bind_uniform(slot, st[key])
            for slot, img in enumerate(st["textures"]):
                if img is not None or slot < 2:
                    cmd.bind_texture(slot, img)
            if st["ib"] is not None:
                cmd.bind_index_buffer(st["ib"], 0, IndexType.UINT16 if d.index_type == 2 else IndexType.UINT32)
                cmd.draw_indexed(d.count, getattr(d, "instances", 1), d.first, d.vertex_offset, 0)
            else:
                cmd.draw(d.count, getattr(d, "instances", 1), d.first, 0)
        cmd.end_rendering()
        cmd.end()

    def render(self, fence: Optional[Fence] = None):
        self.device.submit([self.cmd], fence)

    def read(self):
        self.device.wait_idle()
        out = {"color": self.color.read()}
        if self.prim:
            out["prim"] = self.prim.read()
        if self.depth:
            out["depth"] = self.depth.read()
        return out

    def destroy(self):
        self.device.wait_idle()
        self.cmd.destroy()
        for o in self.objs:
            o.destroy()
        for o in (self.prim, self.depth, self.color):
            if o is not None:
                o.destroy()
        self.objs = []
