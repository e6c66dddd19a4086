"""ctypes binding of libmirhost.so (include/mirhost.h): the reference's draw-submit loop -- Renderer::render_frame over
FrameManager, crates/renderer/src/renderer.rs:367-557, frame_manager.rs:299-539 -- running natively (C++ over the C ABI), every
frame re-recorded.  The loop's resources (pipelines, buffers, images) are the caller's mirhi objects."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Sequence

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libmirhost.so")
SLOT_COUNT, TEXTURE_COUNT = 6, 5


class _Viewport(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("width", C.c_float), ("height", C.c_float), ("min_depth", C.c_float), ("max_depth", C.c_float)]


class _Rect2D(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("width", C.c_uint32), ("height", C.c_uint32)]


class _Uniform(C.Structure):
    _fields_ = [("buffer", C.c_void_p), ("offset", C.c_uint64), ("range", C.c_uint64)]


class HostDraw(C.Structure):          # mirhost_draw
    _fields_ = [("pipeline", C.c_void_p), ("vertex_buffer", C.c_void_p), ("vertex_offset_bytes", C.c_uint64),
                ("index_buffer", C.c_void_p), ("index_offset_bytes", C.c_uint64), ("index_type", C.c_int32),
                ("uniforms", _Uniform * SLOT_COUNT), ("textures", C.c_void_p * TEXTURE_COUNT),
                ("viewport", _Viewport), ("scissor", _Rect2D),
                ("count", C.c_uint32), ("instance_count", C.c_uint32), ("first", C.c_uint32), ("vertex_offset", C.c_int32)]


class FrameDesc(C.Structure):         # mirhost_frame_desc
    _fields_ = [("frames_in_flight", C.c_uint32), ("image_count", C.c_uint32), ("images", C.POINTER(C.c_void_p)), ("depth", C.c_void_p),
                ("clear_color", C.c_float * 4), ("clear_depth", C.c_float), ("draw_count", C.c_uint32), ("draws", C.POINTER(HostDraw)),
                ("vary_triangles", C.c_uint32), ("submit_thread", C.c_uint32),
                ("per_frame_uniform", C.c_uint32), ("reserved", C.c_uint32)]


_lib = None


def build(force: bool = False) -> str:
    """g++ -> renderer-rs_amd/libmirhost.so (host code only; links libmirhi.so)."""
    srcs = [os.path.join(HERE, "host", n) for n in ("frame_loop.cpp", "mirhi.hpp")] + [os.path.join(os.path.dirname(HERE), "include", n) for n in ("mirhost.h", "mirhi.h")]
    stale = not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs if os.path.exists(s))
    if force or stale:
        subprocess.check_call(["make", "-C", os.path.join(HERE, "host"), "../libmirhost.so"] + (["-B"] if force else []))
    return LIB


def lib():
    global _lib
    if _lib is None:
        from . import lib as mirhi_lib
        mirhi_lib()                                   # libmirhi.so first (libmirhost.so links it)
        if not os.path.exists(LIB):
            build()
        L = C.CDLL(LIB)
        L.mirhost_frame_loop_create.restype, L.mirhost_frame_loop_create.argtypes = C.c_int32, [C.c_void_p, C.POINTER(FrameDesc), C.POINTER(C.c_void_p)]
        L.mirhost_frame_loop_run.restype, L.mirhost_frame_loop_run.argtypes = C.c_int32, [C.c_void_p, C.c_uint64, C.POINTER(C.c_double)]
        L.mirhost_frame_loop_last_image.restype, L.mirhost_frame_loop_last_image.argtypes = C.c_int32, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
        L.mirhost_frame_loop_phase_seconds.restype, L.mirhost_frame_loop_phase_seconds.argtypes = C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(C.c_double)]
        L.mirhost_frame_loop_destroy.restype, L.mirhost_frame_loop_destroy.argtypes = C.c_int32, [C.c_void_p]
        L.mirhost_last_error_message.restype, L.mirhost_last_error_message.argtypes = C.c_char_p, []
        _lib = L
    return _lib


def _check(rc: int):
    if rc != 0:
        from . import RhiError, lib as mirhi_lib
        raise RhiError(rc, lib().mirhost_last_error_message().decode("utf-8", "replace"), mirhi_lib().mirhi_result_name(rc).decode())


class FrameLoop:
    """Renderer::render_frame in a native loop.  `resources`: a SceneResources (its pipelines / buffers / textures and draw list are
    what every frame records); `images`: the colour targets cycled as swapchain images (frames_in_flight + 1 in the reference)."""

    def __init__(self, device, resources, images: Sequence, frames_in_flight: int = 2, depth=None, vary_triangles: int = 0, submit_thread: bool = False,
                 per_frame_uniform=None):
        from . import IndexType, Slot
        s = resources.scene
        n = len(resources.draw_state)
        self._draws = (HostDraw * max(1, n))()
        for i, st in enumerate(resources.draw_state):
            d, hd = st["draw"], self._draws[i]
            hd.pipeline = st["pipe"].handle
            hd.vertex_buffer, hd.vertex_offset_bytes = st["vb"].handle, 0
            if st["ib"] is not None:
                hd.index_buffer, hd.index_offset_bytes = st["ib"].handle, 0
                hd.index_type = IndexType.UINT16 if d.index_type == 2 else IndexType.UINT32
            for slot, key in ((Slot.CAMERA, "camera"), (Slot.OBJECT, "object"), (Slot.LIGHTS, "light"), (Slot.MATERIAL, "material"),
                              (Slot.POINT_LIGHTS, "point"), (Slot.SPOT_LIGHTS, "spot")):
                if st[key] is not None:
                    hd.uniforms[slot].buffer = st[key].handle
            for t, img in enumerate(st["textures"]):
                if img is not None and t < TEXTURE_COUNT:
                    hd.textures[t] = img.handle
            vp = d.viewport or (0.0, 0.0, float(s.width), float(s.height), 0.0, 1.0)
            sc = d.scissor or (0, 0, s.width, s.height)
            hd.viewport = _Viewport(*vp)
            hd.scissor = _Rect2D(*sc)
            hd.count, hd.instance_count, hd.first, hd.vertex_offset = d.count, getattr(d, "instances", 1), d.first, d.vertex_offset
        self._images = (C.c_void_p * len(images))(*[im.handle for im in images])
        desc = FrameDesc()
        desc.frames_in_flight, desc.image_count, desc.images = frames_in_flight, len(images), self._images
        desc.depth = depth.handle if depth is not None else None
        desc.clear_color = (C.c_float * 4)(*s.clear_color)
        desc.clear_depth = s.clear_depth
        desc.draw_count, desc.draws = n, self._draws
        desc.vary_triangles = vary_triangles
        desc.submit_thread = int(submit_thread)
        desc.per_frame_uniform = 0 if per_frame_uniform is None else int(per_frame_uniform) + 1     # a Slot: one copy per frame in flight, rewritten every frame
        h = C.c_void_p()
        _check(lib().mirhost_frame_loop_create(device.handle, C.byref(desc), C.byref(h)))
        self.handle, self.images = h, list(images)

    def run(self, frames: int) -> float:
        """renders `frames` frames and waits for them; returns the wall time of the call in seconds"""
        sec = C.c_double()
        _check(lib().mirhost_frame_loop_run(self.handle, frames, C.byref(sec)))
        return sec.value

    def phase_seconds(self, enable: bool = True):
        """host seconds per phase since the last call: (fence wait, recording, end(), submit); switches the accounting on / off"""
        out = (C.c_double * 4)()
        _check(lib().mirhost_frame_loop_phase_seconds(self.handle, int(enable), out))
        return tuple(out)

    def last_image(self):
        idx, n = C.c_uint32(), C.c_uint64()
        _check(lib().mirhost_frame_loop_last_image(self.handle, C.byref(idx), C.byref(n)))
        return self.images[idx.value], n.value

    def destroy(self):
        if self.handle:
            _check(lib().mirhost_frame_loop_destroy(self.handle))
            self.handle = None
