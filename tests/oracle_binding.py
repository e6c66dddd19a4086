"""ctypes binding of oracle/libmirhi_oracle.so -- the CPU parity oracle (test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libmirhi_oracle.so")


class OracleTexture(C.Structure):
    _fields_ = [("rgba8", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32), ("levels", C.c_uint32), ("srgb", C.c_uint32),
                ("max_anisotropy", C.c_uint32)]


class OracleDraw(C.Structure):
    _fields_ = [
        ("vertex_data", C.c_void_p), ("vertex_stride", C.c_uint32),
        ("index_data", C.c_void_p), ("index_type", C.c_uint32), ("count", C.c_uint32), ("first", C.c_uint32),
        ("vertex_offset", C.c_int32),
        ("program", C.c_uint32), ("cull_mode", C.c_uint32), ("front_face", C.c_uint32),
        ("depth_test", C.c_uint32), ("depth_write", C.c_uint32), ("depth_compare", C.c_uint32),
        ("viewport", C.c_float * 6), ("scissor", C.c_int32 * 4),
        ("camera", C.c_void_p), ("object", C.c_void_p), ("light_ubo", C.c_void_p), ("material", C.c_void_p),
        ("point_lights", C.c_void_p), ("spot_lights", C.c_void_p),
        ("albedo_map", OracleTexture), ("normal_map", OracleTexture),
        ("metallic_roughness_map", OracleTexture), ("occlusion_map", OracleTexture), ("emissive_map", OracleTexture),
        ("blend_enable", C.c_uint32), ("src_color_factor", C.c_uint32), ("dst_color_factor", C.c_uint32), ("color_op", C.c_uint32),
        ("src_alpha_factor", C.c_uint32), ("dst_alpha_factor", C.c_uint32), ("alpha_op", C.c_uint32), ("color_write_mask", C.c_uint32),
    ]


class OraclePass(C.Structure):
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32), ("clear_color", C.c_float * 4), ("clear_depth", C.c_float),
        ("num_draws", C.c_uint32), ("draws", C.POINTER(OracleDraw)), ("row_begin", C.c_uint32), ("row_end", C.c_uint32),
    ]


_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(ORACLE_DIR, "mirhi_oracle.c")
    if force or not os.path.exists(LIB_PATH) or (
            os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(LIB_PATH)):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-B", "libmirhi_oracle.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.oracle_render.restype = C.c_int
        L.oracle_render.argtypes = [C.POINTER(OraclePass), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_srgb8.restype = C.c_uint8
        L.oracle_srgb8.argtypes = [C.c_float]
        L.oracle_attenuation.restype = C.c_float
        L.oracle_attenuation.argtypes = [C.c_float, C.c_float]
        L.oracle_distribution_ggx.restype = C.c_float
        L.oracle_distribution_ggx.argtypes = [C.c_float, C.c_float]
        L.oracle_geometry_schlick_ggx.restype = C.c_float
        L.oracle_geometry_schlick_ggx.argtypes = [C.c_float, C.c_float]
        L.oracle_roughness_to_shininess.restype = C.c_float
        L.oracle_roughness_to_shininess.argtypes = [C.c_float]
        L.oracle_glam_determinant.restype = C.c_float
        _lib = L
    return _lib


def _buf(b):
    """bytes / ndarray -> (keepalive, address or None)."""
    if b is None:
        return None, None
    if isinstance(b, (bytes, bytearray)):
        if len(b) == 0:
            return None, None
        arr = np.frombuffer(bytes(b), dtype=np.uint8).copy()
    else:
        arr = np.ascontiguousarray(b)
    return arr, arr.ctypes.data


def render(scene, nthreads: int = 1, want_bgra8: bool = True, rows=None):
    """Renders a scenes.Scene with the oracle. Returns dict(rgba, prim, depth, bgra8)."""
    L = lib()
    keep = []
    # an instanced draw is its primitives once per instance, in instance order (the path has no per-instance input: command.rs:583-628
    # passes instance_count through, binding 0 is per-vertex and no program reads SV_InstanceID)
    expanded = [d for d in scene.draws for _ in range(max(0, int(getattr(d, "instances", 1))))]
    draws = (OracleDraw * max(1, len(expanded)))()
    for i, d in enumerate(expanded):
        od = draws[i]
        vb = d.vertex_bytes()
        keep.append(vb)
        od.vertex_data = vb.ctypes.data
        od.vertex_stride = d.stride
        if d.indices is not None:
            ib = np.ascontiguousarray(d.indices)
            keep.append(ib)
            od.index_data = ib.ctypes.data
        od.index_type = d.index_type
        od.count, od.first, od.vertex_offset = d.count, d.first, d.vertex_offset
        od.program, od.cull_mode, od.front_face = d.program, d.cull_mode, d.front_face
        od.depth_test, od.depth_write, od.depth_compare = int(d.depth_test), int(d.depth_write), d.depth_compare
        if getattr(d, "blend", None) is not None:
            od.blend_enable = 1
            (od.src_color_factor, od.dst_color_factor, od.color_op, od.src_alpha_factor, od.dst_alpha_factor, od.alpha_op,
             od.color_write_mask) = d.blend
        vp = d.viewport or (0.0, 0.0, float(scene.width), float(scene.height), 0.0, 1.0)
        sc = d.scissor or (0, 0, scene.width, scene.height)
        od.viewport = (C.c_float * 6)(*vp)
        od.scissor = (C.c_int32 * 4)(*sc)
        for name, src in (("camera", d.camera), ("object", d.object), ("light_ubo", d.light), ("material", d.material),
                          ("point_lights", d.point_lights), ("spot_lights", d.spot_lights)):
            k, addr = _buf(src)
            keep.append(k)
            setattr(od, name, addr)
        for name, tex in (("albedo_map", d.albedo_map), ("normal_map", d.normal_map),
                          ("metallic_roughness_map", d.metallic_roughness_map), ("occlusion_map", d.occlusion_map),
                          ("emissive_map", d.emissive_map)):
            if tex is not None:
                levels = 1
                arr = np.ascontiguousarray(tex.rgba8)
                if getattr(tex, "mips", False):        # the chain is built here, by numpy, independently of the HIP mip kernel
                    from renderer_rs_amd.scenes import mip_chain
                    chain = mip_chain(tex.rgba8)
                    levels = len(chain)
                    arr = np.concatenate([l.reshape(-1) for l in chain])
                keep.append(arr)
                t = OracleTexture(arr.ctypes.data, tex.width, tex.height, levels, int(getattr(tex, "srgb", False)),
                                  int(getattr(tex, "max_anisotropy", 1)))
                setattr(od, name, t)
    p = OraclePass()
    p.width, p.height = scene.width, scene.height
    p.clear_color = (C.c_float * 4)(*scene.clear_color)
    p.clear_depth = scene.clear_depth
    p.num_draws = len(expanded)
    p.draws = draws
    if rows is not None:
        p.row_begin, p.row_end = rows
    n = scene.width * scene.height
    rgba = np.empty((scene.height, scene.width, 4), dtype=np.float32)
    prim = np.empty((scene.height, scene.width), dtype=np.uint32)
    depth = np.empty((scene.height, scene.width), dtype=np.float32)
    bgra8 = np.empty((scene.height, scene.width, 4), dtype=np.uint8) if want_bgra8 else None
    rc = L.oracle_render(C.byref(p), nthreads, rgba.ctypes.data, prim.ctypes.data, depth.ctypes.data,
                         bgra8.ctypes.data if want_bgra8 else None)
    if rc != 0:
        raise RuntimeError(f"oracle_render failed: {rc}")
    del n
    return {"rgba": rgba, "prim": prim, "depth": depth, "bgra8": bgra8}


def mat_fn(name, *args):
    """Calls an oracle_glam_* function whose last parameter is float out[16]."""
    L = lib()
    out = (C.c_float * 16)()
    cargs = []
    for a in args:
        if isinstance(a, (float, int)):
            cargs.append(C.c_float(a))
        else:
            arr = np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
            cargs.append((C.c_float * arr.size)(*arr.tolist()))
    getattr(L, name)(*cargs, out)
    return np.array(out[:], dtype=np.float32).reshape(4, 4)
