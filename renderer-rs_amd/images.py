"""PNG / JPEG -> RGBA8 through libmiresources.so (include/miresources.h; C++ in host/image_decode.hpp).

The texture-decode half of SURVEY.md section 8f rank 1.  The reference gets decoded images from `gltf::import` and drops
them (crates/resources/src/model.rs:120); this is the piece a caller needs to turn `assets/textures/*` or a glTF's
`images[]` into the RGBA8 texels `Image.write` / `scenes.Texture` take.  Host-only: no GPU is touched here, and there is
no Python fallback decoder -- a missing library is an error.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libmiresources.so")
SOURCES = [os.path.join(HERE, "host", "resources_capi.cpp"), os.path.join(HERE, "host", "image_decode.hpp"),
           os.path.join(HERE, "..", "include", "miresources.h")]


class ImageDecodeError(RuntimeError):
    """`.code` is the MIRES_ERR_* value."""

    def __init__(self, code: int, message: str):
        super().__init__(message)
        self.code = code


class _MiresImage(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("source_channels", C.c_uint32), ("reserved", C.c_uint32),
                ("rgba", C.POINTER(C.c_uint8))]


def build(force: bool = False) -> str:
    """g++ -> renderer-rs_amd/libmiresources.so (host code only)."""
    if not force and os.path.exists(LIB) and all(os.path.getmtime(s) <= os.path.getmtime(LIB) for s in SOURCES):
        return LIB
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-shared", "-fPIC", "-o", LIB, SOURCES[0]])
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise ImageDecodeError(2, f"{LIB} is missing: run `python -m renderer-rs_amd.build` / __graft_entry__.build()")
        l = C.CDLL(LIB)
        l.mires_image_decode.restype = C.c_int32
        l.mires_image_decode.argtypes = [C.c_char_p, C.c_uint64, C.POINTER(_MiresImage)]
        l.mires_image_load.restype = C.c_int32
        l.mires_image_load.argtypes = [C.c_char_p, C.POINTER(_MiresImage)]
        l.mires_image_free.restype = None
        l.mires_image_free.argtypes = [C.POINTER(_MiresImage)]
        l.mires_last_error_message.restype = C.c_char_p
        l.mires_last_error_message.argtypes = []
        _lib = l
    return _lib


class DecodedImage:
    """`rgba`: (h, w, 4) uint8, top row first; `source_channels`: what the file stored (1, 2, 3 or 4)."""

    def __init__(self, rgba: np.ndarray, source_channels: int):
        self.rgba, self.source_channels = rgba, source_channels

    @property
    def width(self) -> int:
        return int(self.rgba.shape[1])

    @property
    def height(self) -> int:
        return int(self.rgba.shape[0])


def _take(rc: int, img: _MiresImage) -> DecodedImage:
    l = lib()
    if rc != 0:
        raise ImageDecodeError(rc, (l.mires_last_error_message() or b"").decode("utf-8", "replace"))
    try:
        n = img.width * img.height * 4
        arr = np.ctypeslib.as_array(img.rgba, shape=(n,)).copy().reshape(img.height, img.width, 4)
        return DecodedImage(arr, int(img.source_channels))
    finally:
        l.mires_image_free(C.byref(img))


def decode_image(data: bytes) -> DecodedImage:
    img = _MiresImage()
    return _take(lib().mires_image_decode(bytes(data), len(data), C.byref(img)), img)


def load_image(path: str) -> DecodedImage:
    img = _MiresImage()
    return _take(lib().mires_image_load(os.fsencode(path), C.byref(img)), img)
