"""glTF 2.0 -> SoA meshes, mirroring `Model::load` of the reference (crates/resources/src/model.rs:111-270):

* iterates `document.meshes()` and their primitives -- node transforms are NOT applied (model.rs:135-144)
* POSITION is required; missing NORMAL -> +Y, TEXCOORD_0 -> (0,0), TANGENT -> (1,0,0,1); missing indices ->
  sequential 0..n (model.rs:160-215)
* materials: base colour factor, metallic, roughness, emissive; AO 1.0 (model.rs:273-309, material.rs:6-30)
* images: the reference lets `gltf::import` decode them and discards the result (model.rs:120); `load(path)` does
  the same by default.  `load(path, images=True)` keeps them: every `images[i]` (file URI, data: URI or bufferView) is
  decoded to RGBA8 by images.decode_image (PNG / JPEG, host/image_decode.hpp), and each material carries the
  image index behind its baseColor / metallicRoughness / normal / occlusion / emissive texture plus alphaMode / cutoff /
  doubleSided -- what model_full.hlsl / model_pbr.hlsl bind at t0..t4.  An image whose file is absent (the dancer asset
  names three and ships one) becomes None and is listed in `Model.missing_images`.

This is SURVEY.md section 8f rank 1 (the caller side of the hot path): it feeds `interleave()` ->
`Vertex` 48 B streams (crates/rhi/src/vertex.rs:88-170) that the rasterizer consumes.
"""
from __future__ import annotations

import base64
import json
import os
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

_COMPONENT = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT2": 4, "MAT3": 9, "MAT4": 16}


class ResourceError(RuntimeError):
    """crates/resources/src/error.rs:6-40"""


@dataclass
class Material:   # crates/resources/src/material.rs:6-30
    base_color: tuple = (1.0, 1.0, 1.0, 1.0)
    metallic: float = 0.0
    roughness: float = 0.5
    ao: float = 1.0
    emissive: tuple = (0.0, 0.0, 0.0, 0.0)
    # texture slots of model_pbr.hlsl:62-95 (t0..t4) as indices into Model.images; only filled by load(images=True)
    base_color_image: Optional[int] = None
    metallic_roughness_image: Optional[int] = None
    normal_image: Optional[int] = None
    occlusion_image: Optional[int] = None
    emissive_image: Optional[int] = None
    normal_scale: float = 1.0
    occlusion_strength: float = 1.0
    alpha_mode: str = "OPAQUE"
    alpha_cutoff: float = 0.5
    double_sided: bool = False


@dataclass
class Mesh:       # crates/resources/src/model.rs:31-44
    positions: np.ndarray
    normals: np.ndarray
    tex_coords: np.ndarray
    tangents: np.ndarray
    indices: np.ndarray
    material_index: Optional[int] = None

    @property
    def vertex_count(self) -> int:
        return int(self.positions.shape[0])

    @property
    def triangle_count(self) -> int:
        return int(self.indices.size // 3)

    def interleave(self) -> np.ndarray:
        """-> (n, 12) float32 = `Vertex` 48 B: position@0 normal@12 tex_coord@24 tangent@32."""
        v = np.zeros((self.vertex_count, 12), dtype=np.float32)
        v[:, 0:3], v[:, 3:6], v[:, 6:8], v[:, 8:12] = self.positions, self.normals, self.tex_coords, self.tangents
        return v


@dataclass
class Model:
    meshes: List[Mesh] = field(default_factory=list)
    materials: List[Material] = field(default_factory=list)
    images: list = field(default_factory=list)          # images.DecodedImage or None, one per glTF image (load(images=True))
    missing_images: List[str] = field(default_factory=list)
    aabb_min: np.ndarray = None
    aabb_max: np.ndarray = None

    @property
    def total_vertices(self) -> int:
        return sum(m.vertex_count for m in self.meshes)

    @property
    def total_triangles(self) -> int:
        return sum(m.triangle_count for m in self.meshes)


def _read_accessor(doc, buffers, index) -> np.ndarray:
    acc = doc["accessors"][index]
    dtype = np.dtype(_COMPONENT[acc["componentType"]])
    ncomp = _NCOMP[acc["type"]]
    count = acc["count"]
    if "bufferView" not in acc:
        return np.zeros((count, ncomp), dtype=dtype)
    bv = doc["bufferViews"][acc["bufferView"]]
    raw = buffers[bv["buffer"]]
    start = bv.get("byteOffset", 0) + acc.get("byteOffset", 0)
    elem = dtype.itemsize * ncomp
    stride = bv.get("byteStride", 0) or elem
    if stride == elem:
        out = np.frombuffer(raw, dtype=dtype, count=count * ncomp, offset=start).reshape(count, ncomp)
    else:
        need = stride * (count - 1) + elem
        view = np.frombuffer(raw, dtype=np.uint8, count=need, offset=start)
        out = np.lib.stride_tricks.as_strided(view, shape=(count, elem), strides=(stride, 1)).copy().view(dtype).reshape(count, ncomp)
    return np.array(out)


def _image_bytes(doc, buffers, base, img):
    """-> (bytes or None when the file is absent, label)"""
    if "uri" in img:
        uri = img["uri"]
        if uri.startswith("data:"):
            return base64.b64decode(uri.split(",", 1)[1]), "data: URI"
        from urllib.parse import unquote
        full = os.path.join(base, unquote(uri))
        if not os.path.exists(full):
            return None, uri
        return open(full, "rb").read(), uri
    bv = doc["bufferViews"][img["bufferView"]]
    start = bv.get("byteOffset", 0)
    return bytes(buffers[bv["buffer"]][start:start + bv["byteLength"]]), f"bufferView {img['bufferView']}"


def _texture_image(doc, info) -> Optional[int]:
    """textureInfo -> index into images[] (through textures[].source)"""
    if not info or "index" not in info:
        return None
    tex = doc.get("textures", [])[info["index"]]
    return tex.get("source")


def load(path: str, images: bool = False) -> Model:
    """Every failure -- missing file, malformed JSON, an index or accessor that points outside its array or buffer -- is a
    ResourceError, like `gltf::import(path)?` in the reference (model.rs:113-120)."""
    try:
        return _load(path, images)
    except ResourceError:
        raise
    except Exception as e:          # IndexError / KeyError / TypeError / ValueError from a document that lies about itself
        raise ResourceError(f"Failed to load glTF {path}: {type(e).__name__}: {e}")


def _load(path: str, images: bool) -> Model:
    if not os.path.exists(path):
        raise ResourceError(f"File not found: {path}")                      # model.rs:113-115
    try:
        doc = json.load(open(path))
        base = os.path.dirname(os.path.abspath(path))
        buffers = []
        for b in doc.get("buffers", []):
            uri = b.get("uri", "")
            if uri.startswith("data:"):
                buffers.append(base64.b64decode(uri.split(",", 1)[1]))
            else:
                buffers.append(open(os.path.join(base, uri), "rb").read())
    except (OSError, ValueError, KeyError) as e:
        raise ResourceError(f"Failed to load glTF {path}: {e}")
    materials = []
    for m in doc.get("materials", []):                                       # model.rs:273-309
        pbr = m.get("pbrMetallicRoughness", {})
        em = m.get("emissiveFactor", [0.0, 0.0, 0.0])
        mat = Material(tuple(pbr.get("baseColorFactor", [1.0, 1.0, 1.0, 1.0])), float(pbr.get("metallicFactor", 1.0)),
                       float(pbr.get("roughnessFactor", 1.0)), 1.0, (em[0], em[1], em[2], 1.0))
        if images:
            mat.base_color_image = _texture_image(doc, pbr.get("baseColorTexture"))
            mat.metallic_roughness_image = _texture_image(doc, pbr.get("metallicRoughnessTexture"))
            mat.normal_image = _texture_image(doc, m.get("normalTexture"))
            mat.occlusion_image = _texture_image(doc, m.get("occlusionTexture"))
            mat.emissive_image = _texture_image(doc, m.get("emissiveTexture"))
            mat.normal_scale = float((m.get("normalTexture") or {}).get("scale", 1.0))
            mat.occlusion_strength = float((m.get("occlusionTexture") or {}).get("strength", 1.0))
            mat.alpha_mode = m.get("alphaMode", "OPAQUE")
            mat.alpha_cutoff = float(m.get("alphaCutoff", 0.5))
            mat.double_sided = bool(m.get("doubleSided", False))
        materials.append(mat)
    decoded, missing = [], []
    if images:
        from . import images as _images
        for img in doc.get("images", []):
            data, label = _image_bytes(doc, buffers, base, img)
            if data is None:
                decoded.append(None)
                missing.append(label)
                continue
            try:
                decoded.append(_images.decode_image(data))
            except _images.ImageDecodeError as e:
                raise ResourceError(f"Failed to decode image {label} of {path}: {e}")
    model = Model(materials=materials, images=decoded, missing_images=missing, aabb_min=np.full(3, np.finfo(np.float32).max, dtype=np.float32),
                  aabb_max=np.full(3, np.finfo(np.float32).min, dtype=np.float32))
    for mesh in doc.get("meshes", []):
        for prim in mesh.get("primitives", []):
            attrs = prim.get("attributes", {})
            if "POSITION" not in attrs:
                raise ResourceError("No position data")                     # model.rs:150-153
            pos = _read_accessor(doc, buffers, attrs["POSITION"]).astype(np.float32)
            n = pos.shape[0]
            if n == 0:
                continue
            nrm = _read_accessor(doc, buffers, attrs["NORMAL"]).astype(np.float32) if "NORMAL" in attrs else np.tile(np.float32([0, 1, 0]), (n, 1))
            uv = _read_accessor(doc, buffers, attrs["TEXCOORD_0"]).astype(np.float32) if "TEXCOORD_0" in attrs else np.zeros((n, 2), dtype=np.float32)
            tan = _read_accessor(doc, buffers, attrs["TANGENT"]).astype(np.float32) if "TANGENT" in attrs else np.tile(np.float32([1, 0, 0, 1]), (n, 1))
            idx = (_read_accessor(doc, buffers, prim["indices"]).reshape(-1).astype(np.uint32) if "indices" in prim
                   else np.arange(n, dtype=np.uint32))
            if idx.size and int(idx.max()) >= n:
                # an index beyond the primitive's vertices would make the GPU's vertex fetch read out of bounds (a device fault,
                # not an error code): refuse the asset here, as host/gltf.hpp does
                raise ResourceError(f"Index {int(idx.max())} out of range for a primitive with {n} vertices")
            model.aabb_min = np.minimum(model.aabb_min, pos.min(axis=0))
            model.aabb_max = np.maximum(model.aabb_max, pos.max(axis=0))
            model.meshes.append(Mesh(pos, nrm, uv, tan, idx, prim.get("material")))
    if not model.meshes:
        raise ResourceError("No meshes found in glTF file")                  # model.rs:262-264
    return model
