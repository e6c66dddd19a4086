"""Synthetic scenes for the BASELINE.json configs and small parity cases.

Host-side data only: vertex/index streams in the reference's layouts
(`TriangleVertex` 24 B / `Vertex` 48 B, crates/rhi/src/vertex.rs:20-61,88-170; u32 indices,
crates/resources/src/model.rs:41) and uniform blocks in the HLSL layouts
(shaders/hlsl/vertex/model.hlsl:5-19, shaders/hlsl/lights.hlsli:17-55,
shaders/hlsl/pixel/model_full.hlsl:34-41).  Nothing here touches a GPU or the oracle.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

PROGRAM_TRIANGLE, PROGRAM_MODEL, PROGRAM_MODEL_FULL, PROGRAM_MODEL_PBR = 0, 1, 2, 3
CULL_NONE, CULL_FRONT, CULL_BACK, CULL_FRONT_AND_BACK = 0, 1, 2, 3
FRONT_CCW, FRONT_CW = 0, 1
CMP_NEVER, CMP_LESS, CMP_EQUAL, CMP_LESS_OR_EQUAL, CMP_GREATER, CMP_NOT_EQUAL, CMP_GREATER_OR_EQUAL, CMP_ALWAYS = range(8)
# crates/rhi/src/pipeline.rs:411-448 BlendFactor, :452-476 BlendOp (reference enum order)
(BF_ZERO, BF_ONE, BF_SRC_COLOR, BF_ONE_MINUS_SRC_COLOR, BF_DST_COLOR, BF_ONE_MINUS_DST_COLOR, BF_SRC_ALPHA, BF_ONE_MINUS_SRC_ALPHA,
 BF_DST_ALPHA, BF_ONE_MINUS_DST_ALPHA, BF_CONSTANT_COLOR, BF_ONE_MINUS_CONSTANT_COLOR, BF_CONSTANT_ALPHA, BF_ONE_MINUS_CONSTANT_ALPHA,
 BF_SRC_ALPHA_SATURATE) = range(15)
BO_ADD, BO_SUBTRACT, BO_REVERSE_SUBTRACT, BO_MIN, BO_MAX = range(5)
# ColorBlendAttachment::alpha_blend() (pipeline.rs:518-529): (src colour, dst colour, colour op, src alpha, dst alpha, alpha op, write mask)
ALPHA_BLEND = (BF_SRC_ALPHA, BF_ONE_MINUS_SRC_ALPHA, BO_ADD, BF_ONE, BF_ZERO, BO_ADD, 0xF)

f32 = np.float32


# ------------------------------------------------------------------------------------------------
# PCG32 (O'Neill, XSH-RR 64/32) -- the seeded generator SURVEY 8d names for the synthetic inputs
# ------------------------------------------------------------------------------------------------
class PCG32:
    MULT = 6364136223846793005
    MASK = (1 << 64) - 1

    def __init__(self, seed: int, seq: int = 0xDA3E39CB94B95BDB):
        self.inc = ((seq << 1) | 1) & self.MASK
        self.state = 0
        self.next_u32()
        self.state = (self.state + seed) & self.MASK
        self.next_u32()

    def next_u32(self) -> int:
        old = self.state
        self.state = (old * self.MULT + self.inc) & self.MASK
        xorshifted = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        return ((xorshifted >> rot) | (xorshifted << ((-rot) & 31))) & 0xFFFFFFFF

    def uniform(self, n: int) -> np.ndarray:
        """n float64 values uniform in [0,1) with 32 random bits each."""
        out = np.empty(n, dtype=np.float64)
        for i in range(n):
            out[i] = self.next_u32() * (1.0 / 4294967296.0)
        return out


# ------------------------------------------------------------------------------------------------
# glam-style float32 matrix helpers (column-major, m[col, row] flattened col-major to 16 floats)
# restating crates/scene/src/camera.rs:110-142 and crates/resources/src/ubo.rs:109-117,243-259
# ------------------------------------------------------------------------------------------------
def _v(x):
    return np.asarray(x, dtype=f32)


def _normalize(v):
    v = _v(v)
    return (v / f32(math.sqrt(float(np.dot(v, v))))).astype(f32)


def perspective_rh(fovy, aspect, near, far) -> np.ndarray:
    s, c = f32(math.sin(0.5 * fovy)), f32(math.cos(0.5 * fovy))
    h = f32(c / s)
    w = f32(h / f32(aspect))
    r = f32(f32(far) / f32(f32(near) - f32(far)))
    m = np.zeros((4, 4), dtype=f32)  # m[col][row]
    m[0, 0] = w
    m[1, 1] = h
    m[2, 2] = r
    m[2, 3] = -1.0
    m[3, 2] = f32(r * f32(near))
    return m


def look_at_rh(eye, center, up) -> np.ndarray:
    eye, center, up = _v(eye), _v(center), _v(up)
    f = _normalize(center - eye)
    s = _normalize(np.cross(f, up).astype(f32))
    u = np.cross(s, f).astype(f32)
    m = np.zeros((4, 4), dtype=f32)
    m[0] = [s[0], u[0], -f[0], 0]
    m[1] = [s[1], u[1], -f[1], 0]
    m[2] = [s[2], u[2], -f[2], 0]
    m[3] = [-np.dot(eye, s), -np.dot(eye, u), np.dot(eye, f), 1]
    return m


def mat_mul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """glam a * b for column-major [col,row] arrays."""
    return (b.astype(f32) @ a.astype(f32)).astype(f32)


def projection_vulkan(fovy, aspect, near, far) -> np.ndarray:
    p = perspective_rh(fovy, aspect, near, far)
    p[1, 1] = f32(-p[1, 1])  # camera.rs:135 Vulkan Y flip
    return p


def trs(scale, quat, translation) -> np.ndarray:
    x, y, z, w = [f32(q) for q in quat]
    x2, y2, z2 = x + x, y + y, z + z
    xx, xy, xz, yy, yz, zz = x * x2, x * y2, x * z2, y * y2, y * z2, z * z2
    wx, wy, wz = w * x2, w * y2, w * z2
    s = _v(scale)
    m = np.zeros((4, 4), dtype=f32)
    m[0] = [(1 - (yy + zz)) * s[0], (xy + wz) * s[0], (xz - wy) * s[0], 0]
    m[1] = [(xy - wz) * s[1], (1 - (xx + zz)) * s[1], (yz + wx) * s[1], 0]
    m[2] = [(xz + wy) * s[2], (yz - wx) * s[2], (1 - (xx + yy)) * s[2], 0]
    m[3] = [translation[0], translation[1], translation[2], 1]
    return m


def quat_axis_angle(axis, angle) -> np.ndarray:
    a = _normalize(axis)
    s = math.sin(0.5 * angle)
    return np.array([a[0] * s, a[1] * s, a[2] * s, math.cos(0.5 * angle)], dtype=f32)


def normal_matrix(model: np.ndarray) -> np.ndarray:
    m64 = model.astype(np.float64).T  # row-major maths matrix
    det = np.linalg.det(m64)
    if abs(det) < 1e-6:
        return np.eye(4, dtype=f32)
    inv_t = np.linalg.inv(m64).T  # maths matrix (row-major) of inverse-transpose
    return inv_t.T.astype(f32)  # back to [col,row]


def camera_ubo(view: np.ndarray, proj: np.ndarray, eye) -> bytes:
    """CameraData 208 B: view@0 projection@64 viewProjection@128 cameraPosition@192 (ubo.rs:64-117)."""
    vp = mat_mul(proj, view)
    return (view.astype(f32).tobytes() + proj.astype(f32).tobytes() + vp.tobytes()
            + _v(eye).tobytes() + f32(0).tobytes())


def object_ubo(model: np.ndarray) -> bytes:
    """ObjectData 128 B: model@0 normalMatrix@64 (ubo.rs:174-259)."""
    return model.astype(f32).tobytes() + normal_matrix(model).tobytes()


def light_ubo(direction=(0.0, -1.0, 0.0), intensity=0.0, color=(1.0, 1.0, 1.0), num_point=0, num_spot=0) -> bytes:
    """LightUBO 48 B in the HLSL layout (lights.hlsli:17-23,49-55): dir, intensity, colour, pad, counts."""
    return (_v(direction).tobytes() + f32(intensity).tobytes() + _v(color).tobytes() + f32(0).tobytes()
            + np.array([num_point, num_spot], dtype=np.uint32).tobytes() + np.zeros(2, dtype=f32).tobytes())


def material_ubo(base_color=(1.0, 1.0, 1.0, 1.0), metallic=0.0, roughness=0.5, ao=1.0) -> bytes:
    """MaterialData 32 B (model_full.hlsl:34-41; defaults crates/resources/src/material.rs:20-30)."""
    return _v(base_color).tobytes() + np.array([metallic, roughness, ao, 0.0], dtype=f32).tobytes()


def pbr_material_ubo(base_color=(1.0, 1.0, 1.0, 1.0), metallic=0.0, roughness=0.5, ao=1.0, normal_scale=1.0,
                     emissive=(0.0, 0.0, 0.0), alpha_cutoff=0.0, has_base_color=False, has_normal=False,
                     has_metallic_roughness=False, has_occlusion=False, has_emissive=False) -> bytes:
    """MaterialData 80 B of the Cook-Torrance program (pixel/model_pbr.hlsl:36-59)."""
    out = (_v(base_color).tobytes() + np.array([metallic, roughness, ao, normal_scale], dtype=f32).tobytes()
           + _v(emissive).tobytes() + f32(alpha_cutoff).tobytes()
           + np.array([has_base_color, has_normal, has_metallic_roughness, has_occlusion, has_emissive, 0, 0, 0],
                      dtype=np.int32).tobytes())
    assert len(out) == 80
    return out


def point_light(position, radius, color, intensity) -> bytes:
    """PointLight 32 B (lights.hlsli:27-33 == crates/scene/src/light.rs:31-42)."""
    return _v(position).tobytes() + f32(radius).tobytes() + _v(color).tobytes() + f32(intensity).tobytes()


def spot_light(position, inner_cos, direction, outer_cos, color, intensity) -> bytes:
    """SpotLight 48 B in the HLSL layout (lights.hlsli:37-45)."""
    return (_v(position).tobytes() + f32(inner_cos).tobytes() + _v(direction).tobytes() + f32(outer_cos).tobytes()
            + _v(color).tobytes() + f32(intensity).tobytes())


# ------------------------------------------------------------------------------------------------
# scene description shared by the oracle binding (tests/) and the product binding
# ------------------------------------------------------------------------------------------------
@dataclass
class Texture:
    rgba8: np.ndarray  # (h, w, 4) uint8
    mips: bool = False  # sampled trilinearly through a full mip chain (mirhi_image_generate_mips / mip_chain below)
    srgb: bool = False  # R8G8B8A8_SRGB: the RGB bytes are decoded to linear when sampled
    max_anisotropy: int = 1  # > 1 (at most 16, needs mips): anisotropic filtering (mirhi_image_set_max_anisotropy)

    @property
    def width(self):
        return int(self.rgba8.shape[1])

    @property
    def height(self):
        return int(self.rgba8.shape[0])


def mip_chain(rgba8: np.ndarray) -> List[np.ndarray]:
    """Full mip chain of an (h, w, 4) uint8 image, level 0 first: 2x2 box filter on the stored bytes, round half up, edge clamp
    for odd sizes -- the specification `mirhi_image_generate_mips` and the oracle's callers follow (include/mirhi.h)."""
    levels = [np.ascontiguousarray(rgba8, dtype=np.uint8)]
    while levels[-1].shape[0] > 1 or levels[-1].shape[1] > 1:
        src = levels[-1].astype(np.uint32)
        h, w = src.shape[:2]
        dh, dw = max(1, h >> 1), max(1, w >> 1)
        y0 = np.minimum(2 * np.arange(dh), h - 1); y1 = np.minimum(2 * np.arange(dh) + 1, h - 1)
        x0 = np.minimum(2 * np.arange(dw), w - 1); x1 = np.minimum(2 * np.arange(dw) + 1, w - 1)
        acc = src[y0][:, x0] + src[y0][:, x1] + src[y1][:, x0] + src[y1][:, x1]
        levels.append(((acc + 2) >> 2).astype(np.uint8))
    return levels


@dataclass
class DrawSpec:
    vertices: np.ndarray                  # uint8 bytes or structured float32 (n, stride/4)
    stride: int
    count: int                            # vertex_count or index_count
    indices: Optional[np.ndarray] = None  # uint16 / uint32
    first: int = 0
    vertex_offset: int = 0
    program: int = PROGRAM_TRIANGLE
    cull_mode: int = CULL_BACK            # pipeline.rs:661 default
    front_face: int = FRONT_CCW           # pipeline.rs:662 default
    depth_test: bool = True               # pipeline.rs:677-679 defaults
    depth_write: bool = True
    depth_compare: int = CMP_LESS
    viewport: Optional[tuple] = None      # (x, y, w, h, min_depth, max_depth); None = full extent, 0..1
    scissor: Optional[tuple] = None       # (x, y, w, h); None = full extent
    camera: Optional[bytes] = None
    object: Optional[bytes] = None
    light: Optional[bytes] = None
    material: Optional[bytes] = None
    point_lights: bytes = b""
    spot_lights: bytes = b""
    albedo_map: Optional[Texture] = None
    normal_map: Optional[Texture] = None
    metallic_roughness_map: Optional[Texture] = None   # MODEL_PBR only (model_pbr.hlsl:62-95 t2..t4)
    occlusion_map: Optional[Texture] = None
    emissive_map: Optional[Texture] = None
    alpha_test: bool = False              # pipeline fragment_discard_enable: alpha-masked MODEL_PBR material (per-fragment discard)
    instances: int = 1                    # instance_count of the draw call: the same primitives again, instance after instance
    blend: Optional[tuple] = None         # None = opaque; else (src colour, dst colour, colour op, src alpha, dst alpha, alpha op, write mask)

    @property
    def textures(self):
        return (self.albedo_map, self.normal_map, self.metallic_roughness_map, self.occlusion_map, self.emissive_map)

    @property
    def index_type(self) -> int:
        if self.indices is None:
            return 0
        return 2 if self.indices.dtype == np.uint16 else 4

    @property
    def num_triangles(self) -> int:
        return (self.count // 3) * self.instances

    def vertex_bytes(self) -> np.ndarray:
        return np.ascontiguousarray(self.vertices).view(np.uint8).reshape(-1)


@dataclass
class Scene:
    name: str
    width: int
    height: int
    draws: List[DrawSpec] = field(default_factory=list)
    clear_color: tuple = (0.0, 0.0, 0.0, 1.0)  # rendering.rs:102-115 default
    clear_depth: float = 1.0                   # rendering.rs:356-370 default

    @property
    def num_triangles(self) -> int:
        return sum(d.num_triangles for d in self.draws)

    def algorithmic_bytes(self, bpp_out: int = 4) -> int:
        """Nv*stride + Ni*4 + W*H*bpp_out (+ texture bytes), summed over draws."""
        total = self.width * self.height * bpp_out
        seen = set()
        for d in self.draws:
            key = id(d.vertices)
            if key not in seen:
                seen.add(key)
                total += d.vertex_bytes().size
            if d.indices is not None and id(d.indices) not in seen:
                seen.add(id(d.indices))
                total += d.indices.size * d.indices.dtype.itemsize
            for t in d.textures:
                if t is not None and id(t) not in seen and t.rgba8.size > 4:
                    seen.add(id(t))
                    total += t.rgba8.size
        return total


# ------------------------------------------------------------------------------------------------
# C1: hello triangle (crates/renderer/src/renderer.rs:228-246,479-518)
# ------------------------------------------------------------------------------------------------
def hello_triangle(width: int = 256, height: int = 256) -> Scene:
    verts = np.array([
        [0.0, -0.5, 0.0, 1.0, 0.0, 0.0],   # top - red
        [-0.5, 0.5, 0.0, 0.0, 1.0, 0.0],   # bottom-left - green
        [0.5, 0.5, 0.0, 0.0, 0.0, 1.0],    # bottom-right - blue
    ], dtype=f32)
    d = DrawSpec(vertices=verts, stride=24, count=3, program=PROGRAM_TRIANGLE, cull_mode=CULL_NONE,
                 depth_test=False, depth_write=False)
    return Scene("C1-hello-triangle", width, height, [d], clear_color=(0.1, 0.1, 0.15, 1.0))


# ------------------------------------------------------------------------------------------------
# C2: N random flat-shaded triangles
# ------------------------------------------------------------------------------------------------
def random_triangles(n: int = 10000, width: int = 1920, height: int = 1080, seed: int = 0x5EED0002,
                     rmin: float = 4.0, rmax: float = 48.0) -> Scene:
    rng = PCG32(seed)
    u = rng.uniform(n * 10).reshape(n, 10)
    cx = u[:, 0] * 2.0 - 1.0
    cy = u[:, 1] * 2.0 - 1.0
    z = 0.05 + 0.9 * u[:, 2]
    r = rmin + (rmax - rmin) * u[:, 3]
    ang = u[:, 4:7] * (2.0 * math.pi)
    col = u[:, 7:10]
    verts = np.empty((n, 3, 6), dtype=f32)
    for k in range(3):
        verts[:, k, 0] = cx + r * np.cos(ang[:, k]) * (2.0 / width)
        verts[:, k, 1] = cy + r * np.sin(ang[:, k]) * (2.0 / height)
        verts[:, k, 2] = z
        verts[:, k, 3:6] = col
    d = DrawSpec(vertices=verts.reshape(n * 3, 6), stride=24, count=3 * n, program=PROGRAM_TRIANGLE,
                 cull_mode=CULL_NONE, depth_test=True, depth_write=True, depth_compare=CMP_LESS)
    return Scene(f"C2-random-{n}", width, height, [d], clear_color=(0.1, 0.1, 0.15, 1.0))


# ------------------------------------------------------------------------------------------------
# seeded value noise on a lattice (periodic in u), used by the C3/C4 stand-ins
# ------------------------------------------------------------------------------------------------
def _value_noise(u: np.ndarray, v: np.ndarray, seed: int, cells_u: int, cells_v: int, periodic_u: bool) -> np.ndarray:
    rng = PCG32(seed)
    lat = rng.uniform((cells_u + 1) * (cells_v + 1)).reshape(cells_v + 1, cells_u + 1)
    if periodic_u:
        lat[:, cells_u] = lat[:, 0]
    fu, fv = u * cells_u, v * cells_v
    iu = np.clip(np.floor(fu).astype(np.int64), 0, cells_u - 1)
    iv = np.clip(np.floor(fv).astype(np.int64), 0, cells_v - 1)
    au, av = fu - iu, fv - iv
    au = au * au * (3 - 2 * au)
    av = av * av * (3 - 2 * av)
    a = lat[iv, iu] * (1 - au) + lat[iv, iu + 1] * au
    b = lat[iv + 1, iu] * (1 - au) + lat[iv + 1, iu + 1] * au
    return a * (1 - av) + b * av


def _grid_indices(nu: int, nv: int) -> np.ndarray:
    """Two CCW (as seen from +normal with u to the right, v up) triangles per quad; (nu+1) verts per row."""
    j, i = np.meshgrid(np.arange(nv, dtype=np.uint32), np.arange(nu, dtype=np.uint32), indexing="ij")
    v00 = j * (nu + 1) + i
    v10 = v00 + 1
    v01 = v00 + (nu + 1)
    v11 = v01 + 1
    tris = np.stack([v00, v10, v11, v00, v11, v01], axis=-1)
    return tris.reshape(-1).astype(np.uint32)


def _smooth_normals(pos: np.ndarray, idx: np.ndarray) -> np.ndarray:
    p = pos.astype(np.float64)
    t = idx.reshape(-1, 3)
    fn = np.cross(p[t[:, 1]] - p[t[:, 0]], p[t[:, 2]] - p[t[:, 0]])
    n = np.zeros_like(p)
    for k in range(3):
        np.add.at(n, t[:, k], fn)
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    ln[ln == 0] = 1.0
    return n / ln


def _pack_vertex48(pos, nrm, uv, tan) -> np.ndarray:
    n = pos.shape[0]
    v = np.zeros((n, 12), dtype=f32)
    v[:, 0:3] = pos
    v[:, 3:6] = nrm
    v[:, 6:8] = uv
    v[:, 8:12] = tan
    return v


def default_camera(width: int, height: int, eye=(0.0, 0.0, 5.0), target=(0.0, 0.0, 0.0), fov_deg: float = 45.0):
    """Reference default camera (camera.rs:43-56): fov 45 deg, near 0.1, far 1000; aspect from the target."""
    view = look_at_rh(eye, target, (0.0, 1.0, 0.0))
    proj = projection_vulkan(math.radians(fov_deg), width / height, 0.1, 1000.0)
    return view, proj, camera_ubo(view, proj, eye)


WHITE_1X1 = Texture(np.full((1, 1, 4), 255, dtype=np.uint8))


# ------------------------------------------------------------------------------------------------
# C3: "bunny" stand-in -- displaced UV sphere, Phong + 1 point light
# ------------------------------------------------------------------------------------------------
def displaced_sphere(nu: int = 188, nv: int = 187, width: int = 1920, height: int = 1080,
                     seed: int = 0x5EED0003, program: int = PROGRAM_MODEL_FULL) -> Scene:
    uu, vv = np.meshgrid(np.linspace(0.0, 1.0, nu + 1), np.linspace(0.0, 1.0, nv + 1), indexing="xy")
    uu, vv = uu.reshape(-1), vv.reshape(-1)
    theta = uu * 2.0 * math.pi
    phi = (0.02 + 0.96 * vv) * math.pi  # keep a small hole at the poles: no degenerate fans
    rad = 1.0 + 0.12 * (_value_noise(uu, vv, seed, 12, 8, True) - 0.5) * 2.0
    pos = np.stack([rad * np.sin(phi) * np.cos(theta), rad * np.cos(phi), rad * np.sin(phi) * np.sin(theta)], axis=1)
    idx = _grid_indices(nu, nv)
    # orientation: make triangles CCW seen from outside
    t = idx.reshape(-1, 3)
    fn = np.cross(pos[t[0, 1]] - pos[t[0, 0]], pos[t[0, 2]] - pos[t[0, 0]])
    if np.dot(fn, pos[t[0, 0]]) < 0:
        idx = t[:, [0, 2, 1]].reshape(-1).astype(np.uint32)
    nrm = _smooth_normals(pos, idx)
    tan = np.stack([-np.sin(theta), np.zeros_like(theta), np.cos(theta), np.ones_like(theta)], axis=1)
    verts = _pack_vertex48(pos, nrm, np.stack([uu, vv], axis=1), tan)
    # camera: reference default eye (0,0,5) looking at the origin; sphere radius ~1 spans ~60 % of height
    view, proj, cam = default_camera(width, height, eye=(0.0, 0.0, 4.0))
    model = trs((1.0, 1.0, 1.0), quat_axis_angle((0.0, 1.0, 0.0), 0.6), (0.0, 0.0, 0.0))
    d = DrawSpec(vertices=verts, stride=48, count=idx.size, indices=idx, program=program,
                 cull_mode=CULL_BACK, front_face=FRONT_CCW,
                 camera=cam, object=object_ubo(model),
                 light=light_ubo(direction=(0.0, -1.0, 0.0), intensity=0.0, num_point=1),
                 material=material_ubo((0.7, 0.7, 0.7, 1.0), 0.0, 0.5, 1.0),
                 point_lights=point_light((2.0, 2.0, 2.0), 10.0, (1.0, 1.0, 1.0), 5.0),
                 albedo_map=WHITE_1X1, normal_map=WHITE_1X1)
    return Scene(f"C3-sphere-{idx.size // 3}", width, height, [d], clear_color=(0.1, 0.1, 0.15, 1.0))


# ------------------------------------------------------------------------------------------------
# C4: 1M-triangle height-field grid, fallback Blinn-Phong
# ------------------------------------------------------------------------------------------------
def heightfield_grid(nu: int = 1000, nv: int = 500, width: int = 3840, height: int = 2160,
                     seed: int = 0x5EED0004) -> Scene:
    uu, vv = np.meshgrid(np.linspace(0.0, 1.0, nu + 1), np.linspace(0.0, 1.0, nv + 1), indexing="xy")
    uu, vv = uu.reshape(-1), vv.reshape(-1)
    h = 0.15 * (_value_noise(uu, vv, seed, 40, 20, False) - 0.5)
    aspect = width / height
    pos = np.stack([(uu - 0.5) * 2.0 * aspect * 1.9, (vv - 0.5) * 2.0 * 1.9, h], axis=1)
    idx = _grid_indices(nu, nv)
    nrm = _smooth_normals(pos, idx)
    tan = np.tile(np.array([1.0, 0.0, 0.0, 1.0]), (pos.shape[0], 1))
    verts = _pack_vertex48(pos, nrm, np.stack([uu, vv], axis=1), tan)
    view, proj, cam = default_camera(width, height, eye=(0.0, 0.0, 5.0))
    model = trs((1.0, 1.0, 1.0), quat_axis_angle((1.0, 0.0, 0.0), -math.radians(30.0)), (0.0, 0.0, 0.0))
    d = DrawSpec(vertices=verts, stride=48, count=idx.size, indices=idx, program=PROGRAM_MODEL,
                 cull_mode=CULL_BACK, front_face=FRONT_CCW, camera=cam, object=object_ubo(model))
    return Scene(f"C4-grid-{idx.size // 3}", width, height, [d], clear_color=(0.1, 0.1, 0.15, 1.0))


# ------------------------------------------------------------------------------------------------
# C5: "Sponza" stand-in -- boxes in a hall, 4 point lights, 4 procedural textures
# ------------------------------------------------------------------------------------------------
def _procedural_texture(kind: int, size: int, seed: int) -> Texture:
    rng = PCG32(seed + kind)
    base = (rng.uniform(3) * 0.5 + 0.4)
    y, x = np.meshgrid(np.arange(size), np.arange(size), indexing="ij")
    if kind % 2 == 0:   # checker
        m = (((x // (size // 16)) + (y // (size // 16))) & 1).astype(np.float64)
        val = 0.55 + 0.45 * m
    else:               # brick
        row = y // (size // 16)
        xo = (x + (row & 1) * (size // 16)) % (size // 8)
        mortar = ((y % (size // 16)) < 3) | (xo < 3)
        val = np.where(mortar, 0.35, 1.0)
    img = np.zeros((size, size, 4), dtype=np.uint8)
    for c in range(3):
        img[:, :, c] = np.clip(val * base[c] * 255.0 + 0.5, 0, 255).astype(np.uint8)
    img[:, :, 3] = 255
    return Texture(img)


def _box_mesh(center, half, sides=(8, 7), caps=(4, 4)):
    """Axis-aligned box, 4 side faces sides[0]*sides[1] quads + 2 caps caps[0]*caps[1] quads = 512 tris."""
    cx, cy, cz = center
    hx, hy, hz = half
    faces = [  # origin corner, u axis, v axis (u x v = outward normal), subdivision
        ((cx - hx, cy - hy, cz + hz), (2 * hx, 0, 0), (0, 2 * hy, 0), sides),   # +Z
        ((cx + hx, cy - hy, cz - hz), (-2 * hx, 0, 0), (0, 2 * hy, 0), sides),  # -Z
        ((cx + hx, cy - hy, cz + hz), (0, 0, -2 * hz), (0, 2 * hy, 0), sides),  # +X
        ((cx - hx, cy - hy, cz - hz), (0, 0, 2 * hz), (0, 2 * hy, 0), sides),   # -X
        ((cx - hx, cy + hy, cz + hz), (2 * hx, 0, 0), (0, 0, -2 * hz), caps),   # +Y
        ((cx - hx, cy - hy, cz - hz), (2 * hx, 0, 0), (0, 0, 2 * hz), caps),    # -Y
    ]
    vs, ids, base = [], [], 0
    for o, du, dv, (nu, nv) in faces:
        o, du, dv = np.array(o), np.array(du), np.array(dv)
        uu, vv = np.meshgrid(np.linspace(0, 1, nu + 1), np.linspace(0, 1, nv + 1), indexing="xy")
        uu, vv = uu.reshape(-1), vv.reshape(-1)
        pos = o[None, :] + uu[:, None] * du[None, :] + vv[:, None] * dv[None, :]
        n = np.cross(du, dv)
        n = n / np.linalg.norm(n)
        t = du / np.linalg.norm(du)
        nrm = np.tile(n, (pos.shape[0], 1))
        tan = np.tile(np.array([t[0], t[1], t[2], 1.0]), (pos.shape[0], 1))
        scale = max(np.linalg.norm(du), np.linalg.norm(dv))
        uv = np.stack([uu * np.linalg.norm(du) / scale * 2.0, vv * np.linalg.norm(dv) / scale * 2.0], axis=1)
        vs.append(_pack_vertex48(pos, nrm, uv, tan))
        ids.append(_grid_indices(nu, nv) + base)
        base += pos.shape[0]
    return np.concatenate(vs, axis=0), np.concatenate(ids).astype(np.uint32)


def box_hall(n_boxes: int = 512, width: int = 3840, height: int = 2160, seed: int = 0x5EED0005,
             tex_size: int = 1024) -> Scene:
    rng = PCG32(seed)
    u = rng.uniform(n_boxes * 6).reshape(n_boxes, 6)
    textures = [_procedural_texture(k, tex_size, seed) for k in range(4)]
    eye = (0.0, 2.5, 14.0)
    view, proj, cam = default_camera(width, height, eye=eye, target=(0.0, 2.0, 0.0), fov_deg=60.0)
    lights = b"".join([
        point_light((-8.0, 6.0, 4.0), 30.0, (1.0, 0.9, 0.8), 40.0),
        point_light((8.0, 6.0, 4.0), 30.0, (0.8, 0.9, 1.0), 40.0),
        point_light((0.0, 8.0, -4.0), 30.0, (1.0, 1.0, 1.0), 40.0),
        point_light((0.0, 2.0, 10.0), 30.0, (1.0, 1.0, 0.9), 30.0),
    ])
    model = np.eye(4, dtype=f32)
    obj = object_ubo(model)
    groups = [([], [], 0) for _ in range(4)]
    groups = [{"v": [], "i": [], "base": 0} for _ in range(4)]
    for b in range(n_boxes):
        c = ((u[b, 0] - 0.5) * 30.0, u[b, 1] * 10.0, (u[b, 2] - 0.5) * 15.0 - 2.0)
        hsz = (0.25 + 0.6 * u[b, 3], 0.25 + 0.9 * u[b, 4], 0.25 + 0.6 * u[b, 5])
        v, i = _box_mesh(c, hsz)
        g = groups[b % 4]
        g["v"].append(v)
        g["i"].append(i + g["base"])
        g["base"] += v.shape[0]
    draws = []
    for k, g in enumerate(groups):
        if not g["v"]:
            continue
        verts = np.concatenate(g["v"], axis=0)
        idx = np.concatenate(g["i"]).astype(np.uint32)
        draws.append(DrawSpec(vertices=verts, stride=48, count=idx.size, indices=idx, program=PROGRAM_MODEL_FULL,
                              cull_mode=CULL_BACK, front_face=FRONT_CCW, camera=cam, object=obj,
                              light=light_ubo(direction=(0.3, -1.0, 0.2), intensity=0.15, num_point=4),
                              material=material_ubo((1.0, 1.0, 1.0, 1.0), 0.0, 0.35 + 0.15 * k, 1.0),
                              point_lights=lights, albedo_map=textures[k], normal_map=WHITE_1X1))
    return Scene(f"C5-boxhall-{sum(d.num_triangles for d in draws)}", width, height, draws,
                 clear_color=(0.02, 0.02, 0.03, 1.0))


# ------------------------------------------------------------------------------------------------
# small parity cases (edge cases the domain has: clipping, culling, shared edges, depth ties, scissor)
# ------------------------------------------------------------------------------------------------
def _tri_verts(pts, cols=None) -> np.ndarray:
    pts = np.asarray(pts, dtype=f32).reshape(-1, 3)
    if cols is None:
        cols = np.tile(np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1]], dtype=f32), (pts.shape[0] // 3 + 1, 1))[:pts.shape[0]]
    return np.concatenate([pts, np.asarray(cols, dtype=f32).reshape(-1, 3)], axis=1)


def shared_edge_fan(width: int = 160, height: int = 120, n: int = 24, seed: int = 7) -> Scene:
    """A fan of thin triangles sharing edges and a centre: every pixel must be covered exactly once
    (top-left rule), including along shared edges; no depth test so double hits would be visible."""
    rng = PCG32(seed)
    ang = np.sort(rng.uniform(n)) * 2 * math.pi
    pts, cols = [], []
    col = rng.uniform(3 * n).reshape(n, 3)
    for k in range(n):
        a0, a1 = ang[k], ang[(k + 1) % n]
        pts += [(0.013, -0.021, 0.5), (0.9 * math.cos(a0), 0.9 * math.sin(a0), 0.5), (0.9 * math.cos(a1), 0.9 * math.sin(a1), 0.5)]
        cols += [col[k]] * 3
    d = DrawSpec(vertices=_tri_verts(pts, cols), stride=24, count=3 * n, cull_mode=CULL_NONE, depth_test=False,
                 depth_write=False)
    return Scene("shared-edge-fan", width, height, [d])


def near_clip_case(width: int = 200, height: int = 150) -> Scene:
    """A ground quad through the camera: crosses the near plane and w=0 (exercises real clipping)."""
    q = 20.0
    pos = np.array([[-q, 0, q], [q, 0, q], [q, 0, -q], [-q, 0, -q]], dtype=f32)
    nrm = np.tile(np.array([0, 1, 0], dtype=f32), (4, 1))
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], dtype=f32)
    tan = np.tile(np.array([1, 0, 0, 1], dtype=f32), (4, 1))
    verts = _pack_vertex48(pos, nrm, uv, tan)
    idx = np.array([0, 1, 2, 0, 2, 3], dtype=np.uint16)
    view, proj, cam = default_camera(width, height, eye=(0.0, 1.0, 3.0), target=(0.0, 0.5, 0.0))
    d = DrawSpec(vertices=verts, stride=48, count=6, indices=idx, program=PROGRAM_MODEL, cull_mode=CULL_NONE,
                 camera=cam, object=object_ubo(np.eye(4, dtype=f32)))
    return Scene("near-clip-quad", width, height, [d], clear_color=(0.1, 0.1, 0.15, 1.0))


def depth_tie_case(width: int = 96, height: int = 64) -> Scene:
    """Coplanar overlapping triangles at identical depth: LESS keeps the earlier primitive."""
    pts = [(-0.8, -0.8, 0.5), (0.8, -0.8, 0.5), (0.0, 0.8, 0.5),
           (-0.8, 0.8, 0.5), (0.0, -0.8, 0.5), (0.8, 0.8, 0.5),
           (-0.5, -0.5, 0.25), (0.5, -0.5, 0.75), (0.0, 0.5, 0.5)]
    cols = [(1, 0, 0)] * 3 + [(0, 1, 0)] * 3 + [(0, 0, 1)] * 3
    d = DrawSpec(vertices=_tri_verts(pts, cols), stride=24, count=9, cull_mode=CULL_NONE)
    return Scene("depth-tie", width, height, [d])


def cull_scissor_case(width: int = 128, height: int = 96) -> Scene:
    """Back-face culling with default CCW front face + a scissor/viewport smaller than the target."""
    pts = [(-0.9, -0.9, 0.3), (-0.1, -0.9, 0.3), (-0.5, 0.9, 0.3),      # one winding
           (0.1, -0.9, 0.4), (0.5, 0.9, 0.4), (0.9, -0.9, 0.4)]         # the opposite winding
    d = DrawSpec(vertices=_tri_verts(pts), stride=24, count=6, cull_mode=CULL_BACK, front_face=FRONT_CCW,
                 viewport=(8.0, 4.0, 100.0, 80.0, 0.0, 1.0), scissor=(16, 8, 90, 60))
    return Scene("cull-scissor", width, height, [d], clear_color=(0.2, 0.3, 0.4, 1.0))


def multi_draw_case(width: int = 160, height: int = 128) -> Scene:
    """Two draws with different programs in one rendering scope + first_vertex / vertex_offset use."""
    s1 = random_triangles(40, width, height, seed=11, rmin=6, rmax=30).draws[0]
    s1.first = 6
    s1.count = 3 * 36
    sph = displaced_sphere(12, 9, width, height, seed=5, program=PROGRAM_MODEL).draws[0]
    pad = np.zeros((5, 12), dtype=f32)
    sph.vertices = np.concatenate([pad, sph.vertices], axis=0)
    sph.vertex_offset = 5
    sph.first = 3
    sph.count -= 6
    return Scene("multi-draw", width, height, [s1, sph], clear_color=(0.05, 0.05, 0.05, 1.0))


def huge_triangle_case(width: int = 192, height: int = 108) -> Scene:
    """Triangles far larger than the guard band and off-screen ones (guard-band clipping, trivial reject)."""
    pts = [(-300.0, -200.0, 0.5), (300.0, -250.0, 0.6), (10.0, 500.0, 0.4),
           (2.0, 2.0, 0.5), (3.0, 2.0, 0.5), (2.0, 3.0, 0.5),
           (-0.2, -0.2, 0.1), (0.3, -0.1, 0.1), (0.0, 0.4, 0.1)]
    d = DrawSpec(vertices=_tri_verts(pts), stride=24, count=9, cull_mode=CULL_NONE)
    return Scene("huge-triangle", width, height, [d])


def textured_quad_case(width: int = 160, height: int = 120) -> Scene:
    """model_full with a real albedo texture, a tangent-space normal map, a spot light and a point light."""
    pos = np.array([[-1.5, -1, 0], [1.5, -1, 0], [1.5, 1, 0], [-1.5, 1, 0]], dtype=f32)
    nrm = np.tile(np.array([0, 0, 1], dtype=f32), (4, 1))
    uv = np.array([[0, 0], [3, 0], [3, 2], [0, 2]], dtype=f32)
    tan = np.tile(np.array([1, 0, 0, 1], dtype=f32), (4, 1))
    verts = _pack_vertex48(pos, nrm, uv, tan)
    idx = np.array([0, 1, 2, 0, 2, 3], dtype=np.uint32)
    albedo = _procedural_texture(1, 64, 99)
    rng = PCG32(123)
    nm = np.zeros((16, 16, 4), dtype=np.uint8)
    r = rng.uniform(16 * 16 * 2).reshape(16, 16, 2)
    nm[:, :, 0] = (128 + (r[:, :, 0] - 0.5) * 80).astype(np.uint8)
    nm[:, :, 1] = (128 + (r[:, :, 1] - 0.5) * 80).astype(np.uint8)
    nm[:, :, 2] = 230
    nm[:, :, 3] = 255
    view, proj, cam = default_camera(width, height, eye=(0.5, 0.3, 3.0))
    model = trs((1.0, 1.2, 1.0), quat_axis_angle((0.0, 1.0, 0.0), 0.5), (0.1, 0.0, 0.0))
    d = DrawSpec(vertices=verts, stride=48, count=6, indices=idx, program=PROGRAM_MODEL_FULL, cull_mode=CULL_NONE,
                 camera=cam, object=object_ubo(model),
                 light=light_ubo(direction=(0.2, -0.5, -1.0), intensity=0.4, color=(1.0, 0.95, 0.9), num_point=1, num_spot=1),
                 material=material_ubo((0.9, 0.8, 0.7, 0.5), 0.0, 0.3, 0.8),
                 point_lights=point_light((1.0, 1.0, 2.0), 8.0, (0.6, 0.7, 1.0), 4.0),
                 spot_lights=spot_light((-1.0, 0.5, 2.5), 0.95, (0.3, -0.1, -1.0), 0.8, (1.0, 0.8, 0.6), 6.0),
                 albedo_map=albedo, normal_map=Texture(nm))
    return Scene("textured-quad", width, height, [d], clear_color=(0.0, 0.0, 0.0, 1.0))


def alpha_mask_case(width: int = 200, height: int = 140) -> Scene:
    """Alpha-masked Cook-Torrance materials (`if (baseColor.a < alphaCutoff) discard;`, pixel/model_pbr.hlsl:176-179; glTF alphaMode
    MASK): an opaque back wall, in front of it a tilted cut-out quad whose texture alpha runs through every value (a fragment is
    kept or dropped one by one, and a dropped one leaves colour, depth and id to whatever lies behind), a second cut-out that is
    also alpha-blended, and a masked quad BEHIND the wall (its kept fragments fail the depth test).  The masked pipelines set
    fragment_discard_enable, so their segments are resolved in primitive order."""
    rng = PCG32(0xA1FA)
    view, proj, cam = default_camera(width, height, eye=(0.3, 0.2, 3.4))
    light = light_ubo(direction=(0.2, -0.6, -0.8), intensity=1.4, color=(1.0, 0.97, 0.92), num_point=1)
    points = point_light((1.0, 1.5, 2.0), 9.0, (0.6, 0.8, 1.0), 5.0)
    n = 32
    yy, xx = np.mgrid[0:n, 0:n]
    leaf = np.zeros((n, n, 4), dtype=np.uint8)
    leaf[..., 0] = 40 + 3 * xx; leaf[..., 1] = 120 + 4 * yy; leaf[..., 2] = 30; leaf[..., 3] = ((xx * 37 + yy * 91 + (xx * yy) % 7 * 29) % 256).astype(np.uint8)
    disc = np.zeros((n, n, 4), dtype=np.uint8)
    rr = np.hypot(xx - 15.5, yy - 15.5)
    disc[..., 0] = 220; disc[..., 1] = 150 + (r8 := (rng.uniform(n * n).reshape(n, n) * 60).astype(np.uint8)); disc[..., 2] = 60
    disc[..., 3] = np.clip(255 - rr * 18, 0, 255).astype(np.uint8)
    wall_tex = _procedural_texture(3, 32, 5)
    quad = lambda w, h, uvs: _pack_vertex48(np.array([[-w, -h, 0], [w, -h, 0], [w, h, 0], [-w, h, 0]], dtype=f32),
                                            np.tile(np.array([0, 0, 1], dtype=f32), (4, 1)),
                                            np.array([[0, 0], [uvs, 0], [uvs, uvs], [0, uvs]], dtype=f32), np.tile(np.array([1, 0, 0, 1], dtype=f32), (4, 1)))
    idx = np.array([0, 1, 2, 0, 2, 3], dtype=np.uint32)
    common = dict(stride=48, count=6, indices=idx, program=PROGRAM_MODEL_PBR, cull_mode=CULL_NONE, camera=cam, light=light, point_lights=points)
    draws = [
        DrawSpec(vertices=quad(2.2, 1.6, 2.0), object=object_ubo(trs((1, 1, 1), quat_axis_angle((0, 1, 0), 0.15), (0.0, 0.0, -0.8))),
                 material=pbr_material_ubo((0.9, 0.9, 0.9, 1.0), 0.1, 0.6, 1.0, has_base_color=True), albedo_map=wall_tex, **common),
        DrawSpec(vertices=quad(1.1, 0.9, 2.5), object=object_ubo(trs((1, 1, 1), quat_axis_angle((0.2, 1, 0.1), 0.7), (-0.5, 0.1, 0.3))),
                 material=pbr_material_ubo((1.0, 1.0, 1.0, 0.9), 0.0, 0.5, 1.0, alpha_cutoff=0.45, has_base_color=True), albedo_map=Texture(leaf),
                 alpha_test=True, **common),
        DrawSpec(vertices=quad(0.8, 0.8, 1.0), object=object_ubo(trs((1, 1, 1), quat_axis_angle((1, 0, 0), -0.4), (0.9, -0.2, 0.6))),
                 material=pbr_material_ubo((1.0, 1.0, 1.0, 1.0), 0.3, 0.4, 1.0, alpha_cutoff=0.25, has_base_color=True), albedo_map=Texture(disc),
                 alpha_test=True, blend=ALPHA_BLEND, **common),
        DrawSpec(vertices=quad(1.5, 1.0, 3.0), object=object_ubo(trs((1, 1, 1), quat_axis_angle((0, 1, 0), 0.0), (0.2, 0.0, -2.0))),
                 material=pbr_material_ubo((0.2, 0.3, 1.0, 1.0), 0.0, 0.5, 1.0, alpha_cutoff=0.5, has_base_color=True), albedo_map=Texture(leaf),
                 alpha_test=True, **common),
    ]
    return Scene("alpha-mask", width, height, draws, clear_color=(0.05, 0.06, 0.1, 1.0))


def pbr_spheres_case(width: int = 224, height: int = 144) -> Scene:
    """Cook-Torrance program (pixel/model_pbr.hlsl): five draws covering every material switch -- factors only,
    all five textures, metallic, a constant-alpha draw the cutoff removes, and a textured draw whose cutoff is
    below every possible texel alpha."""
    rng = PCG32(0xB0B)
    view, proj, cam = default_camera(width, height, eye=(0.0, 0.4, 4.2))
    light = light_ubo(direction=(0.3, -0.8, -0.5), intensity=1.6, color=(1.0, 0.96, 0.9), num_point=2, num_spot=1)
    points = point_light((1.5, 1.2, 2.0), 9.0, (0.5, 0.7, 1.0), 6.0) + point_light((-2.0, 0.3, 1.5), 7.0, (1.0, 0.5, 0.3), 4.0)
    spots = spot_light((0.0, 2.5, 2.5), 0.93, (0.0, -0.7, -0.7), 0.75, (0.9, 1.0, 0.8), 9.0)
    base = _procedural_texture(2, 32, 7)
    nm = np.zeros((16, 16, 4), dtype=np.uint8)
    r = rng.uniform(16 * 16 * 2).reshape(16, 16, 2)
    nm[:, :, 0] = (128 + (r[:, :, 0] - 0.5) * 90).astype(np.uint8)
    nm[:, :, 1] = (128 + (r[:, :, 1] - 0.5) * 90).astype(np.uint8)
    nm[:, :, 2] = 225
    nm[:, :, 3] = 255
    mr = (rng.uniform(8 * 8 * 4).reshape(8, 8, 4) * 255).astype(np.uint8)
    occ = (128 + rng.uniform(8 * 8 * 4).reshape(8, 8, 4) * 127).astype(np.uint8)
    emi = (rng.uniform(4 * 4 * 4).reshape(4, 4, 4) * 255).astype(np.uint8)
    sphere = displaced_sphere(20, 14, width, height, seed=11).draws[0]
    mats = [
        dict(material=pbr_material_ubo((0.8, 0.3, 0.2, 1.0), 0.0, 0.45, 0.9, emissive=(0.02, 0.0, 0.05))),
        dict(material=pbr_material_ubo((1.0, 0.9, 0.8, 0.7), 0.9, 0.8, 1.0, normal_scale=0.7, emissive=(0.5, 0.4, 0.1),
                                       has_base_color=True, has_normal=True, has_metallic_roughness=True,
                                       has_occlusion=True, has_emissive=True),
             albedo_map=base, normal_map=Texture(nm), metallic_roughness_map=Texture(mr), occlusion_map=Texture(occ),
             emissive_map=Texture(emi)),
        dict(material=pbr_material_ubo((0.95, 0.8, 0.4, 1.0), 1.0, 0.02, 1.0)),                       # roughness clamp 0.04
        dict(material=pbr_material_ubo((0.1, 0.9, 0.1, 0.3), 0.0, 0.5, 1.0, alpha_cutoff=0.5)),      # dropped by the cutoff
        dict(material=pbr_material_ubo((0.3, 0.4, 0.9, 0.6), 0.2, 0.6, 0.7, alpha_cutoff=-0.25, has_base_color=True),
             albedo_map=base),
    ]
    draws = []
    for k, extra in enumerate(mats):
        x = -2.4 + 1.2 * k
        model = trs((0.55, 0.55, 0.55), quat_axis_angle((0.3, 1.0, 0.1), 0.4 * k), (x, 0.25 * (k % 2), -0.3 * k))
        draws.append(DrawSpec(vertices=sphere.vertices, stride=48, count=sphere.count, indices=sphere.indices,
                              program=PROGRAM_MODEL_PBR, cull_mode=CULL_BACK, front_face=sphere.front_face, camera=cam,
                              object=object_ubo(model), light=light, point_lights=points, spot_lights=spots, **extra))
    return Scene("pbr-spheres", width, height, draws, clear_color=(0.02, 0.02, 0.03, 1.0))


def mip_ground_case(width: int = 240, height: int = 150, max_anisotropy: int = 1) -> Scene:
    """Texture fidelity: a ground plane receding to the horizon under a fine checker -- strong, anisotropic
    minification -- through a mip chain with trilinear filtering and an sRGB-encoded albedo, plus an upright quad that is
    magnified (lambda clamps to 0).  A non-power-of-two normal map exercises the odd-size rule of the chain.
    max_anisotropy > 1: the same frame through the anisotropic filter (albedo at max_anisotropy, normal map at a quarter of it)."""
    rng = PCG32(0x717)
    n = 64
    yy, xx = np.mgrid[0:n, 0:n]
    checker = (((xx // 2) + (yy // 2)) & 1).astype(np.uint8)
    alb = np.zeros((n, n, 4), dtype=np.uint8)
    alb[..., 0] = 40 + 190 * checker; alb[..., 1] = 60 + 150 * (1 - checker); alb[..., 2] = 90 + (xx * 2).astype(np.uint8); alb[..., 3] = 255
    nm = np.zeros((24, 40, 4), dtype=np.uint8)
    r = rng.uniform(24 * 40 * 2).reshape(24, 40, 2)
    nm[..., 0] = (128 + (r[..., 0] - 0.5) * 70).astype(np.uint8); nm[..., 1] = (128 + (r[..., 1] - 0.5) * 70).astype(np.uint8)
    nm[..., 2] = 235; nm[..., 3] = 255
    albedo = Texture(alb, mips=True, srgb=True, max_anisotropy=max_anisotropy)
    normal = Texture(nm, mips=True, max_anisotropy=max(1, max_anisotropy // 4))
    view, proj, cam = default_camera(width, height, eye=(0.0, 1.2, 4.0), target=(0.0, 0.3, 0.0))
    light = light_ubo(direction=(0.2, -1.0, -0.3), intensity=0.9, color=(1.0, 0.97, 0.9), num_point=1)
    points = point_light((1.0, 2.0, 1.5), 12.0, (0.8, 0.9, 1.0), 5.0)
    ground = _pack_vertex48(np.array([[-30, 0, 6], [30, 0, 6], [30, 0, -120], [-30, 0, -120]], dtype=f32),
                            np.tile(np.array([0, 1, 0], dtype=f32), (4, 1)),
                            np.array([[0, 0], [40, 0], [40, 84], [0, 84]], dtype=f32), np.tile(np.array([1, 0, 0, 1], dtype=f32), (4, 1)))
    wall = _pack_vertex48(np.array([[-0.6, 0.0, 1.5], [0.6, 0.0, 1.5], [0.6, 1.2, 1.5], [-0.6, 1.2, 1.5]], dtype=f32),
                          np.tile(np.array([0, 0, 1], dtype=f32), (4, 1)),
                          np.array([[0.1, 0.1], [0.3, 0.1], [0.3, 0.3], [0.1, 0.3]], dtype=f32), np.tile(np.array([1, 0, 0, 1], dtype=f32), (4, 1)))
    idx = np.array([0, 1, 2, 0, 2, 3], dtype=np.uint32)
    common = dict(stride=48, count=6, indices=idx, program=PROGRAM_MODEL_FULL, cull_mode=CULL_NONE, camera=cam,
                  object=object_ubo(np.eye(4, dtype=f32)), light=light, point_lights=points,
                  material=material_ubo((1.0, 1.0, 1.0, 1.0), 0.0, 0.6, 0.9), albedo_map=albedo, normal_map=normal)
    return Scene("mip-ground" if max_anisotropy <= 1 else f"aniso{max_anisotropy}-ground", width, height, [DrawSpec(vertices=ground, **common), DrawSpec(vertices=wall, **common)],
                 clear_color=(0.3, 0.5, 0.8, 1.0))


def gltf_model(path: str, width: int = 1920, height: int = 1080, program: int = PROGRAM_MODEL_FULL,
               eye=(0.0, 0.0, 3.2), yaw: float = 0.4, textures: bool = False) -> Scene:
    """A glTF asset through the reference's loader semantics (gltf.load) -> `Vertex` streams, lit like config 3:
    1 point light, base colour 0.7, roughness 0.5, white 1x1 albedo / "no normal map" textures.

    textures=True binds the asset's own images where its files exist (gltf.load(images=True)): base colour as
    R8G8B8A8_SRGB, normal / metallic-roughness / occlusion as UNORM, all with a mip chain and trilinear sampling
    (model_pbr.hlsl:62-95 slot order); the material's factors replace the config-3 constants."""
    from . import gltf
    model = gltf.load(path, images=textures)
    view, proj, cam = default_camera(width, height, eye=eye)
    obj = object_ubo(trs((1.0, 1.0, 1.0), quat_axis_angle((0.0, 1.0, 0.0), yaw), (0.0, 0.0, 0.0)))
    cache = {}

    def tex(index, srgb=False):
        if not textures or index is None or index >= len(model.images) or model.images[index] is None:
            return None
        if (index, srgb) not in cache:
            cache[(index, srgb)] = Texture(model.images[index].rgba, mips=True, srgb=srgb)
        return cache[(index, srgb)]

    draws = []
    for mesh in model.meshes:
        mat = model.materials[mesh.material_index] if textures and mesh.material_index is not None and mesh.material_index < len(model.materials) else None
        maps = dict(albedo_map=WHITE_1X1, normal_map=WHITE_1X1)
        material = (pbr_material_ubo((0.7, 0.7, 0.7, 1.0), 0.0, 0.5, 1.0) if program == PROGRAM_MODEL_PBR
                    else material_ubo((0.7, 0.7, 0.7, 1.0), 0.0, 0.5, 1.0))
        if program == PROGRAM_MODEL_PBR:
            maps.update(metallic_roughness_map=WHITE_1X1, occlusion_map=WHITE_1X1, emissive_map=WHITE_1X1)
        if mat is not None:
            maps["albedo_map"] = tex(mat.base_color_image, srgb=True) or WHITE_1X1
            maps["normal_map"] = tex(mat.normal_image) or WHITE_1X1
            if program == PROGRAM_MODEL_PBR:
                slots = {"metallic_roughness_map": tex(mat.metallic_roughness_image), "occlusion_map": tex(mat.occlusion_image),
                         "emissive_map": tex(mat.emissive_image, srgb=True)}
                maps.update({k: v or WHITE_1X1 for k, v in slots.items()})
                material = pbr_material_ubo(mat.base_color, mat.metallic, mat.roughness, mat.ao, normal_scale=mat.normal_scale,
                                            emissive=mat.emissive[:3], has_base_color=maps["albedo_map"] is not WHITE_1X1,
                                            has_normal=maps["normal_map"] is not WHITE_1X1,
                                            has_metallic_roughness=slots["metallic_roughness_map"] is not None,
                                            has_occlusion=slots["occlusion_map"] is not None, has_emissive=slots["emissive_map"] is not None)
            else:
                material = material_ubo(mat.base_color, mat.metallic, mat.roughness, mat.ao)
        draws.append(DrawSpec(vertices=mesh.interleave(), stride=48, count=int(mesh.indices.size), indices=mesh.indices,
                              program=program, cull_mode=CULL_NONE, front_face=FRONT_CCW, camera=cam, object=obj,
                              light=light_ubo(direction=(0.0, -1.0, 0.0), intensity=0.0, num_point=1),
                              material=material,
                              point_lights=point_light((2.0, 2.0, 2.0), 10.0, (1.0, 1.0, 1.0), 5.0), **maps))
    tag = "-textured" if textures else ""
    return Scene(f"gltf-{os.path.basename(os.path.dirname(os.path.abspath(path)))}-{model.total_triangles}{tag}", width, height, draws,
                 clear_color=(0.1, 0.1, 0.15, 1.0))


SMALL_CASES = {
    "hello": lambda: hello_triangle(64, 64),
    "fan": shared_edge_fan,
    "near_clip": near_clip_case,
    "depth_tie": depth_tie_case,
    "cull_scissor": cull_scissor_case,
    "multi_draw": multi_draw_case,
    "huge": huge_triangle_case,
    "textured": textured_quad_case,
    "pbr": pbr_spheres_case,
    "mips": mip_ground_case,
    "aniso": lambda: mip_ground_case(max_anisotropy=16),
    "alpha_mask": alpha_mask_case,
    "random_small": lambda: random_triangles(300, 320, 200, seed=42, rmin=2, rmax=40),
    "sphere_small": lambda: displaced_sphere(24, 17, 256, 160, seed=3),
}
