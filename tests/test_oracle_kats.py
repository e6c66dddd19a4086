"""Pins the CPU oracle against every known answer the reference holds for the hot path
(tests/golden/kats.json, produced by tools/make_kats.py from the reference checkout)."""
import json
import math
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
KATS = json.load(open(os.path.join(HERE, "golden", "kats.json")))


def test_k2_hello_triangle_coverage_256(oracle, scenes):
    """Analytic coverage under the Vulkan top-left rule (crates/renderer/src/renderer.rs:242-246 vertices)."""
    k = KATS["K2_hello_coverage_256"]
    r = oracle.render(scenes.hello_triangle(256, 256))
    cov = r["prim"] != 0xFFFFFFFF
    assert int(cov.sum()) == k["covered_pixels"] == 8192
    rows = np.where(cov.any(axis=1))[0]
    assert rows.min() == k["first_row"] and rows.max() == k["last_row"]
    assert list(np.where(cov[65])[0]) == k["row65"]
    xs = np.where(cov[191])[0]
    assert [int(xs.min()), int(xs.max())] == k["row191_range"]
    for y, n in k["row_counts"].items():
        assert int(cov[int(y)].sum()) == n
    # clear colour elsewhere (renderer.rs:484-488), alpha 1
    assert np.allclose(r["rgba"][0, 0], [0.1, 0.1, 0.15, 1.0])
    # centroid colour = (1/3,1/3,1/3): smooth perspective-correct interpolation with w = 1
    cy, cx = 149, 128   # centroid of (128,64),(64,192),(192,192) is (128, 149.33)
    assert np.allclose(r["rgba"][cy, cx, :3].sum(), 1.0, atol=1e-6)


def test_k1_screenshot_statistics(oracle, scenes):
    """The only golden render: screenshots/Hello Triangle.png (sRGB swapchain, crates/rhi/src/swapchain.rs:561-570)."""
    k = KATS["K1_screenshot"]
    w, h = k["client_size"]
    r = oracle.render(scenes.hello_triangle(w, h))
    rgb = r["bgra8"][:, :, [2, 1, 0]].astype(np.int32)
    assert rgb[h - 5, 5].tolist() == k["background_srgb8"]          # sRGB8 of linear (0.1, 0.1, 0.15)
    cov = r["prim"] != 0xFFFFFFFF
    ys, xs = np.where(cov)
    assert abs(int(xs.min()) - k["triangle_bbox_x"][0]) <= 2 and abs(int(xs.max()) - k["triangle_bbox_x"][1]) <= 2
    assert abs(int(ys.min()) - k["triangle_bbox_y_client"][0]) <= 2 and abs(int(ys.max()) - k["triangle_bbox_y_client"][1]) <= 2
    assert abs(int(cov.sum()) - k["covered_pixels"]) / k["covered_pixels"] < 0.005
    cx, cy = k["centroid_client"]
    assert np.abs(rgb[cy, cx] - np.array(k["centroid_srgb8"])).max() <= 2
    # red vertex at the top (no Y inversion in the shader), green bottom-left, blue bottom-right
    top = rgb[int(ys.min()) + 3, int(round(xs[ys == ys.min() + 3].mean()))]
    assert np.abs(top - np.array(k["top_row_rgb"])).max() <= 6 and top.argmax() == 0
    bl = rgb[int(ys.max()) - 3, int(xs[ys == ys.max() - 3].min()) + 8]
    br = rgb[int(ys.max()) - 3, int(xs[ys == ys.max() - 3].max()) - 8]
    assert np.abs(bl - np.array(k["bottom_left_rgb"])).max() <= 6 and bl.argmax() == 1
    assert np.abs(br - np.array(k["bottom_right_rgb"])).max() <= 6 and br.argmax() == 2


def test_k3_shader_constants(oracle):
    """lights.hlsli:63-73,152-159 and the fallback constants of pixel/model.hlsl:36-43,60."""
    k = KATS["K3_shader_constants"]
    L = oracle.lib()
    for d, rad, want in k["attenuation"]:
        assert L.oracle_attenuation(d, rad) == pytest.approx(want, abs=1e-6)
    for r, want in k["roughness_to_shininess"]:
        assert L.oracle_roughness_to_shininess(r) == want
    assert L.oracle_roughness_to_shininess(0.5) == k["fallback_shininess"] == 1025.0
    import ctypes as C
    v = (C.c_float * 3)(*k["fallback_light_dir"])
    one = (C.c_float * 3)(1.0, 1.0, 1.0)
    alb = (C.c_float * 3)(0.7, 0.7, 0.7)
    out = (C.c_float * 3)()
    L.oracle_blinn_phong(v, v, v, one, alb, C.c_float(1025.0), out)   # N = L = V
    assert out[0] + k["ambient_per_channel"] == pytest.approx(k["aligned_NLV_color"], abs=2e-5)


def test_k3_fallback_model_pixel(oracle, scenes):
    """A quad facing the fallback light direction, viewed along it: colour = 0.021 + 0.7 + 1 = 1.721."""
    k = KATS["K3_shader_constants"]
    n = np.array([1.0, 1.0, 1.0]) / math.sqrt(3.0)
    u = np.cross(n, [0.0, 1.0, 0.0]); u /= np.linalg.norm(u)
    v = np.cross(n, u)
    pos = np.array([-u - v, u - v, u + v, -u + v]) * 0.5
    verts = np.zeros((4, 12), dtype=np.float32)
    verts[:, 0:3] = pos; verts[:, 3:6] = n; verts[:, 8:12] = [1, 0, 0, 1]
    idx = np.array([0, 1, 2, 0, 2, 3], dtype=np.uint32)
    eye = n * 300.0   # far away: V ~ n everywhere
    view = scenes.look_at_rh(eye, (0, 0, 0), (0, 1, 0))
    proj = scenes.projection_vulkan(math.radians(1.0), 1.0, 0.1, 1000.0)
    d = scenes.DrawSpec(vertices=verts, stride=48, count=6, indices=idx, program=scenes.PROGRAM_MODEL,
                        cull_mode=scenes.CULL_NONE, camera=scenes.camera_ubo(view, proj, eye),
                        object=scenes.object_ubo(np.eye(4, dtype=np.float32)))
    r = oracle.render(scenes.Scene("k3", 64, 64, [d]))
    c = r["rgba"][32, 32]
    assert c[:3] == pytest.approx([k["aligned_NLV_color"]] * 3, abs=2e-3) and c[3] == 1.0
    assert oracle.lib().oracle_srgb8(float(c[0])) == 255      # saturates in the sRGB8 target


def test_k4_default_camera_matrices(oracle):
    """crates/scene/src/camera.rs:43-56,110-142 + the reference's tests camera.rs:552-569."""
    k = KATS["K4_matrices"]
    p = k["perspective"]
    P = oracle.mat_fn("oracle_camera_projection_perspective", math.radians(p["fovy_deg"]), p["aspect"], p["near"], p["far"])
    # column-major storage: P[col][row]
    assert P[0, 0] == pytest.approx(p["w"], rel=1e-6) and P[1, 1] == pytest.approx(p["m11_after_flip"], rel=1e-6)
    assert P[1, 1] < 0                                           # camera.rs:563-569
    assert P[2, 2] == pytest.approx(p["r"], rel=1e-6) and P[2, 3] == -1.0 and P[3, 2] == pytest.approx(p["m32"], rel=1e-6)
    V = oracle.mat_fn("oracle_camera_view_matrix", k["default_eye"], [0.0, 0.0, 0.0, 1.0])
    origin = V.T @ np.array([0, 0, 0, 1], dtype=np.float32)
    assert origin[:3] == pytest.approx(k["view_times_origin"], abs=1e-5)   # camera.rs:552-560


def test_k4_ubo_and_transform_behaviour(oracle, scenes):
    """ubo.rs:436-448 (view_projection == projection*view), :479-523 (normal matrix), transform.rs:231-267."""
    view = scenes.look_at_rh((0, 2, 5), (0, 0, 0), (0, 1, 0))
    proj = scenes.perspective_rh(math.radians(45.0), 16 / 9, 0.1, 100.0)
    vp = oracle.mat_fn("oracle_glam_mat4_mul", proj, view)
    assert np.allclose(vp, scenes.mat_mul(proj, view), atol=1e-6)
    # normal matrix = inverse transpose; zero scale -> identity
    m = scenes.trs((2.0, 3.0, 4.0), scenes.quat_axis_angle((0, 1, 0), 0.7), (1.0, 2.0, 3.0))
    nm = oracle.mat_fn("oracle_normal_matrix", m)
    want = np.linalg.inv(m.T.astype(np.float64)).T.T
    assert np.allclose(nm, want, atol=1e-5)
    z = scenes.trs((0.0, 1.0, 1.0), (0, 0, 0, 1), (0, 0, 0))
    assert np.array_equal(oracle.mat_fn("oracle_normal_matrix", z), np.eye(4, dtype=np.float32))
    # TRS composition: translation after rotation after scale
    t = oracle.mat_fn("oracle_glam_from_scale_rotation_translation", [2.0, 2.0, 2.0], [0.0, 0.0, 0.0, 1.0], [1.0, 2.0, 3.0])
    assert np.allclose(t.T @ np.array([1, 1, 1, 1], dtype=np.float32), [3, 4, 5, 1])
    assert np.allclose(t, scenes.trs((2, 2, 2), (0, 0, 0, 1), (1, 2, 3)))


def test_k6_layouts(scenes):
    """vertex.rs:177-319 and ubo.rs:421-596 size/offset assertions, as seen by the byte builders."""
    k = KATS["K6_layouts"]
    s = scenes.hello_triangle()
    assert s.draws[0].stride == k["TriangleVertex"]["size"] and s.draws[0].vertex_bytes().size == 3 * 24
    v = scenes._pack_vertex48(np.zeros((1, 3)), np.ones((1, 3)), np.full((1, 2), 2.0), np.full((1, 4), 3.0)).view(np.uint8).reshape(-1)
    f = v.view(np.float32)
    assert v.size == k["Vertex"]["size"]
    assert f[k["Vertex"]["normal"] // 4] == 1.0 and f[k["Vertex"]["tex_coord"] // 4] == 2.0 and f[k["Vertex"]["tangent"] // 4] == 3.0
    view, proj, cam = scenes.default_camera(1920, 1080)
    assert len(cam) == k["CameraUbo"]["size"]
    assert np.frombuffer(cam, dtype=np.float32)[k["CameraUbo"]["camera_position"] // 4 + 2] == 5.0
    assert len(scenes.object_ubo(np.eye(4, dtype=np.float32))) == k["ObjectUbo"]["size"]
    assert len(scenes.light_ubo()) == k["LightUBO_hlsl"]["size"] and len(scenes.material_ubo()) == k["MaterialData_hlsl"]["size"]
    assert len(scenes.point_light((0, 0, 0), 1, (1, 1, 1), 1)) == k["PointLight"]["size"]
    assert len(scenes.spot_light((0, 0, 0), 1, (0, 0, 1), 0, (1, 1, 1), 1)) == k["SpotLight_hlsl"]["size"]


def test_oracle_edge_cases(oracle, scenes):
    """Domain edge cases: shared edges are covered exactly once, depth ties keep the earlier primitive,
    back faces are culled, scissor clips, off-screen / degenerate input yields the clear colour."""
    fan = scenes.shared_edge_fan()
    r = oracle.render(fan)
    # every covered pixel is hit by exactly one fan triangle: brute-force count with the oracle itself,
    # one triangle at a time, must sum to the union
    total = np.zeros((fan.height, fan.width), dtype=np.int32)
    d = fan.draws[0]
    for t in range(d.num_triangles):
        one = scenes.DrawSpec(vertices=d.vertices[3 * t:3 * t + 3], stride=24, count=3, cull_mode=scenes.CULL_NONE,
                              depth_test=False, depth_write=False)
        total += (oracle.render(scenes.Scene("one", fan.width, fan.height, [one]))["prim"] != 0xFFFFFFFF)
    assert total.max() == 1 and np.array_equal(total == 1, r["prim"] != 0xFFFFFFFF)
    tie = oracle.render(scenes.depth_tie_case())
    both = tie["prim"][32, 48]
    assert both == 0          # triangles 0 and 1 overlap at equal depth: LESS keeps primitive 0
    cs = oracle.render(scenes.cull_scissor_case())
    assert set(np.unique(cs["prim"])) <= {0, 0xFFFFFFFF} or set(np.unique(cs["prim"])) <= {1, 0xFFFFFFFF}
    ys, xs = np.where(cs["prim"] != 0xFFFFFFFF)
    assert xs.min() >= 16 and xs.max() < 16 + 90 and ys.min() >= 8 and ys.max() < 8 + 60
    empty = scenes.Scene("empty", 33, 17, [], clear_color=(0.25, 0.5, 0.75, 1.0))
    e = oracle.render(empty)
    assert (e["prim"] == 0xFFFFFFFF).all() and np.allclose(e["rgba"], [0.25, 0.5, 0.75, 1.0])


def test_oracle_threads_and_bands_agree(oracle, scenes):
    s = scenes.random_triangles(500, 320, 200, seed=9)
    a = oracle.render(s, nthreads=1)
    b = oracle.render(s, nthreads=5)
    assert np.array_equal(a["prim"], b["prim"]) and np.array_equal(a["rgba"], b["rgba"]) and np.array_equal(a["bgra8"], b["bgra8"])
    band = oracle.render(s, rows=(64, 128))
    assert np.array_equal(band["prim"][64:128], a["prim"][64:128]) and (band["prim"][:64] == 0xFFFFFFFF).all()


def test_near_clip_is_watertight(oracle, scenes):
    """The ground quad crosses the near plane and w = 0: clipping must not leave cracks along the shared diagonal."""
    r = oracle.render(scenes.near_clip_case())
    cov = r["prim"] != 0xFFFFFFFF
    assert cov.sum() > 1000 and not np.isnan(r["rgba"]).any()
    # below the horizon every pixel of the bottom rows is ground
    assert cov[-1].all() and cov[-20].all()


# ------------------------------------------------------------------------------------------------
# K7: Cook-Torrance program (pbr.hlsli, pixel/model_pbr.hlsl) against closed forms evaluated in float64
# ------------------------------------------------------------------------------------------------
def _pbr_direct64(N, V, L, radiance, albedo, metallic, roughness):
    """pbr.hlsli:292-333 written from the published Cook-Torrance / GGX / Smith / Schlick formulas."""
    H = (V + L) / np.linalg.norm(V + L)
    a2 = roughness ** 4
    ndh, ndv, ndl = max(N @ H, 0.0), max(N @ V, 0.0), max(N @ L, 0.0)
    D = a2 / max(math.pi * (ndh * ndh * (a2 - 1.0) + 1.0) ** 2, 1e-4)
    k = (roughness + 1.0) ** 2 / 8.0
    G = (ndv / max(ndv * (1 - k) + k, 1e-4)) * (ndl / max(ndl * (1 - k) + k, 1e-4))
    F0 = 0.04 + (albedo - 0.04) * metallic
    F = F0 + (1.0 - F0) * (1.0 - min(max(H @ V, 0.0), 1.0)) ** 5
    spec = D * G * F / (4.0 * ndv * ndl + 1e-4)
    kD = (1.0 - F) * (1.0 - metallic)
    return (kD * albedo / math.pi + spec) * radiance * ndl


def test_k7_pbr_brdf_terms(oracle):
    L = oracle.lib()
    # D(N.H = 1, roughness 0.5) = a^2 / (pi a^4) = 1 / (pi * 0.0625)
    assert L.oracle_distribution_ggx(1.0, 0.5) == pytest.approx(1.0 / (math.pi * 0.0625), rel=1e-6)
    # fully rough: a = 1 -> D = 1/pi for every N.H
    for x in (0.0, 0.3, 1.0):
        assert L.oracle_distribution_ggx(x, 1.0) == pytest.approx(1.0 / math.pi, rel=1e-6)
    # EPSILON clamp of the denominator (pbr.hlsli:18,68): a^2 = 1e-8 at N.H = 1 gives pi * 1e-16 -> clamped to 1e-4
    assert L.oracle_distribution_ggx(1.0, 0.01) == pytest.approx(1e-8 / 1e-4, rel=1e-5)
    # Schlick-GGX: G(1, r) = 1; G(0.5, 1) = 0.5 / (0.5 * 0.5 + 0.5); G(0, r) = 0
    assert L.oracle_geometry_schlick_ggx(1.0, 0.3) == pytest.approx(1.0, rel=1e-6)
    assert L.oracle_geometry_schlick_ggx(0.5, 1.0) == pytest.approx(0.5 / 0.75, rel=1e-6)
    assert L.oracle_geometry_schlick_ggx(0.0, 0.7) == 0.0


@pytest.mark.parametrize("metallic,roughness", [(0.0, 0.5), (1.0, 0.3), (0.4, 0.01)])
def test_k7_pbr_pixel_closed_form(oracle, scenes, metallic, roughness):
    """A z = 0 quad with normal +z, one directional and one point light, no textures: every centre-row pixel must
    equal the float64 evaluation of model_pbr.hlsl:159-320 at the ray/plane intersection of that pixel."""
    W = H = 48
    pos = np.array([[-2, -2, 0], [2, -2, 0], [2, 2, 0], [-2, 2, 0]], dtype=np.float32)
    verts = np.zeros((4, 12), dtype=np.float32)
    verts[:, 0:3] = pos; verts[:, 3:6] = [0, 0, 1]; verts[:, 8:12] = [1, 0, 0, 1]
    idx = np.array([0, 1, 2, 0, 2, 3], dtype=np.uint32)
    eye = np.array([0.0, 0.0, 3.0])
    fov = math.radians(45.0)
    view = scenes.look_at_rh(tuple(eye), (0, 0, 0), (0, 1, 0))
    proj = scenes.projection_vulkan(fov, 1.0, 0.1, 100.0)
    albedo = np.array([0.8, 0.5, 0.3])
    emissive = np.array([0.01, 0.02, 0.03])
    ao = 0.8
    ldir = np.array([0.3, -0.2, -1.0])
    lcol, lint = np.array([1.0, 0.9, 0.8]), 2.0
    ppos, prad, pcol, pint = np.array([1.0, 1.0, 1.5]), 6.0, np.array([0.4, 0.6, 1.0]), 5.0
    d = scenes.DrawSpec(vertices=verts, stride=48, count=6, indices=idx, program=scenes.PROGRAM_MODEL_PBR,
                        cull_mode=scenes.CULL_NONE, camera=scenes.camera_ubo(view, proj, tuple(eye)),
                        object=scenes.object_ubo(np.eye(4, dtype=np.float32)),
                        light=scenes.light_ubo(direction=tuple(ldir), intensity=lint, color=tuple(lcol), num_point=1),
                        point_lights=scenes.point_light(tuple(ppos), prad, tuple(pcol), pint),
                        material=scenes.pbr_material_ubo(tuple(albedo) + (0.9,), metallic, roughness, ao, emissive=tuple(emissive)))
    r = oracle.render(scenes.Scene("k7", W, H, [d]))
    rough = max(roughness, 0.04)
    N = np.array([0.0, 0.0, 1.0])
    t = math.tan(fov / 2)
    for px in range(4, W, 5):
        py = H // 2
        # pixel centre -> NDC (Vulkan y down, projection_vulkan flips y) -> ray from the eye -> plane z = 0
        nx, ny = (px + 0.5) / W * 2 - 1, (py + 0.5) / H * 2 - 1
        ray = np.array([nx * t, -ny * t, -1.0])
        wp = eye + ray * (eye[2] / 1.0)
        V = (eye - wp) / np.linalg.norm(eye - wp)
        Ld = -ldir / np.linalg.norm(ldir)
        lighting = _pbr_direct64(N, V, Ld, lcol * lint, albedo, metallic, rough)
        lv = ppos - wp
        dist = np.linalg.norm(lv)
        att = (1.0 / (dist * dist + 1.0)) * min(max(1.0 - dist / prad, 0.0), 1.0) ** 2
        lighting = lighting + _pbr_direct64(N, V, lv / dist, pcol * pint * att, albedo, metallic, rough)
        up = N[1] * 0.5 + 0.5
        amb = np.array([0.08, 0.06, 0.04]) + (np.array([0.15, 0.18, 0.25]) - np.array([0.08, 0.06, 0.04])) * up
        want = amb * albedo * ao * (1.0 - metallic) + lighting * (1.0 + (ao - 1.0) * 0.5) + emissive
        got = r["rgba"][py, px]
        assert r["prim"][py, px] != 0xFFFFFFFF
        assert got[:3] == pytest.approx(want, rel=2e-4, abs=2e-5), (px, got, want)
        assert got[3] == pytest.approx(0.9)


def test_k7_pbr_alpha_cutoff(oracle, scenes):
    """model_pbr.hlsl:174-178: constant alpha below the cutoff removes the draw; a textured base colour whose texels
    straddle the cutoff loses exactly the fragments whose filtered alpha is below it (colour, depth and id untouched)."""
    sc = scenes.SMALL_CASES["pbr"]()
    r = oracle.render(sc)
    nt = sc.draws[0].num_triangles
    drawn = set(np.unique(r["prim"][r["prim"] != 0xFFFFFFFF] // nt).tolist())
    assert drawn == {0, 1, 2, 4}
    # a textured base colour whose texel alpha straddles the cutoff is discarded fragment by fragment: the left half of this quad's
    # texture has alpha 0 (discarded: the clear colour and NO_PRIM stay), the right half alpha 255 (kept)
    tex = np.full((8, 8, 4), 255, dtype=np.uint8)
    tex[:, :4, 3] = 0
    quad = _textured_quad_scene(scenes, scenes.Texture(tex), 64, 64, 1.0)
    d = quad.draws[0]
    d.program = scenes.PROGRAM_MODEL_PBR
    d.material = scenes.pbr_material_ubo((1.0, 1.0, 1.0, 1.0), alpha_cutoff=0.5, has_base_color=True)
    r = oracle.render(quad)
    kept = r["prim"] != 0xFFFFFFFF
    # bilinear alpha crosses 0.5 exactly between texel columns 3 and 4 = at u = 0.5: pixel centres left of it are discarded
    assert not kept[:, :32].any() and kept[:, 32:].all()
    assert np.allclose(r["rgba"][:, :32], np.array(quad.clear_color, dtype=np.float32))
    assert (r["depth"][:, :32] == quad.clear_depth).all()


# ------------------------------------------------------------------------------------------------
# K8: texture fidelity -- mip chain rule, LOD selection and sRGB decode against closed forms
# ------------------------------------------------------------------------------------------------
def _textured_quad_scene(scenes, tex, W, H, uv_scale):
    """A quad filling the viewport exactly, uv = uv_scale * (0..1): texels per pixel = uv_scale * tex_size / W."""
    pos = np.array([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0]], dtype=np.float32)
    verts = np.zeros((4, 12), dtype=np.float32)
    verts[:, 0:3] = pos; verts[:, 3:6] = [0, 0, 1]; verts[:, 8:12] = [1, 0, 0, 1]
    verts[:, 6:8] = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], dtype=np.float32) * uv_scale
    idx = np.array([0, 1, 2, 0, 2, 3], dtype=np.uint32)
    ident = np.eye(4, dtype=np.float32)
    d = scenes.DrawSpec(vertices=verts, stride=48, count=6, indices=idx, program=scenes.PROGRAM_MODEL_FULL,
                        cull_mode=scenes.CULL_NONE, depth_test=False, depth_write=False,
                        camera=scenes.camera_ubo(ident, ident, (0.0, 0.0, 1.0)), object=scenes.object_ubo(ident),
                        light=scenes.light_ubo(intensity=0.0), material=scenes.material_ubo((1.0, 1.0, 1.0, 1.0), 0.0, 1.0, 1.0),
                        albedo_map=tex, normal_map=scenes.WHITE_1X1)
    return scenes.Scene("k8", W, H, [d])


def test_k8_mip_chain_rule(scenes):
    """2x2 box filter on the bytes, round half up; odd sizes clamp the last row / column; the chain ends at 1x1."""
    img = np.zeros((3, 5, 4), dtype=np.uint8)
    img[..., 0] = np.arange(15).reshape(3, 5) * 10
    img[..., 3] = 255
    chain = scenes.mip_chain(img)
    assert [l.shape[:2] for l in chain] == [(3, 5), (1, 2), (1, 1)]
    assert chain[1][0, 0, 0] == (0 + 10 + 50 + 60 + 2) // 4 and chain[1][0, 1, 0] == (20 + 30 + 70 + 80 + 2) // 4
    assert chain[2][0, 0, 0] == (int(chain[1][0, 0, 0]) * 2 + int(chain[1][0, 1, 0]) * 2 + 2) // 4
    assert all(l[..., 3].min() == 255 for l in chain)


@pytest.mark.parametrize("texels_per_pixel,expect_level", [(1.0, 0.0), (2.0, 1.0), (4.0, 2.0), (0.25, 0.0)])
def test_k8_lod_selects_the_level_whose_texel_matches_the_pixel(oracle, scenes, texels_per_pixel, expect_level):
    """With n texels per pixel lambda = log2(n): a 1-texel checker is resolved at level 0 and becomes its mean from level 1
    on (every 2x2 block of the checker averages to the same grey); the ambient term 0.03 * albedo * ao is the only light."""
    n = 64
    yy, xx = np.mgrid[0:n, 0:n]
    tex = np.zeros((n, n, 4), dtype=np.uint8)
    tex[..., 0:3] = (255 * ((xx + yy) & 1))[..., None]
    tex[..., 3] = 255
    W = H = 64
    sc = _textured_quad_scene(scenes, scenes.Texture(tex, mips=True), W, H, uv_scale=texels_per_pixel * W / n)
    out = oracle.render(sc)["rgba"][8:56, 8:56, 0] / 0.03      # ambient = albedo * 0.03 * ao(1)
    if expect_level >= 1.0:
        assert np.allclose(out, 128.0 / 255.0, atol=1e-5)       # (0 + 255 + 255 + 0 + 2) >> 2 = 128 on every coarser level
    else:
        assert out.min() < 0.3 and out.max() > 0.7              # the checker is still resolved (bilinear between texels)


def test_k8_srgb_decode(oracle, scenes):
    tex = np.zeros((1, 1, 4), dtype=np.uint8)
    for byte in (0, 10, 11, 128, 200, 255):
        tex[0, 0] = (byte, byte, byte, 255)
        sc = _textured_quad_scene(scenes, scenes.Texture(tex.copy(), srgb=True), 8, 8, 1.0)
        got = float(oracle.render(sc)["rgba"][4, 4, 0]) / 0.03
        c = byte / 255.0
        want = c / 12.92 if c <= 0.04045 else ((c + 0.055) / 1.055) ** 2.4
        assert got == pytest.approx(want, rel=2e-6, abs=1e-8), byte


# ------------------------------------------------------------------------------------------------
# K9: anisotropic filtering (the Vulkan specification's example filter) against closed forms
# ------------------------------------------------------------------------------------------------
def _stripe_texture(scenes, n, along_v, max_anisotropy):
    """1-texel stripes that vary along v (along_v) or along u: every 2x2 block averages to the same grey from level 1 on."""
    yy, xx = np.mgrid[0:n, 0:n]
    tex = np.zeros((n, n, 4), dtype=np.uint8)
    tex[..., 0:3] = (255 * ((yy if along_v else xx) & 1))[..., None]
    tex[..., 3] = 255
    return scenes.Texture(tex, mips=True, max_anisotropy=max_anisotropy)


def _aniso_quad(scenes, tex, W, H, su, sv):
    """The K8 quad with separate uv scales: su * n / W texels per pixel along x, sv * n / H along y."""
    sc = _textured_quad_scene(scenes, tex, W, H, 1.0)
    v = sc.draws[0].vertices
    v[:, 6] *= su
    v[:, 7] *= sv
    return sc


@pytest.mark.parametrize("max_aniso,resolved", [(1, False), (2, False), (4, False), (8, True), (16, True)])
def test_k9_taps_follow_the_long_axis_and_lod_drops_by_log2_n(oracle, scenes, max_aniso, resolved):
    """8 texels per pixel along x, 1 along y, stripes that vary along y only.  Trilinear: lambda = log2(8) = 3, the stripes are
    gone (grey 128/255).  Anisotropic: N = min(ceil(8 / 1), max) taps along x at lambda = log2(8 / N): with N = 8 that is level 0,
    and taps displaced along x leave the stripes untouched; with N = 2 or 4 lambda is 2 or 1, still grey."""
    n = W = H = 64
    sc = _aniso_quad(scenes, _stripe_texture(scenes, n, True, max_aniso), W, H, 8.0, 1.0)
    out = oracle.render(sc)["rgba"][8:56, 8:56, 0] / 0.03
    if resolved:
        # the pixel centres sit on texel centres along v (1 texel per pixel): the bilinear tap returns the texel itself
        assert set(np.round(out.reshape(-1), 4)) == {0.0, 1.0}
        assert np.array_equal(out[:, 0] > 0.5, (np.arange(8, 56) & 1).astype(bool))
    else:
        assert np.allclose(out, 128.0 / 255.0, atol=1e-5)


def test_k9_major_axis_y_and_isotropic_footprint(oracle, scenes):
    """The same with the axes exchanged (taps along y), and an isotropic footprint: N = ceil(1) = 1 tap at the centre at the
    trilinear lambda, whatever max_anisotropy allows -- the image equals the trilinear one bit for bit."""
    n = W = H = 64
    sc = _aniso_quad(scenes, _stripe_texture(scenes, n, False, 16), W, H, 1.0, 8.0)
    out = oracle.render(sc)["rgba"][8:56, 8:56, 0] / 0.03
    assert set(np.round(out.reshape(-1), 4)) == {0.0, 1.0}
    assert np.array_equal(out[0, :] > 0.5, (np.arange(8, 56) & 1).astype(bool))
    rng = np.random.default_rng(9)
    noise = rng.integers(0, 256, (n, n, 4), dtype=np.uint8)
    a = oracle.render(_aniso_quad(scenes, scenes.Texture(noise, mips=True, max_anisotropy=16), W, H, 3.0, 3.0))["rgba"]
    b = oracle.render(_aniso_quad(scenes, scenes.Texture(noise, mips=True, max_anisotropy=1), W, H, 3.0, 3.0))["rgba"]
    assert np.array_equal(a, b)


def test_k9_mean_of_taps_closed_form(oracle, scenes):
    """Taps displaced along the long axis (y) over a texture that varies along x only all return the value at the centre, so
    the image is the trilinear lookup of that texture at the anisotropic lambda = log2(Pmax / N) -- which differs from the
    isotropic one (log2(Pmax)) -- checked against float64 arithmetic.  Level 0 holds floor(x / 2), level 1 its exact 2x2 means."""
    n, W, H = 256, 64, 64
    yy, xx = np.mgrid[0:n, 0:n]
    tex = np.zeros((n, n, 4), dtype=np.uint8)
    tex[..., 0:3] = (xx // 2)[..., None]           # value = floor(x / 2): level l holds floor(x_l * 2^l / 2) for l >= 1 exactly
    tex[..., 3] = 255
    # 1.5 texels per pixel along x (the short axis), 6 along y: N = ceil(6 / 1.5) = 4, lambda = log2(6 / 4)
    sc = _aniso_quad(scenes, scenes.Texture(tex, mips=True, max_anisotropy=16), W, H, 1.5 * W / n, 6.0 * H / n)
    out = oracle.render(sc)["rgba"][:, :, 0].astype(np.float64) / 0.03
    lam = np.log2(6.0 / 4.0)
    f = lam                                        # between level 0 and level 1
    for px in (10, 20, 33, 50):
        u = (px + 0.5) / W * (1.5 * W / n)         # uv at the pixel centre
        def level_value(l):                        # bilinear lookup of the ramp on level l (texel k holds floor(k * 2^l / 2) for l >= 1, k // 2 on level 0)
            size = n >> l
            fx = u * size - 0.5
            x0 = int(np.floor(fx)); a = fx - x0
            val = (lambda k: (k // 2) if l == 0 else k * (1 << l) // 2)
            return ((1 - a) * val(x0) + a * val(x0 + 1)) / 255.0
        want = (1 - f) * level_value(0) + f * level_value(1)
        assert out[30, px] == pytest.approx(want, abs=2e-6), px


def test_k9_binding_default_is_trilinear(oracle, scenes):
    sc = scenes.SMALL_CASES["mips"]()
    assert all(d.albedo_map.max_anisotropy == 1 for d in sc.draws)
    assert scenes.SMALL_CASES["aniso"]().draws[0].albedo_map.max_anisotropy == 16
    a, b = oracle.render(sc)["rgba"], oracle.render(scenes.SMALL_CASES["aniso"]())["rgba"]
    assert not np.array_equal(a, b)                # the receding ground is where the two filters differ


def overlapping_clip_pieces_scene(scenes, op, write=True, blend=None, clear_depth=0.0):
    """One triangle that crosses the near AND the far plane (found by tools/soak_fuzz.py, seed 522444, primitive 94): its clip polygon is a fan of
    three pieces, and after the snap to the sub-pixel grid a sliver piece lies flipped over its neighbour -- pixel (183, 38) is covered twice by
    the same primitive (cull mode None keeps both orientations)."""
    v = np.array([[0.39559028, 0.7726351, -0.00514638, 0.5161093, 0.54139465, 0.9170683],
                  [0.33086586, 0.6824755, 1.0311986, 0.8583772, 0.9776306, 0.608155],
                  [0.3072718, 0.6554115, 0.8854149, 0.7304598, 0.20604184, 0.6845877]], dtype=np.float32)
    d = scenes.DrawSpec(vertices=v, stride=24, count=3, program=scenes.PROGRAM_TRIANGLE, cull_mode=scenes.CULL_NONE,
                        depth_test=True, depth_write=write, depth_compare=op, blend=blend)
    return scenes.Scene(f"overlapping-clip-pieces-op{op}", 272, 45, [d], clear_color=(0.2, 0.3, 0.4, 0.5), clear_depth=clear_depth)


def test_fragments_of_one_primitive_are_resolved_without_an_order(oracle, scenes):
    """Overlapping pieces of one clipped triangle: Always-with-write keeps the nearer fragment (a7 of the oracle; DESIGN.md "Depth key"), so the
    result does not depend on the order the clipper emits pieces in -- it equals what Less leaves from a far clear."""
    less = oracle.render(overlapping_clip_pieces_scene(scenes, scenes.CMP_LESS, clear_depth=1.0), want_bgra8=False)
    greater = oracle.render(overlapping_clip_pieces_scene(scenes, scenes.CMP_GREATER, clear_depth=0.0), want_bgra8=False)
    always = oracle.render(overlapping_clip_pieces_scene(scenes, scenes.CMP_ALWAYS, clear_depth=0.0), want_bgra8=False)
    cov = less["prim"] == 0
    assert cov.sum() > 0 and np.array_equal(cov, always["prim"] == 0) and np.array_equal(cov, greater["prim"] == 0)
    twice = cov & (less["depth"] != greater["depth"])
    assert twice[38, 183] and twice.sum() >= 1                      # the scene does what its name says
    assert np.array_equal(always["depth"][cov].view(np.uint32), less["depth"][cov].view(np.uint32))
