"""GPU parity tests: the HIP path through the C ABI vs the CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star): winning primitive id per pixel bit-exact, stored depth bit-exact,
linear float colour |dRGB| < 1e-4, sRGB8 output within 1 LSB.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RGB_TOL = 1e-4


def _render_both(mirhi, oracle, device, scene, fmt=None, want_depth=False):
    fmt = mirhi.Format.R32G32B32A32_SFLOAT if fmt is None else fmt
    res = mirhi.SceneResources(device, scene, fmt, want_prim=True, want_depth=want_depth)
    res.render()
    out = res.read()
    res.destroy()
    ref = oracle.render(scene, want_bgra8=True)
    return out, ref


def _check(out, ref, name, depth=False):
    diff = out["prim"] != ref["prim"]
    assert not diff.any(), f"{name}: {int(diff.sum())} pixels differ in winning primitive id (first at {np.argwhere(diff)[0]})"
    if out["color"].dtype == np.float32:
        # compare where finite in the oracle; NaNs must agree
        a, b = out["color"][..., :4], ref["rgba"]
        nan_a, nan_b = np.isnan(a), np.isnan(b)
        assert np.array_equal(nan_a, nan_b), f"{name}: NaN pattern differs"
        err = np.abs(np.where(nan_b, 0, a) - np.where(nan_b, 0, b))
        # tolerance is absolute on [0,1]; HDR values (float target is unclamped) get the same relative bound
        scale = np.maximum(1.0, np.abs(np.where(nan_b, 0, b)))
        worst = float((err / scale).max())
        assert worst < RGB_TOL, f"{name}: max |dRGBA| = {worst}"
    else:
        d = np.abs(out["color"].astype(np.int32) - ref["bgra8"].astype(np.int32))
        assert d.max() <= 1, f"{name}: sRGB8 differs by {d.max()} LSB"
    if depth:
        covered = ref["prim"] != 0xFFFFFFFF
        assert np.array_equal(out["depth"].view(np.uint32)[covered], ref["depth"].view(np.uint32)[covered]), f"{name}: depth bits differ"


@pytest.mark.parametrize("case", ["hello", "fan", "near_clip", "depth_tie", "cull_scissor", "multi_draw", "huge",
                                  "textured", "pbr", "mips", "aniso", "alpha_mask", "random_small", "sphere_small"])
def test_small_cases_float(mirhi, oracle, device, scenes, case):
    scene = scenes.SMALL_CASES[case]()
    out, ref = _render_both(mirhi, oracle, device, scene, want_depth=any(d.depth_test for d in scene.draws))
    _check(out, ref, scene.name, depth=any(d.depth_test for d in scene.draws))


@pytest.mark.parametrize("case", ["hello", "random_small", "sphere_small", "textured", "pbr", "mips", "aniso", "alpha_mask"])
def test_small_cases_srgb8(mirhi, oracle, device, scenes, case):
    scene = scenes.SMALL_CASES[case]()
    out, ref = _render_both(mirhi, oracle, device, scene, fmt=mirhi.Format.B8G8R8A8_SRGB)
    _check(out, ref, scene.name)


def test_hello_triangle_kat(mirhi, device, scenes):
    """K2: 8192 covered pixels, rows 65..191, row 65 = x{127,128}, row 191 = x 64..191."""
    scene = scenes.hello_triangle(256, 256)
    res = mirhi.SceneResources(device, scene, want_prim=True)
    res.render()
    out = res.read()
    res.destroy()
    cov = out["prim"] != 0xFFFFFFFF
    assert int(cov.sum()) == 8192
    rows = np.where(cov.any(axis=1))[0]
    assert rows.min() == 65 and rows.max() == 191
    assert list(np.where(cov[65])[0]) == [127, 128]
    assert np.where(cov[191])[0].min() == 64 and np.where(cov[191])[0].max() == 191
    assert np.allclose(out["color"][0, 0], [0.1, 0.1, 0.15, 1.0])


@pytest.mark.parametrize("compare", ["Less", "LessOrEqual", "Greater", "GreaterOrEqual", "Always"])
def test_depth_compare_ops(mirhi, oracle, device, scenes, compare):
    op = getattr(mirhi.CompareOp, compare)
    scene = scenes.depth_tie_case()
    extra = scenes.random_triangles(60, scene.width, scene.height, seed=77, rmin=5, rmax=40).draws[0]
    scene.draws.append(extra)
    for d in scene.draws:
        d.depth_compare = op
    scene.clear_depth = 0.0 if "Greater" in compare else 1.0
    out, ref = _render_both(mirhi, oracle, device, scene, want_depth=True)
    _check(out, ref, f"depth-{compare}", depth=(compare != "Always"))


def test_config2_10k_triangles_1080p(mirhi, oracle, device, scenes):
    scene = scenes.random_triangles()
    out, ref = _render_both(mirhi, oracle, device, scene)
    _check(out, ref, scene.name)


def test_config3_sphere_70k(mirhi, oracle, device, scenes):
    scene = scenes.displaced_sphere()
    out, ref = _render_both(mirhi, oracle, device, scene)
    _check(out, ref, scene.name)


def test_determinism_and_resubmit(mirhi, device, scenes):
    scene = scenes.random_triangles(2000, 640, 360, seed=5)
    res = mirhi.SceneResources(device, scene, want_prim=True)
    res.render()
    a = res.read()
    for _ in range(3):
        res.render()
    b = res.read()
    res.destroy()
    assert np.array_equal(a["prim"], b["prim"]) and np.array_equal(a["color"], b["color"])


def test_config4_grid_1m_triangles_4k(mirhi, oracle, device, scenes):
    scene = scenes.heightfield_grid()
    out, ref = _render_both(mirhi, oracle, device, scene)
    _check(out, ref, scene.name)


def test_config5_boxhall_textured_4k(mirhi, oracle, device, scenes):
    scene = scenes.box_hall()
    out, ref = _render_both(mirhi, oracle, device, scene)
    _check(out, ref, scene.name)


def test_gltf_dancer_1080p(mirhi, oracle, device, scenes):
    """The only real mesh the reference ships (integration_test.rs:7-83), loaded with Model::load semantics."""
    import os
    scene = scenes.gltf_model(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dancer", "scene.gltf"))
    out, ref = _render_both(mirhi, oracle, device, scene, want_depth=True)
    _check(out, ref, scene.name, depth=True)


@pytest.mark.parametrize("program", ["full", "pbr"])
def test_gltf_dancer_with_its_normal_map(mirhi, oracle, device, scenes, program):
    """SURVEY 8f rank 1, texture half: the asset's own normal map (PNG -> RGBA8 by host/image_decode.hpp, mip chain on the
    device, trilinear) drives model_full / model_pbr shading of the real mesh."""
    import os
    prog = scenes.PROGRAM_MODEL_FULL if program == "full" else scenes.PROGRAM_MODEL_PBR
    scene = scenes.gltf_model(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dancer", "scene.gltf"),
                              program=prog, textures=True)
    assert scene.draws[0].normal_map.rgba8.shape == (1024, 1024, 4)
    out, ref = _render_both(mirhi, oracle, device, scene, want_depth=True)
    _check(out, ref, scene.name, depth=True)


@pytest.mark.parametrize("variant", ["greater", "two_states", "less_then_greater", "teams_off"])
def test_gltf_dancer_two_team_variant_states(mirhi, oracle, device, scenes, variant, monkeypatch):
    """The mesh raster variant with two teams of waves per tile (chosen for mesh scopes that cover a small part of the frame,
    like this asset) under the generic depth key (GREATER against a 0.0 clear), across two segments of one scope (the
    second one loads the depth the first one carried over, into one team's keys), and switched off for comparison."""
    import copy
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dancer", "scene.gltf")
    scene = scenes.gltf_model(path, 1280, 720)
    if variant == "greater":
        for d in scene.draws:
            d.depth_compare = scenes.CMP_GREATER
        scene.clear_depth = 0.0
    elif variant in ("two_states", "less_then_greater"):
        second = copy.copy(scene.draws[0])
        second.object = scenes.object_ubo(scenes.trs((0.8, 0.8, 0.8), scenes.quat_axis_angle((0.0, 1.0, 0.0), 2.0), (0.3, 0.1, 0.2)))
        # a different depth state: the scope is cut into two segments.  GREATER behind LESS is the case where the loaded
        # depth (<= the 1.0 clear) is WORSE than the cleared key of the second segment: only one team may start from it
        second.depth_compare = scenes.CMP_LESS_OR_EQUAL if variant == "two_states" else scenes.CMP_GREATER
        scene.draws.append(second)
    else:
        monkeypatch.setenv("MIRHI_RASTER_TEAMS", "1")
    out, ref = _render_both(mirhi, oracle, device, scene, want_depth=True)
    _check(out, ref, f"{scene.name}-{variant}", depth=True)


def test_mesh_heap_overflows_the_per_xcd_bins(mirhi, oracle, device, scenes):
    """20,000 lit triangles heaped into a 2x2-tile corner of a 1080p frame: on average under 16 per tile, so the scope gets
    the concentrated-mesh mode (two teams per tile, one sub-bin of 256 records per XCD and tile) -- and every sub-bin of
    those tiles overflows into the big list.  The frame must still be the oracle's."""
    rng = np.random.default_rng(77)
    nt = 20000
    c = rng.uniform(0.005, 0.045, (nt, 1, 2))
    p = c + rng.normal(0, 0.006, (nt, 3, 2))
    z = rng.uniform(0.1, 0.9, (nt, 3, 1))
    pos = np.concatenate([p, z], axis=2).reshape(nt * 3, 3)
    nrm = rng.normal(0, 1, (nt * 3, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    tan = np.tile(np.float32([1, 0, 0, 1]), (nt * 3, 1))
    verts = scenes._pack_vertex48(pos, nrm, rng.uniform(0, 1, (nt * 3, 2)), tan)
    eye4 = np.eye(4, dtype=np.float32)
    d = scenes.DrawSpec(vertices=verts, stride=48, count=nt * 3, indices=np.arange(nt * 3, dtype=np.uint32),
                        program=scenes.PROGRAM_MODEL_FULL, cull_mode=scenes.CULL_NONE,
                        camera=scenes.camera_ubo(eye4, eye4, (0.0, 0.0, 3.0)), object=scenes.object_ubo(eye4),
                        light=scenes.light_ubo(direction=(0.3, -1.0, 0.2), intensity=1.5, num_point=1),
                        material=scenes.material_ubo((0.8, 0.6, 0.4, 1.0), 0.0, 0.4, 1.0),
                        point_lights=scenes.point_light((0.5, 0.5, 2.0), 10.0, (1.0, 1.0, 1.0), 3.0),
                        albedo_map=scenes.WHITE_1X1, normal_map=scenes.WHITE_1X1)
    scene = scenes.Scene("mesh-heap", 1920, 1080, [d], clear_color=(0.1, 0.1, 0.15, 1.0))
    res = mirhi.SceneResources(device, scene, mirhi.Format.R32G32B32A32_SFLOAT, want_prim=True, want_depth=True)
    fence = mirhi.Fence(device)
    res.render(); res.render(fence); fence.wait()
    out = res.read()
    big = device.stats().last_big_list
    fence.destroy(); res.destroy()
    assert big > 1000, f"expected the sub-bins to overflow into the big list, got {big} entries"
    ref = oracle.render(scene, want_bgra8=True)
    _check(out, ref, scene.name, depth=True)


def test_many_draws_equal_one_draw(mirhi, oracle, device, scenes):
    """The same 12,000 smooth-shaded triangles as one draw and as 1,000 draws of 12 (the per-primitive draw table, the
    waterfall over the draws present in a wave, draws padded to whole waves in the geometry kernel): identical frames, and
    the oracle's frame."""
    base = scenes.random_triangles(12000, 1280, 720, seed=11, rmin=2, rmax=20)
    v = base.draws[0].vertices.reshape(-1, 6).copy()
    v[:, 3:6] = np.random.default_rng(3).uniform(0, 1, v[:, 3:6].shape)          # not flat: the fragment program runs
    one = scenes.Scene("one-draw", 1280, 720, [scenes.DrawSpec(vertices=v, stride=24, count=v.shape[0], cull_mode=scenes.CULL_NONE)])
    many = scenes.Scene("many-draws", 1280, 720, [scenes.DrawSpec(vertices=v[i * 36:(i + 1) * 36].copy(), stride=24, count=36,
                                                                  cull_mode=scenes.CULL_NONE) for i in range(1000)])
    out1, ref = _render_both(mirhi, oracle, device, one, want_depth=True)
    outn, refn = _render_both(mirhi, oracle, device, many, want_depth=True)
    _check(out1, ref, one.name, depth=True)
    _check(outn, refn, many.name, depth=True)
    assert np.array_equal(out1["prim"], outn["prim"]) and np.array_equal(out1["depth"].view(np.uint32), outn["depth"].view(np.uint32))
    assert np.array_equal(out1["color"].view(np.uint32), outn["color"].view(np.uint32))


def test_maximum_target_size_8192(mirhi, oracle, device, scenes):
    """The largest frame the viewport range admits (+-8192 px, include/mirhi.h): 8192x8192 = 65,536 tiles, a 256 MB BGRA8
    target.  Triangles far beyond the guard band, off-screen ones, small ones in the far corner; winning primitive ids
    bit for bit, sRGB8 within 1 LSB."""
    scene = scenes.huge_triangle_case(8192, 8192)
    extra = scenes._tri_verts([(0.97, 0.97, 0.3), (0.9999, 0.97, 0.3), (0.97, 0.9999, 0.3),
                               (-1.0, 0.99, 0.2), (-0.99, 1.0, 0.2), (-1.0, 1.0, 0.2)])
    scene.draws.append(scenes.DrawSpec(vertices=extra, stride=24, count=6, cull_mode=scenes.CULL_NONE))
    res = mirhi.SceneResources(device, scene, mirhi.Format.B8G8R8A8_SRGB, want_prim=True)
    res.render()
    out = res.read()
    # the workspace grows with the tiles (one fixed 2 KB bin page + a 256-byte page-table row each: 151 MB here) and with the scene's
    # triangles, not with tiles x bin capacity (round 1: 65,536 tiles x 1024 records x 48 B = 3.2 GB for a 10k-triangle scene at this size)
    assert device.stats().workspace_bytes < 200e6
    res.destroy()
    ref = oracle.render(scene, want_bgra8=True)
    assert np.array_equal(out["prim"], ref["prim"])
    assert int(np.abs(out["color"].astype(np.int16) - ref["bgra8"].astype(np.int16)).max()) <= 1
    assert (ref["prim"][-300:, -300:] != 0xFFFFFFFF).any() and (ref["prim"][-100:, :100] != 0xFFFFFFFF).any()


def test_submission_order_does_not_change_depth(mirhi, device, scenes):
    """Size-independent property at BASELINE configs[1] size: with LESS and distinct depths the stored depth image is a
    function of the triangle SET; reversing the submission order must leave it bit-identical (and permute prim ids)."""
    scene = scenes.random_triangles()
    res = mirhi.SceneResources(device, scene, want_prim=True, want_depth=True)
    res.render()
    a = res.read()
    res.destroy()
    d = scene.draws[0]
    n = d.num_triangles
    d.vertices = np.ascontiguousarray(d.vertices.reshape(n, 3, 6)[::-1].reshape(n * 3, 6))
    res = mirhi.SceneResources(device, scene, want_prim=True, want_depth=True)
    res.render()
    b = res.read()
    res.destroy()
    assert np.array_equal(a["depth"].view(np.uint32), b["depth"].view(np.uint32))
    cov = a["prim"] != 0xFFFFFFFF
    assert np.array_equal(cov, b["prim"] != 0xFFFFFFFF)
    # z is constant per triangle and distinct between triangles (u32-resolution draws), so ids map exactly
    assert np.array_equal(a["prim"][cov], (n - 1 - b["prim"][cov].astype(np.int64)).astype(np.uint32))
    assert np.array_equal(a["color"], b["color"])


def test_repeated_frames_are_identical(mirhi, oracle, device, scenes):
    """The workspace (bins, counters) is re-armed by the kernels themselves; a frame rendered again and again on
    the same command buffer must not change by a single pixel (guards the counter re-arm ordering inside a workgroup)."""
    scene = scenes.random_triangles(20000, 1920, 1080, seed=77)
    ref = oracle.render(scene, want_bgra8=False)
    res = mirhi.SceneResources(device, scene, want_prim=True)
    for it in range(25):
        res.render()
        out = res.read()
        assert np.array_equal(out["prim"], ref["prim"]), f"frame {it} differs from the oracle"
    res.destroy()


@pytest.mark.parametrize("shape", [(64, 64), (37, 21), (1, 9), (128, 2)])
def test_mip_chain_matches_the_stated_rule(mirhi, device, scenes, shape):
    """mirhi_image_generate_mips against the numpy restatement of its rule (scenes.mip_chain), level by level, read back
    through a 1:1 textured draw of each level's footprint is overkill: the chain is contiguous behind level 0, so the
    whole allocation is read through a wrapped view."""
    import ctypes as C
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    h, w = shape
    img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    t = mirhi.Image(device, w, h, mirhi.Format.R8G8B8A8_UNORM)
    t.upload(img)
    t.generate_mips()
    chain = scenes.mip_chain(img)
    assert t.mip_levels == len(chain)
    total = sum(l.size for l in chain)
    view = mirhi.Image(device, total // 4, 1, mirhi.Format.R8G8B8A8_UNORM, device_ptr=mirhi.lib().mirhi_image_device_ptr(t.handle))
    got = view.read().reshape(-1)
    want = np.concatenate([l.reshape(-1) for l in chain])
    assert np.array_equal(got, want)
    view.destroy()
    t.destroy()


# ---- the path bench.py times ---------------------------------------------------------
# B8G8R8A8_SRGB target, no primitive-id image, no depth image: raster_kernel<1,0,0,1> takes the flat-colour fast exit
# (packed colours from the geometry kernel, 32-bit-offset stores; interior-tile and edge-tile branches), with several
# command buffers in flight on separate queue lanes.  Spec: crates/rhi/src/swapchain.rs:561-570 (target format),
# crates/renderer/src/renderer.rs:479-488 (clear / store).
def _bench_like_frames(mirhi, scene, lanes, frames, ordinal=0):
    """Renders `frames` frames round-robin over `lanes` command buffers exactly as bench.py does; returns every target."""
    dev = mirhi.Device(ordinal)
    dev.set_queue_lanes(lanes)
    shared = {}

    def wrap_shared(device, usage, arr):     # geometry is uploaded once and shared by all frames in flight
        key = (usage, arr.ctypes.data, arr.size)
        if key not in shared:
            shared[key] = mirhi.Buffer.new_with_data(device, usage, arr)
        return shared[key]

    slots = [mirhi.SceneResources(dev, scene, mirhi.Format.B8G8R8A8_SRGB, wrap_buffers=wrap_shared) for _ in range(lanes)]
    for f in range(frames):
        slots[f % lanes].render()
    dev.wait_idle()
    outs = [s.color.read() for s in slots]
    seen = set()
    for s in slots:
        s.objs = [o for o in s.objs if not (id(o) in seen or seen.add(id(o)))]
        s.destroy()
    dev.destroy()
    return outs


@pytest.mark.parametrize("shape", ["c2_1080p", "edge_1900x1070", "small_640x360", "tiny_33x31"])
def test_bench_path_pixels(mirhi, oracle, device, scenes, shape):
    scene = {"c2_1080p": lambda: scenes.random_triangles(),
             "edge_1900x1070": lambda: scenes.random_triangles(10000, 1900, 1070, seed=0x5EED0102),
             "small_640x360": lambda: scenes.random_triangles(2000, 640, 360, seed=9),
             "tiny_33x31": lambda: scenes.random_triangles(40, 33, 31, seed=5, rmin=2, rmax=12)}[shape]()
    ref = oracle.render(scene, want_bgra8=True)
    outs = _bench_like_frames(mirhi, scene, lanes=4, frames=12)
    # the same scene through the general resolve (a bound primitive-id image disables the fast exit)
    res = mirhi.SceneResources(device, scene, mirhi.Format.B8G8R8A8_SRGB, want_prim=True)
    res.render()
    general = res.read()
    res.destroy()
    assert np.array_equal(general["prim"], ref["prim"])
    for i, out in enumerate(outs):
        d = np.abs(out.astype(np.int32) - ref["bgra8"].astype(np.int32))
        assert d.max() <= 1, f"{shape} lane {i}: sRGB8 differs from the oracle by {d.max()} LSB at {np.argwhere(d > 1)[:1]}"
        assert np.array_equal(out, general["color"]), f"{shape} lane {i}: fast exit differs from the general resolve at {np.argwhere(out != general['color'])[:1]}"


def test_bench_path_load_second_scope(mirhi, oracle, device, scenes):
    """A second scope with LOAD_OP_LOAD on the same sRGB8 target (fast exit, `color_load` branch): pixels the second scene
    does not cover keep the first scene's colour."""
    a = scenes.random_triangles(1500, 700, 390, seed=21)
    b = scenes.random_triangles(300, 700, 390, seed=22, rmin=4, rmax=30)
    ra, rb = oracle.render(a, want_bgra8=True), oracle.render(b, want_bgra8=True)
    first = mirhi.SceneResources(device, a, mirhi.Format.B8G8R8A8_SRGB)
    second = mirhi.SceneResources(device, b, mirhi.Format.B8G8R8A8_SRGB, color_image=first.color, color_load_op=mirhi.LoadOp.LOAD)
    for _ in range(3):
        first.render()
        second.render()
    out = second.read()["color"]
    covered = (rb["prim"] != 0xFFFFFFFF)[..., None]
    want = np.where(covered, rb["bgra8"], ra["bgra8"]).astype(np.int32)
    assert np.abs(out.astype(np.int32) - want).max() <= 1
    second.color = None          # shared with `first`
    second.destroy()
    first.destroy()


def test_bench_path_mixed_flat_and_smooth(mirhi, oracle, device, scenes):
    """Waves that hold a covered pixel WITHOUT a flat colour (smooth-shaded triangles) must leave the fast exit for the general
    loop; flat and smooth triangles are mixed in one draw."""
    scene = scenes.random_triangles(1200, 640, 360, seed=31)
    v = scene.draws[0].vertices.copy()
    rng = np.random.default_rng(7)
    smooth = rng.random(len(v) // 3) < 0.3                      # 30 % of the triangles get three different vertex colours
    cols = v.reshape(-1, 3, 6)[:, :, 3:6]
    cols[smooth] = rng.random((int(smooth.sum()), 3, 3), dtype=np.float32)
    scene.draws[0].vertices = v
    ref = oracle.render(scene, want_bgra8=True)
    outs = _bench_like_frames(mirhi, scene, lanes=2, frames=4)
    for out in outs:
        assert np.abs(out.astype(np.int32) - ref["bgra8"].astype(np.int32)).max() <= 1


@pytest.mark.parametrize("program", ["triangle", "model_full"])
def test_batched_submit_matches_oracle(mirhi, oracle, scenes, program):
    """Several command buffers in one mirhi_queue_submit (vkQueueSubmit with several command buffers, renderer.rs:407-424): frames of
    the same shape run as ONE batch of launches (vertex_kernel_batch / geometry_kernel_batch / raster_kernel_batch).  Every frame is
    a different scene of that shape and must come out as the oracle's; a submit that cannot be batched (different target sizes)
    falls back to scope-by-scope launches with the same result."""
    dev = mirhi.Device(0)
    dev.set_queue_lanes(2)
    if program == "triangle":
        made = [scenes.random_triangles(1500 + 300 * i, 640, 360, seed=40 + i) for i in range(5)]
        fmt = mirhi.Format.B8G8R8A8_SRGB
    else:
        made = [scenes.displaced_sphere(24 + 4 * i, 17 + 2 * i, 320, 200, seed=3 + i) for i in range(4)]
        fmt = mirhi.Format.R32G32B32A32_SFLOAT
    res = [mirhi.SceneResources(dev, sc, fmt, want_prim=(program != "triangle")) for sc in made]
    fence = mirhi.Fence(dev)
    for _ in range(3):                                   # re-submitted batches re-arm their own counters
        dev.submit([r.cmd for r in res], fence)
        fence.wait()
    for r, sc in zip(res, made):
        out = r.read()
        ref = oracle.render(sc, want_bgra8=True)
        if program == "triangle":
            assert np.abs(out["color"].astype(np.int32) - ref["bgra8"].astype(np.int32)).max() <= 1, sc.name
        else:
            _check(out, ref, sc.name)
    # mixed shapes: not batchable, same results; and a batch followed by single submits of its members on their own lanes
    odd = mirhi.SceneResources(dev, scenes.random_triangles(700, 333, 222, seed=77), fmt)
    dev.submit([res[0].cmd, odd.cmd, res[1].cmd], fence)
    fence.wait()
    res[1].render(); res[0].render()
    ref_odd = oracle.render(odd.scene, want_bgra8=True)
    got = odd.read()["color"]
    if program == "triangle":
        assert np.abs(got.astype(np.int32) - ref_odd["bgra8"].astype(np.int32)).max() <= 1
        assert np.abs(res[0].read()["color"].astype(np.int32) - oracle.render(made[0], want_bgra8=True)["bgra8"].astype(np.int32)).max() <= 1
    for r in res + [odd]:
        r.destroy()
    fence.destroy()
    dev.destroy()


@pytest.mark.parametrize("layout", ["bands", "interleaved"])
@pytest.mark.parametrize("config", ["c4", "c5"])
def test_full_size_eight_way_tile_row_split_equals_the_unsplit_frame(mirhi, scenes, config, layout):
    """BASELINE configs[3] and [4] as they are defined: 3840x2160 split by tile rows across 8 ranks -- 68 tile rows: seven bands of 9 and one
    of 5 (288 pixel rows each, 160 in the last), or interleaved rows r, r + 8, ... (9 tile rows for ranks 0-3, 8 for ranks 4-7).  Each rank's
    rows, rendered here one after the other into one sRGB8 target through the bench path (no primitive-id image), must tile the frame exactly:
    byte for byte the unsplit render (which the full-size tests above compare with the oracle).  Rows of other ranks stay untouched."""
    from renderer_rs_amd import multigpu
    scene = scenes.heightfield_grid() if config == "c4" else scenes.box_hall()
    dev = mirhi.Device(0)
    whole = mirhi.SceneResources(dev, scene, mirhi.Format.B8G8R8A8_SRGB)
    whole.render()
    ref = whole.read()["color"].copy()
    whole.destroy()
    world = 8
    assert [multigpu.band_rows(scene.height, r, world) for r in (0, 6, 7)] == [(0, 288), (1728, 2016), (2016, 2160)]
    assert [multigpu.split_tile_rows(scene.height, r, world, "interleaved") for r in (0, 3, 4, 7)] == [(0, 8, 9), (3, 8, 9), (4, 8, 8), (7, 8, 8)]
    target = mirhi.Image(dev, scene.width, scene.height, mirhi.Format.B8G8R8A8_SRGB)
    expect = np.full((scene.height, scene.width, 4), 0xA5, dtype=np.uint8)
    target.upload(expect)
    for rank in range(world):
        dev.set_tile_split(rank, world, layout=layout)
        res = mirhi.SceneResources(dev, scene, mirhi.Format.B8G8R8A8_SRGB, color_image=target)
        res.render()
        dev.wait_idle()
        for r0, r1 in multigpu.owned_pixel_rows(scene.height, rank, world, layout):
            expect[r0:r1] = ref[r0:r1]
        assert np.array_equal(target.read(), expect), f"{config} {layout}: rank {rank}'s rows differ from the unsplit frame, or it wrote outside them"
        res.color = None
        res.destroy()
    assert np.array_equal(target.read(), ref)
    target.destroy()
    dev.set_tile_split(0, 1)
    dev.destroy()


def _k9_quad(scenes, tex, W, H, su, sv):
    """The oracle KATs' viewport-filling textured quad (tests/test_oracle_kats.py K8 / K9) with separate uv scales."""
    pos = np.array([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0]], dtype=np.float32)
    verts = np.zeros((4, 12), dtype=np.float32)
    verts[:, 0:3] = pos; verts[:, 3:6] = [0, 0, 1]; verts[:, 8:12] = [1, 0, 0, 1]
    verts[:, 6:8] = np.array([[0, 0], [su, 0], [su, sv], [0, sv]], dtype=np.float32)
    idx = np.array([0, 1, 2, 0, 2, 3], dtype=np.uint32)
    ident = np.eye(4, dtype=np.float32)
    d = scenes.DrawSpec(vertices=verts, stride=48, count=6, indices=idx, program=scenes.PROGRAM_MODEL_FULL,
                        cull_mode=scenes.CULL_NONE, depth_test=False, depth_write=False,
                        camera=scenes.camera_ubo(ident, ident, (0.0, 0.0, 1.0)), object=scenes.object_ubo(ident),
                        light=scenes.light_ubo(intensity=0.0), material=scenes.material_ubo((1.0, 1.0, 1.0, 1.0), 0.0, 1.0, 1.0),
                        albedo_map=tex, normal_map=scenes.WHITE_1X1)
    return scenes.Scene("k9", W, H, [d])


@pytest.mark.parametrize("max_aniso,resolved", [(1, False), (4, False), (8, True), (16, True)])
def test_anisotropic_filter_kat(mirhi, oracle, device, scenes, max_aniso, resolved):
    """K9 on the HIP path: 8 texels per pixel along x, 1 along y, 1-texel stripes along y.  Trilinear (and N < 8) blurs them to
    grey; N = 8 taps along x at level 0 keep them -- and the image equals the oracle's within the float tolerance."""
    n = W = H = 64
    yy, _ = np.mgrid[0:n, 0:n]
    tex = np.zeros((n, n, 4), dtype=np.uint8)
    tex[..., 0:3] = (255 * (yy & 1))[..., None]
    tex[..., 3] = 255
    scene = _k9_quad(scenes, scenes.Texture(tex, mips=True, max_anisotropy=max_aniso), W, H, 8.0, 1.0)
    out, ref = _render_both(mirhi, oracle, device, scene)
    _check(out, ref, f"k9-aniso{max_aniso}")
    lum = out["color"][8:56, 8:56, 0] / 0.03
    if resolved:
        assert lum.min() < 1e-4 and lum.max() > 0.9999
    else:
        assert np.allclose(lum, 128.0 / 255.0, atol=1e-4)


def test_anisotropic_filter_isotropic_footprint_is_trilinear(mirhi, device, scenes):
    """N = ceil(Pmax / Pmin) = 1 on an isotropic footprint: one tap at the centre at the trilinear lambda -- the frame is the
    trilinear frame bit for bit, whatever max_anisotropy allows (the N the HIP path computes is an exact integer, not an estimate)."""
    rng = np.random.default_rng(9)
    noise = rng.integers(0, 256, (64, 64, 4), dtype=np.uint8)
    frames = []
    for a in (1, 16):
        res = mirhi.SceneResources(device, _k9_quad(scenes, scenes.Texture(noise, mips=True, max_anisotropy=a), 96, 96, 3.0, 3.0),
                                   mirhi.Format.R32G32B32A32_SFLOAT, want_prim=True)
        res.render()
        frames.append(res.read()["color"].copy())
        res.destroy()
    assert np.array_equal(frames[0], frames[1])


@pytest.mark.parametrize("teams", ["1", "2"])
@pytest.mark.parametrize("state", ["less", "greater", "always_nowrite", "equal_nowrite", "ordered_env"])
def test_alpha_masked_materials_every_resolve_path(mirhi, oracle, device, scenes, monkeypatch, teams, state):
    """Alpha-masked Cook-Torrance draws (`discard`, model_pbr.hlsl:176-179) through each way the library resolves them: the masked
    raster variants (plain key: LESS; generic key: GREATER; no depth write: later primitive wins), one and two teams per tile,
    small boxes walked by the lane that built them and large ones pixel-parallel (the case has both) -- and the ordered resolve
    where the key cannot decide (a predicate depth state) or where MIRHI_MASKED_ORDERED asks for it."""
    monkeypatch.setenv("MIRHI_RASTER_TEAMS", teams)
    if state == "ordered_env":
        monkeypatch.setenv("MIRHI_MASKED_ORDERED", "1")
    scene = scenes.SMALL_CASES["alpha_mask"]()
    for d in scene.draws:
        if state == "greater":
            d.depth_compare = scenes.CMP_GREATER
        elif state == "always_nowrite":
            d.depth_compare, d.depth_write = scenes.CMP_ALWAYS, False
        elif state == "equal_nowrite" and d is not scene.draws[0]:
            d.depth_compare, d.depth_write = scenes.CMP_LESS_OR_EQUAL, False
    if state == "greater":
        scene.clear_depth = 0.0
    out, ref = _render_both(mirhi, oracle, device, scene, want_depth=True)
    _check(out, ref, f"alpha-mask-{state}-teams{teams}", depth=True)
    assert (ref["prim"] != 0xFFFFFFFF).sum() > 5000


def test_alpha_masked_mesh_of_small_triangles(mirhi, oracle, device, scenes):
    """A finely tessellated masked sheet (every record takes the triangle-parallel walk) over an opaque scene, mip-free cut-out discs."""
    n = 32
    yy, xx = np.mgrid[0:n, 0:n]
    leaf = np.zeros((n, n, 4), dtype=np.uint8)
    leaf[..., 0] = 60; leaf[..., 1] = 150 + (xx % 8) * 8; leaf[..., 2] = 40
    leaf[..., 3] = np.clip(255 - np.hypot((xx % 16) - 7.5, (yy % 16) - 7.5) * 30, 0, 255).astype(np.uint8)
    W, H = 320, 200
    grid = scenes.heightfield_grid(48, 40, W, H).draws[0]
    verts = np.ascontiguousarray(grid.vertices).view(np.float32).reshape(-1, 12).copy()
    verts[:, 6:8] *= 6.0
    view, proj, cam = scenes.default_camera(W, H, eye=(0.0, 0.0, 5.0))
    masked = scenes.DrawSpec(vertices=verts, stride=48, count=grid.count, indices=grid.indices, program=scenes.PROGRAM_MODEL_PBR,
                             cull_mode=scenes.CULL_NONE, camera=cam, object=grid.object,
                             light=scenes.light_ubo(direction=(0.2, -0.6, -0.8), intensity=1.2, color=(1.0, 1.0, 1.0)),
                             material=scenes.pbr_material_ubo((1.0, 1.0, 1.0, 1.0), 0.0, 0.6, 1.0, alpha_cutoff=0.5, has_base_color=True),
                             albedo_map=scenes.Texture(leaf), alpha_test=True)
    opaque = scenes.random_triangles(200, W, H, seed=5, rmin=4, rmax=40).draws[0]
    scene = scenes.Scene("masked-sheet", W, H, [opaque, masked], clear_color=(0.1, 0.1, 0.15, 1.0))
    out, ref = _render_both(mirhi, oracle, device, scene, want_depth=True)
    _check(out, ref, scene.name, depth=True)
    kept = (ref["prim"] >= 200) & (ref["prim"] != 0xFFFFFFFF)
    assert 0.03 < kept.mean() < 0.8         # the discs keep about a fifth of the sheet, holes show what lies behind


@pytest.mark.parametrize("waves", [8, 16])
@pytest.mark.parametrize("case", ["c3", "c3_greater", "c3_two_states", "sphere_small", "textured", "pbr", "mips", "aniso", "multi_draw", "dancer", "dancer_tex", "depth_image",
                                  "dancer_xcd_bins", "dancer_tex_xcd_bins", "heap_xcd_bins"])
def test_wide_mesh_variants(mirhi, oracle, device, scenes, case, waves, monkeypatch):
    """raster_kernel_wide (eight / sixteen waves per tile with two / one 8x8 blocks each, 512 / 1024 records staged per pass; the host picks
    the first for mesh scopes whose triangles sit in few tiles) forced with MIRHI_RASTER_WIDE -- and, for the two-team candidates, MIRHI_RASTER_TEAMS=1 so that
    the scope is eligible: plain and generic depth keys, two segments of one scope, the full-featured programs, a stored depth image."""
    import copy
    import os
    monkeypatch.setenv("MIRHI_RASTER_WIDE", str(waves))
    if not case.endswith("xcd_bins"):                    # (*_xcd_bins: the concentrated-mesh mode keeps its eight per-XCD lists per tile, read by the wide variant)
        monkeypatch.setenv("MIRHI_RASTER_TEAMS", "1")
    monkeypatch.setenv("MIRHI_TP_DENSITY", "0")          # (every mesh scope gets the triangle-parallel path the wide variant builds on)
    dancer = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dancer", "scene.gltf")
    depth = False
    if case.startswith("c3"):
        scene = scenes.displaced_sphere()
        if case == "c3_greater":
            for d in scene.draws:
                d.depth_compare = scenes.CMP_GREATER
            scene.clear_depth = 0.0
        elif case == "c3_two_states":
            second = copy.copy(scene.draws[0])
            second.object = scenes.object_ubo(scenes.trs((0.8, 0.8, 0.8), scenes.quat_axis_angle((0.0, 1.0, 0.0), 2.0), (0.3, 0.1, 0.2)))
            second.depth_compare = scenes.CMP_LESS_OR_EQUAL
            scene.draws.append(second)
        depth = True
    elif case in ("dancer", "dancer_xcd_bins"):
        scene, depth = scenes.gltf_model(dancer), True
    elif case in ("dancer_tex", "dancer_tex_xcd_bins"):
        scene, depth = scenes.gltf_model(dancer, program=scenes.PROGRAM_MODEL_PBR, textures=True), True
    elif case == "heap_xcd_bins":                        # 20,000 lit triangles heaped into a 2x2-tile corner: every per-XCD list overflows into the big list
        rng = np.random.default_rng(77)
        nt = 20000
        c = rng.uniform(0.005, 0.045, (nt, 1, 2))
        p = c + rng.normal(0, 0.006, (nt, 3, 2))
        z = rng.uniform(0.1, 0.9, (nt, 3, 1))
        pos = np.concatenate([p, z], axis=2).reshape(nt * 3, 3)
        nrm = rng.normal(0, 1, (nt * 3, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        verts = scenes._pack_vertex48(pos, nrm, rng.uniform(0, 1, (nt * 3, 2)), np.tile(np.float32([1, 0, 0, 1]), (nt * 3, 1)))
        eye4 = np.eye(4, dtype=np.float32)
        d0 = scenes.DrawSpec(vertices=verts, stride=48, count=nt * 3, indices=np.arange(nt * 3, dtype=np.uint32),
                             program=scenes.PROGRAM_MODEL_FULL, cull_mode=scenes.CULL_NONE,
                             camera=scenes.camera_ubo(eye4, eye4, (0.0, 0.0, 3.0)), object=scenes.object_ubo(eye4),
                             light=scenes.light_ubo(direction=(0.3, -1.0, 0.2), intensity=1.5, num_point=1),
                             material=scenes.material_ubo((0.8, 0.6, 0.4, 1.0), 0.0, 0.4, 1.0),
                             point_lights=scenes.point_light((0.5, 0.5, 2.0), 10.0, (1.0, 1.0, 1.0), 3.0),
                             albedo_map=scenes.WHITE_1X1, normal_map=scenes.WHITE_1X1)
        scene, depth = scenes.Scene("mesh-heap", 1920, 1080, [d0], clear_color=(0.1, 0.1, 0.15, 1.0)), True
    elif case == "depth_image":
        scene, depth = scenes.displaced_sphere(64, 47, 640, 360, seed=9), True
    else:
        scene = scenes.SMALL_CASES[case]()
        depth = any(d.depth_test for d in scene.draws)
    out, ref = _render_both(mirhi, oracle, device, scene, want_depth=depth)
    _check(out, ref, f"{scene.name}-wide{waves}-{case}", depth=depth)


def test_wide_variant_is_chosen_from_the_busy_tile_count(mirhi, oracle, scenes):
    """The host's own choice: a mesh in a part of the frame reports few busy tiles, the plan is rebuilt with the wide variant after the
    feedback has arrived (and stays the oracle's frame through the switch); a mesh that covers the frame keeps four waves per tile."""
    dev = mirhi.Device(0)
    for scene, expect_wide in ((scenes.displaced_sphere(), True), (scenes.heightfield_grid(300, 200, 1920, 1080), False)):
        res = mirhi.SceneResources(dev, scene, mirhi.Format.R32G32B32A32_SFLOAT, want_prim=True)
        ref = oracle.render(scene, want_bgra8=False)
        f = mirhi.Fence(dev)
        for it in range(5):
            res.render(f); f.wait(); f.reset()
            assert np.array_equal(res.read()["prim"], ref["prim"]), f"{scene.name}: frame {it}"
        dev.reset_kernel_times(); dev.set_profiling(mirhi.Profile.TIMING)
        res.render(f); f.wait(); f.reset()
        ms, n = dev.kernel_time(mirhi.Kernel.NAMES.index("raster"))
        dev.set_profiling(0)
        print(f"{scene.name}: raster {1e3 * ms / max(1, n):.1f} us with the host's choice (wide expected: {expect_wide})")
        res.destroy(); f.destroy()
    dev.destroy()
