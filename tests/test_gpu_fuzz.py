"""Differential fuzzing of the HIP path against the oracle: random pipeline state x random geometry, including
triangles that cross the near plane / guard band, sub-pixel slivers, shared vertices through u16 / u32 indices,
partial viewports and scissors, every supported depth compare op and cull mode.  Integer outputs must be bit-exact."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _random_scene(scenes, seed):
    rng = np.random.default_rng(seed)
    W, H = int(rng.integers(33, 300)), int(rng.integers(17, 220))
    draws = []
    for _ in range(int(rng.integers(1, 4))):
        kind = rng.integers(0, 3)
        cull = int(rng.integers(0, 3))
        front = int(rng.integers(0, 2))
        cmp_ = [scenes.CMP_LESS, scenes.CMP_LESS_OR_EQUAL, scenes.CMP_GREATER, scenes.CMP_GREATER_OR_EQUAL][int(rng.integers(0, 4))]
        if rng.random() < 0.3:
            vp = (float(rng.integers(-20, 20)), float(rng.integers(-20, 20)), float(W + rng.integers(-30, 40)), float(H + rng.integers(-20, 30)) or 1.0,
                  float(rng.uniform(0, 0.3)), float(rng.uniform(0.6, 1.0)))      # (a negative height flips y: legal; 0 is not)
        else:
            vp = None
        sc = (int(rng.integers(0, W // 3)), int(rng.integers(0, H // 3)), int(rng.integers(W // 3, W)), int(rng.integers(H // 3, H))) if rng.random() < 0.3 else None
        if kind == 0:      # clip-space triangles with wild extents, some behind the camera (w varies through z only for MODEL)
            n = int(rng.integers(1, 200))
            scale = 10 ** rng.uniform(-2.5, 1.0)
            c = rng.uniform(-1.2, 1.2, (n, 1, 2))
            p = c + rng.normal(0, scale, (n, 3, 2))
            z = rng.uniform(-0.2, 1.2, (n, 3, 1)) if rng.random() < 0.5 else np.repeat(rng.uniform(0, 1, (n, 1, 1)), 3, axis=1)
            col = rng.uniform(0, 1, (n, 3, 3))
            verts = np.concatenate([p, z, col], axis=2).astype(np.float32).reshape(n * 3, 6)
            d = scenes.DrawSpec(vertices=verts, stride=24, count=3 * n, program=scenes.PROGRAM_TRIANGLE)
        else:              # indexed lit mesh through a perspective camera placed so that some triangles cross the near plane
            nu, nv = int(rng.integers(2, 14)), int(rng.integers(2, 11))
            s = scenes.displaced_sphere(nu, nv, W, H, seed=int(rng.integers(1, 1 << 30)),
                                        program=scenes.PROGRAM_MODEL if kind == 1 else scenes.PROGRAM_MODEL_FULL).draws[0]
            eye = (float(rng.uniform(-1.5, 1.5)), float(rng.uniform(-1.0, 1.0)), float(rng.uniform(0.3, 4.0)))
            view = scenes.look_at_rh(eye, (0.0, 0.0, 0.0), (0.0, 1.0, 0.0))
            proj = scenes.projection_vulkan(math.radians(float(rng.uniform(30, 100))), W / H, float(rng.uniform(0.05, 0.6)), float(rng.uniform(3, 50)))
            s.camera = scenes.camera_ubo(view, proj, eye)
            model = scenes.trs(tuple(rng.uniform(0.5, 1.5, 3)), scenes.quat_axis_angle(rng.normal(size=3), float(rng.uniform(0, 6))), tuple(rng.uniform(-0.5, 0.5, 3)))
            s.object = scenes.object_ubo(model)
            if rng.random() < 0.5:
                s.indices = s.indices.astype(np.uint16)
            if rng.random() < 0.4:   # draw a sub-range through first_index / vertex_offset
                pad = int(rng.integers(1, 7))
                s.vertices = np.concatenate([np.zeros((pad, 12), dtype=np.float32), s.vertices], axis=0)
                s.vertex_offset = pad
                s.first = 3 * int(rng.integers(0, s.count // 6))
                s.count = 3 * int(rng.integers(1, (s.indices.size - s.first) // 3 + 1))
            d = s
        d.cull_mode, d.front_face, d.depth_compare, d.viewport, d.scissor = cull, front, cmp_, vp, sc
        draws.append(d)
    for d in draws:          # one rendering scope shares one depth state
        d.depth_compare = draws[0].depth_compare
    greater = draws[0].depth_compare in (scenes.CMP_GREATER, scenes.CMP_GREATER_OR_EQUAL)
    return scenes.Scene(f"fuzz-{seed}", W, H, draws, clear_color=tuple(rng.uniform(0, 1, 3)) + (1.0,),
                        clear_depth=float(rng.uniform(0.0, 0.4)) if greater else float(rng.uniform(0.6, 1.0)))


@pytest.mark.parametrize("seed", range(40))
def test_fuzz_against_oracle(mirhi, oracle, device, scenes, seed):
    scene = _random_scene(scenes, 1000 + seed)
    res = mirhi.SceneResources(device, scene, want_prim=True, want_depth=True)
    res.render()
    out = res.read()
    res.destroy()
    ref = oracle.render(scene, want_bgra8=False)
    diff = out["prim"] != ref["prim"]
    assert not diff.any(), f"{scene.name}: {int(diff.sum())} pixels differ in winning primitive (first {np.argwhere(diff)[0]})"
    cov = ref["prim"] != 0xFFFFFFFF
    assert np.array_equal(out["depth"].view(np.uint32)[cov], ref["depth"].view(np.uint32)[cov]), f"{scene.name}: depth bits differ"
    a, b = out["color"], ref["rgba"]
    nan = np.isnan(b)
    assert np.array_equal(np.isnan(a), nan)
    err = np.abs(np.where(nan, 0, a) - np.where(nan, 0, b)) / np.maximum(1.0, np.abs(np.where(nan, 0, b)))
    assert err.max() < 1e-4, f"{scene.name}: max |dRGBA| {err.max()}"


@pytest.mark.parametrize("seed", range(8))
def test_hostile_vertex_values(mirhi, oracle, device, scenes, seed):
    """NaN / Inf / huge / denormal coordinates and colours: no hang, no fault, and the same pixels as the oracle."""
    rng = np.random.default_rng(77 + seed)
    n = 300
    verts = scenes.random_triangles(n, 200, 150, seed=500 + seed, rmin=2, rmax=60).draws[0].vertices.copy()
    specials = np.array([np.nan, np.inf, -np.inf, 1e30, -1e30, 1e-42, -0.0, 3.4e38], dtype=np.float32)
    for _ in range(120):
        verts[int(rng.integers(0, 3 * n)), int(rng.integers(0, 6))] = specials[int(rng.integers(0, len(specials)))]
    d = scenes.DrawSpec(vertices=verts, stride=24, count=3 * n, cull_mode=scenes.CULL_NONE)
    scene = scenes.Scene(f"hostile-{seed}", 200, 150, [d])
    res = mirhi.SceneResources(device, scene, want_prim=True, want_depth=True)
    f = mirhi.Fence(device)
    res.render(f)
    f.wait(5_000_000_000)
    out = res.read()
    res.destroy()
    f.destroy()
    ref = oracle.render(scene, want_bgra8=False)
    assert np.array_equal(out["prim"], ref["prim"])
    cov = ref["prim"] != 0xFFFFFFFF
    assert np.array_equal(out["depth"].view(np.uint32)[cov], ref["depth"].view(np.uint32)[cov])
    assert np.array_equal(np.isnan(out["color"]), np.isnan(ref["rgba"]))


def _random_pbr_scene(scenes, seed):
    """Cook-Torrance draws with random factors, texture switches and lights (pixel/model_pbr.hlsl)."""
    rng = np.random.default_rng(seed)
    W, H = int(rng.integers(40, 260)), int(rng.integers(30, 200))

    def tex(n):        # square or not, a third with a mip chain (trilinear), a third sRGB-encoded
        return scenes.Texture(rng.integers(0, 256, (n, int(rng.integers(1, 24)) if rng.random() < 0.3 else n, 4), dtype=np.uint8),
                              mips=bool(rng.random() < 0.35), srgb=bool(rng.random() < 0.35),
                              max_anisotropy=int(rng.choice([1, 1, 2, 4, 7, 16])))

    eye = (float(rng.uniform(-1.0, 1.0)), float(rng.uniform(-0.8, 0.8)), float(rng.uniform(1.2, 4.0)))
    view = scenes.look_at_rh(eye, (0.0, 0.0, 0.0), (0.0, 1.0, 0.0))
    proj = scenes.projection_vulkan(math.radians(float(rng.uniform(35, 90))), W / H, 0.1, 50.0)
    cam = scenes.camera_ubo(view, proj, eye)
    npnt, nspt = int(rng.integers(0, 4)), int(rng.integers(0, 3))
    points = b"".join(scenes.point_light(tuple(rng.uniform(-3, 3, 3)), float(rng.uniform(2, 12)), tuple(rng.uniform(0, 1, 3)),
                                         float(rng.uniform(0, 8))) for _ in range(npnt))
    spots = b"".join(scenes.spot_light(tuple(rng.uniform(-3, 3, 3)), float(rng.uniform(0.85, 0.99)), tuple(rng.normal(size=3)),
                                       float(rng.uniform(0.5, 0.84)), tuple(rng.uniform(0, 1, 3)), float(rng.uniform(0, 10)))
                     for _ in range(nspt))
    light = scenes.light_ubo(direction=tuple(rng.normal(size=3)), intensity=float(rng.uniform(0, 3)), color=tuple(rng.uniform(0, 1, 3)),
                             num_point=npnt, num_spot=nspt)
    draws = []
    for _ in range(int(rng.integers(1, 4))):
        s = scenes.displaced_sphere(int(rng.integers(3, 14)), int(rng.integers(3, 11)), W, H, seed=int(rng.integers(1, 1 << 30)),
                                    program=scenes.PROGRAM_MODEL_PBR).draws[0]
        flags = rng.random(5) < 0.5
        alpha = float(rng.uniform(0.2, 1.0))
        # cutoff only where one decision covers the draw: untextured (either side), or textured with cutoff <= 0
        cutoff = float(rng.choice([0.0, alpha - 0.1, alpha + 0.1])) if not flags[0] else float(rng.choice([0.0, -0.5]))
        s.material = scenes.pbr_material_ubo(tuple(rng.uniform(0, 1, 3)) + (alpha,), float(rng.uniform(0, 1)), float(rng.uniform(0, 1)),
                                             float(rng.uniform(0.3, 1)), normal_scale=float(rng.uniform(0.2, 1.5)),
                                             emissive=tuple(rng.uniform(0, 0.5, 3)), alpha_cutoff=cutoff,
                                             has_base_color=flags[0], has_normal=flags[1], has_metallic_roughness=flags[2],
                                             has_occlusion=flags[3], has_emissive=flags[4])
        s.albedo_map = tex(int(rng.integers(1, 20))) if flags[0] and rng.random() < 0.9 else None
        s.normal_map = tex(int(rng.integers(1, 20))) if flags[1] and rng.random() < 0.9 else None
        s.metallic_roughness_map = tex(int(rng.integers(1, 20))) if flags[2] else None
        s.occlusion_map = tex(int(rng.integers(1, 20))) if flags[3] else None
        s.emissive_map = tex(int(rng.integers(1, 20))) if flags[4] else None
        s.camera, s.light, s.point_lights, s.spot_lights = cam, light, points, spots
        s.object = scenes.object_ubo(scenes.trs(tuple(rng.uniform(0.4, 1.2, 3)), scenes.quat_axis_angle(rng.normal(size=3), float(rng.uniform(0, 6))),
                                                tuple(rng.uniform(-0.8, 0.8, 3))))
        s.cull_mode = int(rng.integers(0, 3))
        draws.append(s)
    if rng.random() < 0.5:      # mix with a TRIANGLE-program draw: the variant that carries every program
        n = int(rng.integers(1, 30))
        p = rng.uniform(-1, 1, (n, 1, 2)) + rng.normal(0, 0.2, (n, 3, 2))
        verts = np.concatenate([p, rng.uniform(0, 1, (n, 3, 1)), rng.uniform(0, 1, (n, 3, 3))], axis=2).astype(np.float32).reshape(n * 3, 6)
        draws.append(scenes.DrawSpec(vertices=verts, stride=24, count=3 * n, program=scenes.PROGRAM_TRIANGLE, cull_mode=scenes.CULL_NONE))
    # a third of the scenes carry an alpha-masked material: a textured draw whose texel alpha straddles the cutoff, drawn by a
    # pipeline with fragment_discard_enable (per-fragment discard, ordered resolve).  Bilinear textures only: under a mip chain the
    # level-of-detail comes from v_log_f32 here and log2f in the oracle, and a discard decision has no tolerance.
    rng2 = np.random.default_rng(seed ^ 0xA17A)
    masked = [d for d in draws if d.program == scenes.PROGRAM_MODEL_PBR and d.albedo_map is not None and not d.albedo_map.mips
              and np.frombuffer(d.material, dtype=np.uint32)[12] != 0]
    if masked and rng2.random() < 0.33:
        d = masked[int(rng2.integers(0, len(masked)))]
        mat = np.frombuffer(d.material, dtype=np.float32).copy()
        mat[11] = mat[3] * float(rng2.uniform(0.2, 0.8))          # alphaCutoff @44 = a fraction of baseColorFactor.a @12
        d.material = mat.tobytes()
        d.alpha_test = True
    return scenes.Scene(f"fuzz-pbr-{seed}", W, H, draws, clear_color=tuple(rng.uniform(0, 1, 3)) + (1.0,))


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_pbr_against_oracle(mirhi, oracle, device, scenes, seed):
    scene = _random_pbr_scene(scenes, 3000 + seed)
    res = mirhi.SceneResources(device, scene, want_prim=True, want_depth=True)
    res.render()
    out = res.read()
    res.destroy()
    ref = oracle.render(scene, want_bgra8=False)
    assert np.array_equal(out["prim"], ref["prim"]), f"{scene.name}: winning primitive differs"
    cov = ref["prim"] != 0xFFFFFFFF
    assert np.array_equal(out["depth"].view(np.uint32)[cov], ref["depth"].view(np.uint32)[cov]), f"{scene.name}: depth bits differ"
    a, b = out["color"], ref["rgba"]
    nan = np.isnan(b)
    assert np.array_equal(np.isnan(a), nan)
    err = np.abs(np.where(nan, 0, a) - np.where(nan, 0, b)) / np.maximum(1.0, np.abs(np.where(nan, 0, b)))
    assert err.max() < 1e-4, f"{scene.name}: max |dRGBA| {err.max()}"


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_predicate_depth_states(mirhi, oracle, device, scenes, seed):
    """The random scenes of test_fuzz_against_oracle with the scope's depth state replaced by a predicate one: depth test
    without write for all eight compare ops, Equal and Always with write (DESIGN.md "Depth key")."""
    scene = _random_scene(scenes, 5000 + seed)
    rng = np.random.default_rng(7000 + seed)
    states = [(op, False) for op in range(8)] + [(scenes.CMP_EQUAL, True), (scenes.CMP_ALWAYS, True)]
    op, write = states[int(rng.integers(0, len(states)))]
    for d in scene.draws:
        d.depth_test, d.depth_write, d.depth_compare = True, write, op
    scene.clear_depth = float(rng.choice([0.0, 0.25, 0.5, 1.0]))
    res = mirhi.SceneResources(device, scene, want_prim=True, want_depth=True)
    res.render()
    out = res.read()
    res.destroy()
    ref = oracle.render(scene, want_bgra8=False)
    assert np.array_equal(out["prim"], ref["prim"]), f"{scene.name} op {op} write {write}: winning primitive differs"
    assert np.array_equal(out["depth"].view(np.uint32), ref["depth"].view(np.uint32)), f"{scene.name} op {op} write {write}: depth differs"


def _randomize_states(scenes, scene, rng):
    """Per-draw depth state (any compare op, test / write on or off) and, for a third of the draws, a random
    ColorBlendAttachment (every factor the pipeline accepts, every op, random write mask)."""
    factors = [f for f in range(15) if not (scenes.BF_CONSTANT_COLOR <= f <= scenes.BF_ONE_MINUS_CONSTANT_ALPHA)]
    for d in scene.draws:
        d.depth_test = bool(rng.random() < 0.8)
        d.depth_compare = int(rng.integers(0, 8))
        d.depth_write = bool(rng.random() < 0.6)
        if rng.random() < 0.35:
            d.blend = (int(rng.choice(factors)), int(rng.choice(factors)), int(rng.integers(0, 5)), int(rng.choice(factors)),
                       int(rng.choice(factors)), int(rng.integers(0, 5)), int(rng.integers(1, 16)))
    scene.clear_color = tuple(float(x) for x in rng.uniform(0, 1, 4))


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_mixed_depth_states_in_one_scope(mirhi, oracle, device, scenes, seed):
    """Every draw of the scope gets its own depth state and maybe a blend state: the scope is cut into segments at each
    change, blended / order-dependent ones go through the ordered kernel (DESIGN.md "Depth key", "Ordered segments"), and
    the result must still equal the oracle's fragment-by-fragment order."""
    scene = _random_scene(scenes, 9000 + seed)
    rng = np.random.default_rng(9500 + seed)
    _randomize_states(scenes, scene, rng)
    scene.clear_depth = float(rng.choice([0.0, 0.3, 0.7, 1.0]))
    res = mirhi.SceneResources(device, scene, want_prim=True, want_depth=True)
    res.render()
    out = res.read()
    res.destroy()
    ref = oracle.render(scene, want_bgra8=False)
    assert np.array_equal(out["prim"], ref["prim"]), f"{scene.name}: winning primitive differs"
    assert np.array_equal(out["depth"].view(np.uint32), ref["depth"].view(np.uint32)), f"{scene.name}: depth differs"
    a, b = out["color"], ref["rgba"]
    nan = np.isnan(b)
    assert np.array_equal(np.isnan(a), nan)
    with np.errstate(invalid="ignore"):      # equal infinities (feedback blend factors overflow on both sides alike) are no error
        err = np.where(a == b, 0.0, np.abs(np.where(nan, 0, a) - np.where(nan, 0, b)) / np.maximum(1.0, np.abs(np.where(nan, 0, b))))
    assert err.max() < 1e-4


@pytest.mark.parametrize("blend", [False, True])
def test_overlapping_pieces_of_one_clipped_triangle_under_always_with_write(mirhi, oracle, device, scenes, blend):
    """Found by tools/soak_fuzz.py (seed 522444): two clip pieces of ONE triangle cover the same pixel, and Always-with-write stored the depth of
    whichever record reached the tile first -- a result that changed from run to run.  Both sides now keep the nearer fragment of a primitive
    (raster_record's predicate mode; ordered_record for the blended segment); rendered several times, every frame equals the oracle."""
    from test_oracle_kats import overlapping_clip_pieces_scene
    bl = (scenes.BF_SRC_ALPHA, scenes.BF_ONE_MINUS_SRC_ALPHA, 0, scenes.BF_ONE, scenes.BF_ZERO, 0, 15) if blend else None
    scene = overlapping_clip_pieces_scene(scenes, scenes.CMP_ALWAYS, blend=bl)
    ref = oracle.render(scene, want_bgra8=False)
    assert (ref["prim"] == 0).sum() > 0
    for _ in range(6):
        res = mirhi.SceneResources(device, scene, want_prim=True, want_depth=True)
        res.render()
        out = res.read()
        res.destroy()
        assert np.array_equal(out["prim"], ref["prim"])
        assert np.array_equal(out["depth"].view(np.uint32), ref["depth"].view(np.uint32))
        assert np.abs(out["color"] - ref["rgba"]).max() < 1e-4
