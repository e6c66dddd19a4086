"""GPU tests of the drop-in boundary: same names, argument meaning and error behaviour as crates/rhi and the
crates/renderer draw-submit loop, exercised through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_buffer_semantics(mirhi, device):
    """crates/rhi/src/buffer.rs:149-293,345-417 (+ its tests :442-553)."""
    with pytest.raises(mirhi.RhiError) as e:
        mirhi.Buffer(device, mirhi.BufferUsage.Vertex, 0)
    assert e.value.variant == "InvalidHandle" and "Buffer size must be greater than 0" in e.value.message
    b = mirhi.Buffer(device, mirhi.BufferUsage.Vertex, 64)
    assert b.size() == 64 and b.usage() == mirhi.BufferUsage.Vertex
    b.write_data(0, b"")                                           # empty write is Ok (buffer.rs:248-250)
    b.write_data(16, np.arange(12, dtype=np.float32))
    with pytest.raises(mirhi.RhiError) as e:
        b.write_data(32, np.zeros(9, dtype=np.float32))
    assert "Write exceeds buffer size: offset 32 + data 36 > buffer 64" in e.value.message
    assert np.array_equal(b.read(16, 48).view(np.float32), np.arange(12, dtype=np.float32))
    b.destroy()
    s = mirhi.Buffer(device, mirhi.BufferUsage.Storage, 32)        # GpuOnly memory: not mapped (buffer.rs:86-100)
    with pytest.raises(mirhi.RhiError) as e:
        s.upload(np.zeros(4, dtype=np.uint32))
    assert "Buffer memory is not mapped" in e.value.message
    s.upload_via_staging(np.arange(8, dtype=np.uint32))
    assert np.array_equal(s.read(0, 32).view(np.uint32), np.arange(8, dtype=np.uint32))
    s.destroy()
    v = mirhi.Buffer.new_with_data(device, mirhi.BufferUsage.Index, np.arange(6, dtype=np.uint16))
    assert v.size() == 12
    v.destroy()


def _tri_builder(mirhi):
    return (mirhi.GraphicsPipelineBuilder().vertex_shader(mirhi.Program.TRIANGLE).fragment_shader(mirhi.Program.TRIANGLE)
            .vertex_binding(24).vertex_attributes((0, 12)).color_attachment_format(mirhi.Format.R32G32B32A32_SFLOAT))


def test_pipeline_build_validation_order_and_text(mirhi, device):
    """GraphicsPipelineBuilder::build (pipeline.rs:918-952): same checks, same order, same messages."""
    B = mirhi.GraphicsPipelineBuilder

    def err(builder):
        with pytest.raises(mirhi.RhiError) as e:
            builder.build(device)
        assert e.value.variant in ("PipelineError", "ShaderError")
        return e.value.message

    assert "Vertex shader is required" in err(B())
    assert "Fragment shader is required" in err(B().vertex_shader(mirhi.Program.TRIANGLE))
    assert "At least one color attachment format is required" in err(
        B().vertex_shader(mirhi.Program.TRIANGLE).fragment_shader(mirhi.Program.TRIANGLE))
    assert "Depth test or write is enabled but no depth attachment format is specified" in err(_tri_builder(mirhi))
    ok = _tri_builder(mirhi).depth_test_enable(False).depth_write_enable(False).cull_mode(mirhi.CullMode.NONE).build(device)
    ok.destroy()
    ok = _tri_builder(mirhi).depth_attachment_format(mirhi.Format.D32_SFLOAT).build(device)
    ok.destroy()
    b = _tri_builder(mirhi).depth_attachment_format(mirhi.Format.D32_SFLOAT)
    b.desc.blend_attachment_count = 2
    assert "Blend attachment count (2) must match color attachment count (1)" in err(b)
    # not implemented by the compute rasterizer: loud, never silently different
    assert "unsupported" in err(_tri_builder(mirhi).depth_attachment_format(mirhi.Format.D32_SFLOAT)
                                .color_blend_attachment(mirhi.BlendFactor.ConstantColor, mirhi.BlendFactor.Zero, mirhi.BlendOp.Add,
                                                        mirhi.BlendFactor.One, mirhi.BlendFactor.Zero, mirhi.BlendOp.Add))
    assert "unsupported" in err(_tri_builder(mirhi).depth_attachment_format(mirhi.Format.D32_SFLOAT).topology(mirhi.PrimitiveTopology.TriangleStrip))
    assert "unsupported" in err(_tri_builder(mirhi).depth_attachment_format(mirhi.Format.D32_SFLOAT).polygon_mode(mirhi.PolygonMode.Line))
    for ok in (_tri_builder(mirhi).depth_attachment_format(mirhi.Format.D32_SFLOAT).depth_compare_op(mirhi.CompareOp.Equal),
               _tri_builder(mirhi).depth_attachment_format(mirhi.Format.D32_SFLOAT).depth_write_enable(False).depth_compare_op(mirhi.CompareOp.NotEqual)):
        ok.build(device).destroy()          # predicate depth states are supported
    mixed = (B().vertex_shader(mirhi.Program.MODEL).fragment_shader(mirhi.Program.TRIANGLE).vertex_binding(48)
             .vertex_attributes((0, 12, 24, 32)).color_attachment_format(mirhi.Format.R32G32B32A32_SFLOAT)
             .depth_attachment_format(mirhi.Format.D32_SFLOAT))
    assert "does not produce the inputs" in err(mixed)


def test_fence_semantics(mirhi, device, scenes):
    """Fence::new(signaled) / wait / reset / is_signaled (sync.rs:168-298)."""
    f = mirhi.Fence(device, signaled=True)
    assert f.is_signaled()
    f.wait(0)
    f.reset()
    assert not f.is_signaled()
    with pytest.raises(mirhi.RhiError) as e:
        f.wait(1_000_000)                       # nothing submitted: times out like vkWaitForFences
    assert e.value.code == mirhi.TIMEOUT and "TIMEOUT" in e.value.message
    res = mirhi.SceneResources(device, scenes.hello_triangle(64, 64))
    res.render(f)
    f.wait()                                    # u64::MAX
    assert f.is_signaled()
    f.reset()
    assert not f.is_signaled()
    res.destroy()
    f.destroy()


def test_frames_in_flight_loop(mirhi, device, scenes):
    """Renderer::render_frame ordering (renderer.rs:367-449): wait fence -> reset -> record -> submit, 2 frames in flight."""
    scene = scenes.random_triangles(500, 320, 200, seed=3)
    frames = [mirhi.SceneResources(device, scene, want_prim=True) for _ in range(mirhi.MAX_FRAMES_IN_FLIGHT)]
    fences = [mirhi.Fence(device, signaled=True) for _ in frames]
    cur = 0
    for _ in range(7):
        fences[cur].wait()
        fences[cur].reset()
        frames[cur].record()                    # reset + begin + commands + end each frame, as the reference does
        frames[cur].render(fences[cur])
        cur = (cur + 1) % mirhi.MAX_FRAMES_IN_FLIGHT
    for f in fences:
        f.wait()
    a, b = frames[0].read(), frames[1].read()
    assert np.array_equal(a["prim"], b["prim"]) and np.array_equal(a["color"], b["color"])
    for r in frames:
        r.destroy()
    for f in fences:
        f.destroy()


def test_command_buffer_state_errors(mirhi, device):
    cmd = mirhi.CommandBuffer(device)
    with pytest.raises(mirhi.RhiError):
        cmd.end()                               # not recording
    with pytest.raises(mirhi.RhiError):
        cmd.draw(3)
    img = mirhi.Image(device, 32, 32, mirhi.Format.R32G32B32A32_SFLOAT)
    cmd.begin()
    with pytest.raises(mirhi.RhiError) as e:
        cmd.draw(3)
    assert "outside a rendering scope" in e.value.message
    cmd.begin_rendering(img)
    with pytest.raises(mirhi.RhiError) as e:
        cmd.draw(3)
    assert "no pipeline bound" in e.value.message
    with pytest.raises(mirhi.RhiError):
        cmd.end()                               # still inside the rendering scope
    cmd.end_rendering()
    cmd.end()
    fresh = mirhi.CommandBuffer(device)
    with pytest.raises(mirhi.RhiError):
        device.submit([fresh])                  # never recorded
    fresh.destroy()
    device.submit([cmd])                        # an empty scope just clears
    device.wait_idle()
    assert np.allclose(img.read(), [0.0, 0.0, 0.0, 1.0])  # ColorAttachment default clear (rendering.rs:108-112)
    with pytest.raises(mirhi.RhiError) as e:
        mirhi.Image(device, 0, 16, mirhi.Format.D32_SFLOAT)  # DepthBuffer::new rejects 0 (depth_buffer.rs:118-127)
    assert "must be greater than 0" in e.value.message
    cmd.destroy()
    img.destroy()


def test_draw_bounds_are_checked(mirhi, device):
    img = mirhi.Image(device, 32, 32, mirhi.Format.R32G32B32A32_SFLOAT)
    pipe = _tri_builder(mirhi).depth_test_enable(False).depth_write_enable(False).build(device)
    vb = mirhi.Buffer.new_with_data(device, mirhi.BufferUsage.Vertex, np.zeros((3, 6), dtype=np.float32))
    ib = mirhi.Buffer.new_with_data(device, mirhi.BufferUsage.Index, np.arange(3, dtype=np.uint32))
    cmd = mirhi.CommandBuffer(device)
    cmd.begin()
    cmd.begin_rendering(img)
    cmd.bind_pipeline(pipe)
    cmd.bind_vertex_buffers(0, [vb], [0])
    with pytest.raises(mirhi.RhiError) as e:
        cmd.draw(3)
    assert "viewport and scissor" in e.value.message            # dynamic state (pipeline.rs:697)
    cmd.set_viewport(0, 0, 32, 32)
    cmd.set_scissor(0, 0, 32, 32)
    with pytest.raises(mirhi.RhiError) as e:
        cmd.draw(6)
    assert "beyond the bound vertex buffer" in e.value.message
    cmd.bind_index_buffer(ib, 0, mirhi.IndexType.UINT32)
    with pytest.raises(mirhi.RhiError) as e:
        cmd.draw_indexed(6)
    assert "beyond the bound index buffer" in e.value.message
    cmd.draw(3)
    cmd.draw_indexed(3)
    cmd.end_rendering()
    cmd.end()
    device.submit([cmd])
    device.wait_idle()
    for o in (cmd, vb, ib, pipe, img):
        o.destroy()


def test_device_refuses_destroy_with_live_children(mirhi):
    dev = mirhi.Device(0)
    b = mirhi.Buffer(dev, mirhi.BufferUsage.Uniform, 16)
    with pytest.raises(mirhi.RhiError) as e:
        dev.destroy()
    assert "live child" in e.value.message
    b.destroy()
    dev.destroy()


@pytest.mark.parametrize("layout", ["bands", "interleaved"])
def test_tile_row_split_rows_assemble(mirhi, oracle, scenes, layout):
    """each rank renders the tile rows it owns -- one contiguous band, or rows rank, rank + world, ... (interleaved) -- and the
    ranks' rows tile the frame exactly; nothing outside a rank's rows is touched."""
    from renderer_rs_amd import multigpu
    scene = scenes.random_triangles(3000, 640, 360, seed=12, rmin=3, rmax=60)
    ref = oracle.render(scene, want_bgra8=False)
    world = 4
    out_prim = np.full((scene.height, scene.width), 0xFFFFFFFF, dtype=np.uint32)
    out_col = np.zeros((scene.height, scene.width, 4), dtype=np.float32)
    covered = np.zeros(scene.height, dtype=np.int32)
    for rank in range(world):
        dev = mirhi.Device(0)
        dev.set_tile_split(rank, world, layout=layout)
        assert dev.split_rows(scene.height) == multigpu.split_tile_rows(scene.height, rank, world, layout)
        if layout == "bands":
            assert dev.band_rows(scene.height) == multigpu.band_rows(scene.height, rank, world)
        else:
            with pytest.raises(mirhi.RhiError):
                dev.band_rows(scene.height)                       # (no single band: split_rows is the query)
        res = mirhi.SceneResources(dev, scene, want_prim=True)
        res.color.upload(np.full((scene.height, scene.width, 4), -7.0, dtype=np.float32))   # sentinel
        res.render()
        got = res.read()
        mine = np.zeros(scene.height, dtype=bool)
        for r0, r1 in multigpu.owned_pixel_rows(scene.height, rank, world, layout):
            mine[r0:r1] = True
        assert (got["color"][~mine] == -7.0).all()                # the other ranks' rows untouched
        out_prim[mine] = got["prim"][mine]
        out_col[mine] = got["color"][mine]
        covered += mine
        res.destroy()
        dev.destroy()
    assert (covered == 1).all()
    assert np.array_equal(out_prim, ref["prim"])
    assert np.abs(out_col[..., :3] - ref["rgba"][..., :3]).max() < 1e-4


def test_two_scopes_load_color_and_depth(mirhi, oracle, device, scenes):
    """LOAD / STORE ops (rendering.rs:200-209,455-470): drawing B in a second scope over A's colour + depth equals
    drawing A and B in one scope."""
    W, H = 256, 160
    a = scenes.random_triangles(120, W, H, seed=31, rmin=5, rmax=50).draws[0]
    b = scenes.random_triangles(120, W, H, seed=32, rmin=5, rmax=50).draws[0]
    both = scenes.Scene("ab", W, H, [a, b], clear_color=(0.2, 0.1, 0.3, 1.0))
    ref = oracle.render(both, want_bgra8=False)
    color = mirhi.Image(device, W, H, mirhi.Format.R32G32B32A32_SFLOAT)
    depth = mirhi.Image(device, W, H, mirhi.Format.D32_SFLOAT)
    pipe = (_tri_builder(mirhi).depth_attachment_format(mirhi.Format.D32_SFLOAT).cull_mode(mirhi.CullMode.NONE).build(device))
    vba = mirhi.Buffer.new_with_data(device, mirhi.BufferUsage.Vertex, a.vertices)
    vbb = mirhi.Buffer.new_with_data(device, mirhi.BufferUsage.Vertex, b.vertices)
    cmd = mirhi.CommandBuffer(device)
    cmd.begin()
    for i, (vb, d) in enumerate(((vba, a), (vbb, b))):
        cmd.begin_rendering(color, clear_color=both.clear_color, color_load_op=mirhi.LoadOp.CLEAR if i == 0 else mirhi.LoadOp.LOAD,
                            depth=depth, depth_load_op=mirhi.LoadOp.CLEAR if i == 0 else mirhi.LoadOp.LOAD,
                            depth_store_op=mirhi.StoreOp.STORE)
        cmd.set_viewport(0, 0, W, H)
        cmd.set_scissor(0, 0, W, H)
        cmd.bind_pipeline(pipe)
        cmd.bind_vertex_buffers(0, [vb], [0])
        cmd.draw(d.count)
        cmd.end_rendering()
    cmd.end()
    device.submit([cmd])
    device.wait_idle()
    got_c, got_d = color.read(), depth.read()
    assert np.abs(got_c[..., :3] - ref["rgba"][..., :3]).max() < 1e-4
    assert np.array_equal(got_d.view(np.uint32), ref["depth"].view(np.uint32))
    for o in (cmd, vba, vbb, pipe, color, depth):
        o.destroy()


def test_uniform_update_between_submits(mirhi, oracle, device, scenes):
    """Uniform buffers are host-coherent in the reference (buffer.rs:86-100): write_data between submits is seen."""
    scene = scenes.displaced_sphere(16, 11, 160, 120, seed=4, program=scenes.PROGRAM_MODEL)
    res = mirhi.SceneResources(device, scene, want_prim=True)
    res.render()
    first = res.read()
    model = scenes.trs((1.2, 0.8, 1.0), scenes.quat_axis_angle((0, 0, 1), 0.9), (0.3, -0.2, 0.0))
    scene.draws[0].object = scenes.object_ubo(model)
    res.draw_state[0]["object"].write_data(0, scene.draws[0].object)
    res.render()
    second = res.read()
    ref = oracle.render(scene, want_bgra8=False)
    assert not np.array_equal(first["prim"], second["prim"])
    assert np.array_equal(second["prim"], ref["prim"])
    res.destroy()


def test_big_list_and_bin_spill_paths(mirhi, oracle, device, scenes, monkeypatch):
    """Many triangles in one tile overflow its bin into the big list; the result is unchanged (idempotent resolve)."""
    monkeypatch.setenv("MIRHI_BIN_CAP", "1024")          # records per list (default 4096: 64 pages)
    n = 3000
    rng = np.random.default_rng(5)
    c = np.tile(np.array([[-0.9, -0.9]]), (n, 1)) + rng.uniform(0, 0.02, (n, 2))
    pts = np.zeros((n, 3, 6), dtype=np.float32)
    for k, (dx, dy) in enumerate(((0, 0), (0.03, 0.0), (0.0, 0.04))):
        pts[:, k, 0] = c[:, 0] + dx
        pts[:, k, 1] = c[:, 1] + dy
        pts[:, k, 2] = rng.uniform(0.1, 0.9, n)[:]
    pts[:, :, 3:] = rng.uniform(0, 1, (n, 1, 3))
    d = scenes.DrawSpec(vertices=pts.reshape(-1, 6), stride=24, count=3 * n, cull_mode=scenes.CULL_NONE)
    scene = scenes.Scene("spill", 640, 480, [d])
    res = mirhi.SceneResources(device, scene, want_prim=True)
    f = mirhi.Fence(device)
    res.render(f)
    f.wait()
    assert device.stats().last_big_list > 0          # the bin (capacity << 3000) spilled
    assert device.stats().last_bin_pages > 0         # and it grew beyond its fixed first page on the way
    got = res.read()
    ref = oracle.render(scene, want_bgra8=False)
    assert np.array_equal(got["prim"], ref["prim"])
    res.destroy()
    f.destroy()


@pytest.mark.parametrize("pool_pages", [0, 3, 40])
def test_bin_pool_exhaustion_spills_and_grows(mirhi, oracle, device, scenes, monkeypatch, pool_pages):
    """The bin pool runs out of dynamic pages (forced: MIRHI_POOL_PAGES): the pages that could not be had send their records to
    the big list -- the frame is still the oracle's -- and the status word says so.  Without the override the host doubles the
    pool in front of the next submit and the big list is empty again."""
    scene = scenes.random_triangles(6000, 320, 200, seed=8, rmin=3, rmax=14)           # ~100 records per tile: 2+ pages each
    ref = oracle.render(scene, want_bgra8=False)
    monkeypatch.setenv("MIRHI_POOL_PAGES", str(pool_pages))
    monkeypatch.setenv("MIRHI_FIXED_PAGES", "1")         # (the density of this scene would earn every tile two fixed pages)
    res = mirhi.SceneResources(device, scene, want_prim=True, want_depth=True)
    f = mirhi.Fence(device)
    for _ in range(3):
        res.render(f); f.wait()
        got = res.read()
        assert np.array_equal(got["prim"], ref["prim"]) and np.array_equal(got["depth"].view(np.uint32)[ref["prim"] != 0xFFFFFFFF], ref["depth"].view(np.uint32)[ref["prim"] != 0xFFFFFFFF])
        st = device.stats()
        assert st.last_status & 4 and st.last_big_list > 0 and st.last_bin_pages >= pool_pages
    res.destroy()
    monkeypatch.delenv("MIRHI_POOL_PAGES")
    res = mirhi.SceneResources(device, scene, want_prim=True)            # (still one fixed page per tile)
    res.render(f); f.wait()
    st = device.stats()
    assert st.last_status == 0 and st.last_big_list == 0 and 0 < st.last_bin_pages < 1000
    assert np.array_equal(res.read()["prim"], ref["prim"])
    res.destroy()
    f.destroy()


def test_pool_grows_after_exhaustion(mirhi, oracle, device, scenes):
    """A scope with more (triangle, tile) pairs than the initial estimate allows: 8000 triangles that each span up to 4 x 4 tiles
    of a 512 x 512 target make ~16 pairs each against a budget of 8 per triangle.  First frame: correct through the big list,
    status bit 2; the submit after the fence has reported it runs on a doubled pool; after a few frames nothing spills any more."""
    rng = np.random.default_rng(3)
    n = 8000
    c = rng.uniform(-0.8, 0.8, (n, 1, 2))
    ang = rng.uniform(0, 2 * np.pi, (n, 1)) + np.array([[0.0, 2.1, 4.2]])
    pts = np.zeros((n, 3, 6), dtype=np.float32)
    pts[:, :, 0] = c[:, :, 0] + 0.22 * np.cos(ang)           # ~56 px radius at 512 px: spans about 4 tiles each way
    pts[:, :, 1] = c[:, :, 1] + 0.22 * np.sin(ang)
    pts[:, :, 2] = rng.uniform(0.1, 0.9, (n, 1))
    pts[:, :, 3:] = rng.uniform(0, 1, (n, 1, 3))
    d = scenes.DrawSpec(vertices=pts.reshape(-1, 6), stride=24, count=3 * n, cull_mode=scenes.CULL_NONE)
    scene = scenes.Scene("many-pairs", 512, 512, [d])
    ref = oracle.render(scene, want_bgra8=False)
    res = mirhi.SceneResources(device, scene, want_prim=True)
    f = mirhi.Fence(device)
    seen_exhausted, settled = False, False
    for _ in range(8):
        res.render(f); f.wait()
        assert np.array_equal(res.read()["prim"], ref["prim"])
        st = device.stats()
        if st.last_status & 4:
            seen_exhausted = True
        elif seen_exhausted:
            settled = True
            break
    assert seen_exhausted and settled
    res.destroy()
    f.destroy()


def test_pbr_textured_alpha_cutoff_is_reported(mirhi, oracle, device, scenes):
    """pixel/model_pbr.hlsl:174-178: a base-colour texture whose texel alpha could straddle alphaCutoff needs a
    per-fragment discard, which only a pipeline with fragment_discard_enable gets (tests of the `alpha_mask` case): without the
    flag the fence reports a PipelineError that names it and the draw is skipped (the other draws of the frame are still exact)."""
    sc = scenes.SMALL_CASES["pbr"]()
    sc.draws[4].material = scenes.pbr_material_ubo((0.3, 0.4, 0.9, 0.6), alpha_cutoff=0.25, has_base_color=True)
    res = mirhi.SceneResources(device, sc, want_prim=True)
    f = mirhi.Fence(device)
    res.render(f)
    with pytest.raises(mirhi.RhiError) as e:
        f.wait()
    assert e.value.code == 9   # MIRHI_ERR_PIPELINE
    assert "alpha cutoff" in e.value.message and "fragment_discard_enable" in e.value.message
    out = res.read()
    del sc.draws[4]
    ref = oracle.render(sc, want_bgra8=False)
    assert np.array_equal(out["prim"], ref["prim"])
    f.destroy()
    # without a fence the same report comes from wait_idle (once)
    res.render()
    with pytest.raises(mirhi.RhiError) as e:
        device.wait_idle()
    assert e.value.code == 9 and "alpha cutoff" in e.value.message
    device.wait_idle()
    res.destroy()
    # the next clean frame clears the condition
    res = mirhi.SceneResources(device, scenes.hello_triangle(64, 64))
    f = mirhi.Fence(device)
    res.render(f)
    f.wait()
    f.destroy()
    res.destroy()


@pytest.mark.parametrize("op,write", [("Less", False), ("LessOrEqual", False), ("Greater", False), ("GreaterOrEqual", False),
                                      ("Equal", False), ("NotEqual", False), ("Equal", True), ("Always", True), ("Never", True)])
def test_predicate_depth_states_second_scope(mirhi, oracle, device, scenes, op, write):
    """Depth test without depth write, Equal (the depth-pre-pass pattern), Always with write and Never
    (pipeline.rs:375-409,677-679): scope 1 lays down depth with Less + write, scope 2 loads it and draws with the state
    under test.  The oracle applies the same fragments one by one in submission order."""
    W, H = 224, 144
    a = scenes.random_triangles(150, W, H, seed=41, rmin=6, rmax=60).draws[0]
    extra = scenes.random_triangles(150, W, H, seed=42, rmin=6, rmax=60).draws[0]
    # scope 2 draws A's triangles again (so Equal has something to hit) followed by unrelated ones, in new colours
    vb2 = np.concatenate([a.vertices.reshape(-1, 6), extra.vertices.reshape(-1, 6)], axis=0).copy()
    vb2[:, 3:6] = 1.0 - vb2[:, 3:6]
    cmp_ = getattr(scenes, {"Less": "CMP_LESS", "LessOrEqual": "CMP_LESS_OR_EQUAL", "Greater": "CMP_GREATER",
                            "GreaterOrEqual": "CMP_GREATER_OR_EQUAL", "Equal": "CMP_EQUAL", "NotEqual": "CMP_NOT_EQUAL",
                            "Always": "CMP_ALWAYS", "Never": "CMP_NEVER"}[op])
    b = scenes.DrawSpec(vertices=vb2, stride=24, count=vb2.shape[0], cull_mode=scenes.CULL_NONE, depth_test=True,
                        depth_write=write, depth_compare=cmp_)
    both = scenes.Scene("pred", W, H, [a, b], clear_color=(0.2, 0.1, 0.3, 1.0))
    ref = oracle.render(both, want_bgra8=False)
    color = mirhi.Image(device, W, H, mirhi.Format.R32G32B32A32_SFLOAT)
    depth = mirhi.Image(device, W, H, mirhi.Format.D32_SFLOAT)
    base = lambda: _tri_builder(mirhi).depth_attachment_format(mirhi.Format.D32_SFLOAT).cull_mode(mirhi.CullMode.NONE)
    pipe_a = base().build(device)
    pipe_b = base().depth_write_enable(write).depth_compare_op(getattr(mirhi.CompareOp, op)).build(device)
    vba = mirhi.Buffer.new_with_data(device, mirhi.BufferUsage.Vertex, a.vertices)
    vbb = mirhi.Buffer.new_with_data(device, mirhi.BufferUsage.Vertex, vb2)
    cmd = mirhi.CommandBuffer(device)
    cmd.begin()
    for i, (vb, d, pipe) in enumerate(((vba, a, pipe_a), (vbb, b, pipe_b))):
        cmd.begin_rendering(color, clear_color=both.clear_color, color_load_op=mirhi.LoadOp.CLEAR if i == 0 else mirhi.LoadOp.LOAD,
                            depth=depth, depth_load_op=mirhi.LoadOp.CLEAR if i == 0 else mirhi.LoadOp.LOAD,
                            depth_store_op=mirhi.StoreOp.STORE)
        cmd.set_viewport(0, 0, W, H)
        cmd.set_scissor(0, 0, W, H)
        cmd.bind_pipeline(pipe)
        cmd.bind_vertex_buffers(0, [vb], [0])
        cmd.draw(d.count)
        cmd.end_rendering()
    cmd.end()
    device.submit([cmd])
    device.wait_idle()
    got_c, got_d = color.read(), depth.read()
    assert np.array_equal(got_d.view(np.uint32), ref["depth"].view(np.uint32)), f"{op} write={write}: depth differs"
    assert np.abs(got_c[..., :3] - ref["rgba"][..., :3]).max() < 1e-4, f"{op} write={write}: colour differs"
    for o in (cmd, vba, vbb, pipe_a, pipe_b, color, depth):
        o.destroy()


def test_predicate_depth_state_against_cleared_depth(mirhi, oracle, device, scenes):
    """Depth test without write in a single scope: every fragment is tested against the clear value, the last passing one
    stays (SceneResources path, 8-bit target + primitive ids)."""
    sc = scenes.random_triangles(400, 320, 200, seed=43, rmin=4, rmax=50)
    d = sc.draws[0]
    d.depth_write = False
    d.depth_compare = scenes.CMP_GREATER
    sc.clear_depth = 0.5
    res = mirhi.SceneResources(device, sc, mirhi.Format.B8G8R8A8_SRGB, want_prim=True)
    res.render()
    out = res.read()
    ref = oracle.render(sc, want_bgra8=True)
    assert np.array_equal(out["prim"], ref["prim"])
    assert np.abs(out["color"].astype(np.int32) - ref["bgra8"].astype(np.int32)).max() <= 1
    res.destroy()


@pytest.mark.parametrize("want_depth", [False, True])
def test_mixed_depth_states_in_one_scope(mirhi, oracle, device, scenes, want_depth):
    """Pipelines with different depth state inside ONE rendering scope (opaque Less + write, then a far 'sky' quad with
    LessOrEqual and no write, then an overlay without depth test, then Less + write again): the scope is cut into segments
    that hand colour and depth over, with or without a depth attachment; the result equals the oracle's fragment-by-fragment
    order."""
    W, H = 256, 176
    opaque = scenes.random_triangles(200, W, H, seed=51, rmin=6, rmax=70).draws[0]
    sky = np.array([[-1, -1, 1.0, 0.1, 0.2, 0.6], [1, -1, 1.0, 0.1, 0.2, 0.6], [1, 1, 1.0, 0.3, 0.5, 0.9],
                    [-1, -1, 1.0, 0.1, 0.2, 0.6], [1, 1, 1.0, 0.3, 0.5, 0.9], [-1, 1, 1.0, 0.3, 0.5, 0.9]], dtype=np.float32)
    sky_d = scenes.DrawSpec(vertices=sky, stride=24, count=6, cull_mode=scenes.CULL_NONE, depth_test=True, depth_write=False,
                            depth_compare=scenes.CMP_LESS_OR_EQUAL)
    overlay = scenes.random_triangles(12, W, H, seed=52, rmin=8, rmax=30).draws[0]
    overlay.depth_test = False
    overlay.depth_write = False
    more = scenes.random_triangles(150, W, H, seed=53, rmin=6, rmax=70).draws[0]
    sc = scenes.Scene("mixed", W, H, [opaque, sky_d, overlay, more], clear_color=(0.0, 0.0, 0.0, 1.0))
    ref = oracle.render(sc, want_bgra8=False)
    res = mirhi.SceneResources(device, sc, want_prim=True, want_depth=want_depth)
    for _ in range(2):                      # the second submission reuses the re-armed workspace
        res.render()
    out = res.read()
    assert np.array_equal(out["prim"], ref["prim"])
    assert np.abs(out["color"][..., :3] - ref["rgba"][..., :3]).max() < 1e-4
    if want_depth:
        assert np.array_equal(out["depth"].view(np.uint32), ref["depth"].view(np.uint32))
    res.destroy()


def test_frames_in_flight_on_four_lanes_do_not_interfere(mirhi, oracle, scenes):
    """Four different scenes, each with its own command buffer / workspace / targets on its own queue lane, submitted round
    robin without waiting in between (the bench's submission pattern): every one of them must still equal its oracle frame."""
    dev = mirhi.Device(0)
    dev.set_queue_lanes(4)
    cases = [scenes.random_triangles(3000, 640, 360, seed=s, rmin=3, rmax=60) for s in (61, 62)] + \
            [scenes.displaced_sphere(40, 30, 640, 360, seed=63), scenes.SMALL_CASES["pbr"]()]
    res = [mirhi.SceneResources(dev, sc, want_prim=True, want_depth=True) for sc in cases]
    for _ in range(30):
        for r in res:
            r.render()
    dev.wait_idle()
    for sc, r in zip(cases, res):
        out = r.read()
        ref = oracle.render(sc, want_bgra8=False)
        assert np.array_equal(out["prim"], ref["prim"]), sc.name
        cov = ref["prim"] != 0xFFFFFFFF
        assert np.array_equal(out["depth"].view(np.uint32)[cov], ref["depth"].view(np.uint32)[cov]), sc.name
        assert np.abs(out["color"] - ref["rgba"]).max() < 1e-4, sc.name
    for r in res:
        r.destroy()
    dev.destroy()


def _blend_scene(scenes, W, H, blend, seed, program_model=False, n=120, depth_test=True):
    """Opaque geometry first, then blended triangles over it (depth-tested against the opaque depth, not written)."""
    opaque = scenes.random_triangles(n, W, H, seed=seed, rmin=8, rmax=70).draws[0]
    over = scenes.random_triangles(n, W, H, seed=seed + 1, rmin=8, rmax=70).draws[0]
    over.blend = blend
    over.depth_test, over.depth_write = depth_test, False
    draws = [opaque, over]
    if program_model:      # a lit, textured mesh with material alpha 0.5 blended on top (alpha comes from the fragment program)
        sph = scenes.displaced_sphere(24, 16, W, H, seed=seed + 2).draws[0]
        sph.material = scenes.material_ubo((0.9, 0.6, 0.3, 0.5), 0.0, 0.4, 1.0)
        sph.blend = scenes.ALPHA_BLEND
        sph.depth_write = False
        draws.append(sph)
    return scenes.Scene("blend", W, H, draws, clear_color=(0.2, 0.3, 0.1, 0.5))


@pytest.mark.parametrize("name", ["alpha", "additive", "multiply", "min_max", "subtract_masked", "dst_alpha_saturate"])
def test_blending_in_primitive_order(mirhi, oracle, device, scenes, name):
    """ColorBlendAttachment (pipeline.rs:478-531): blended draws are resolved fragment by fragment in primitive order against
    the colour written so far; float target, so the comparison with the oracle is direct."""
    S = scenes
    blend = {"alpha": S.ALPHA_BLEND,
             "additive": (S.BF_ONE, S.BF_ONE, S.BO_ADD, S.BF_ONE, S.BF_ONE, S.BO_ADD, 0xF),
             "multiply": (S.BF_DST_COLOR, S.BF_ZERO, S.BO_ADD, S.BF_DST_ALPHA, S.BF_ZERO, S.BO_ADD, 0xF),
             "min_max": (S.BF_ONE, S.BF_ONE, S.BO_MIN, S.BF_ONE, S.BF_ONE, S.BO_MAX, 0xF),
             "subtract_masked": (S.BF_SRC_COLOR, S.BF_ONE_MINUS_SRC_COLOR, S.BO_REVERSE_SUBTRACT, S.BF_ONE, S.BF_ONE_MINUS_SRC_ALPHA, S.BO_SUBTRACT, 0b0101),
             "dst_alpha_saturate": (S.BF_SRC_ALPHA_SATURATE, S.BF_ONE_MINUS_DST_ALPHA, S.BO_ADD, S.BF_DST_ALPHA, S.BF_ONE_MINUS_DST_COLOR, S.BO_ADD, 0xF)}[name]
    sc = _blend_scene(S, 288, 192, blend, seed=71, program_model=(name == "alpha"))
    ref = oracle.render(sc, want_bgra8=False)
    res = mirhi.SceneResources(device, sc, want_prim=True, want_depth=True)
    res.render(); res.render()
    out = res.read()
    res.destroy()
    assert np.array_equal(out["prim"], ref["prim"])
    assert np.array_equal(out["depth"].view(np.uint32), ref["depth"].view(np.uint32))
    assert np.abs(out["color"] - ref["rgba"]).max() < 1e-4


def test_blending_srgb8_target_and_near_clipped_blended_triangles(mirhi, oracle, device, scenes):
    """8-bit target: the destination is decoded exactly, blended in float through the segment and encoded once at its end
    (+-1 code against the oracle's float pipeline, +1 for the re-quantisation between segments).  The blended mesh is
    placed across the near plane, so some of its triangles reach the kernel as clipped pieces at their place in the order."""
    W, H = 256, 160
    opaque = scenes.random_triangles(150, W, H, seed=81, rmin=8, rmax=60).draws[0]
    s = scenes.displaced_sphere(16, 12, W, H, seed=82).draws[0]
    eye = (0.0, 0.2, 0.9)
    view = scenes.look_at_rh(eye, (0.0, 0.0, 0.0), (0.0, 1.0, 0.0))
    s.camera = scenes.camera_ubo(view, scenes.projection_vulkan(1.2, W / H, 0.3, 20.0), eye)
    s.material = scenes.material_ubo((0.3, 0.7, 0.9, 0.6), 0.0, 0.5, 1.0)
    s.cull_mode = scenes.CULL_NONE
    s.blend = scenes.ALPHA_BLEND
    s.depth_write = False
    sc = scenes.Scene("blend8", W, H, [opaque, s], clear_color=(0.1, 0.1, 0.2, 1.0))
    ref = oracle.render(sc, want_bgra8=True)
    for fmt in (mirhi.Format.R32G32B32A32_SFLOAT, mirhi.Format.B8G8R8A8_SRGB):
        res = mirhi.SceneResources(device, sc, fmt, want_prim=True)
        res.render()
        out = res.read()
        res.destroy()
        assert np.array_equal(out["prim"], ref["prim"])
        if fmt == mirhi.Format.R32G32B32A32_SFLOAT:
            assert np.abs(out["color"] - ref["rgba"]).max() < 1e-4
        else:
            assert np.abs(out["color"].astype(np.int32) - ref["bgra8"].astype(np.int32)).max() <= 2


def test_not_equal_with_depth_write_is_resolved_in_order(mirhi, oracle, device, scenes):
    sc = scenes.random_triangles(300, 256, 160, seed=91, rmin=6, rmax=60)
    d = sc.draws[0]
    d.depth_compare, d.depth_write = scenes.CMP_NOT_EQUAL, True
    sc.clear_depth = 0.5
    ref = oracle.render(sc, want_bgra8=False)
    res = mirhi.SceneResources(device, sc, want_prim=True, want_depth=True)
    res.render()
    out = res.read()
    res.destroy()
    assert np.array_equal(out["prim"], ref["prim"]) and np.array_equal(out["depth"].view(np.uint32), ref["depth"].view(np.uint32))
    assert np.abs(out["color"] - ref["rgba"]).max() < 1e-4


def test_kernel_times_and_timeline(mirhi, device, scenes):
    """SURVEY 8d measurement API: every dispatch carries its own event pair (hipExtLaunchKernelGGL start / stop events), so a
    kernel's time is its begin -> end on the GPU clock -- nothing recorded into the stream, nothing subtracted -- and the
    timeline puts all dispatches on one axis (mirhi_device_kernel_time / mirhi_device_timeline)."""
    res = mirhi.SceneResources(device, scenes.random_triangles(2000, 640, 360, seed=9), mirhi.Format.B8G8R8A8_SRGB)
    for _ in range(5):
        res.render()
    device.wait_idle()
    device.set_profiling(mirhi.Profile.TIMING)
    device.reset_kernel_times()
    n = 40
    for _ in range(n):
        res.render()
    device.wait_idle()
    g_ms, g_n = device.kernel_time(mirhi.Kernel.GEOMETRY)
    r_ms, r_n = device.kernel_time(mirhi.Kernel.RASTER)
    tl = device.timeline()
    device.set_profiling(False)
    assert g_n == n and r_n == n and device.kernel_time(mirhi.Kernel.VERTEX)[1] == 0 and len(tl) == 2 * n
    assert 0.001 < g_ms / n < 1.0 and 0.001 < r_ms / n < 1.0          # microseconds to a fraction of a millisecond each
    # one lane: dispatches are serial -- each begins after the previous one ended (within the timestamp resolution) --
    # kernels alternate geometry / raster, and the durations on the timeline are the ones kernel_time sums
    assert [k for k, _, _, _ in tl] == [mirhi.Kernel.GEOMETRY, mirhi.Kernel.RASTER] * n
    assert abs(tl[0][2]) < 1e-6
    for (k0, l0, b0, e0), (k1, l1, b1, e1) in zip(tl, tl[1:]):
        assert e0 > b0 and b1 >= e0 - 0.5, (b0, e0, b1, e1)
    assert abs(sum(e - b for k, _, b, e in tl if k == mirhi.Kernel.RASTER) - 1e3 * r_ms) < 1e-3 * n
    device.reset_kernel_times()
    assert device.kernel_time(mirhi.Kernel.RASTER) == (0.0, 0) and device.timeline() == []
    res.render()
    device.wait_idle()
    assert device.kernel_time(mirhi.Kernel.RASTER)[1] == 0      # profiling is off: nothing recorded
    res.destroy()


def test_fragment_statistics(mirhi, oracle, device, scenes):
    """shaded pixels = pixels that ran the fragment stage (winners of the depth resolve), covered fragments = what
    the rasterizer emits before the depth test.  Winners are compared with the oracle's primitive-id image; fragments with the
    sum of the oracle's single-triangle coverages."""
    import copy
    scene = scenes.random_triangles(48, 200, 120, seed=4, rmin=6, rmax=40)
    ref = oracle.render(scene, want_bgra8=False)
    winners = int((ref["prim"] != 0xFFFFFFFF).sum())
    frags = 0
    v = scene.draws[0].vertices
    for t in range(scene.num_triangles):
        one = copy.copy(scene)
        d = copy.copy(scene.draws[0])
        d.vertices, d.count = v[3 * t:3 * t + 3].copy(), 3
        one.draws = [d]
        frags += int((oracle.render(one, want_bgra8=False)["prim"] != 0xFFFFFFFF).sum())
    assert frags > winners > 0
    for fmt, want_prim in ((mirhi.Format.B8G8R8A8_SRGB, False), (mirhi.Format.R32G32B32A32_SFLOAT, True)):
        res = mirhi.SceneResources(device, scene, fmt, want_prim=want_prim)
        res.render()                                 # not counted: statistics are off
        device.wait_idle()
        device.reset_kernel_times()
        device.set_profiling(mirhi.Profile.FRAGMENTS)
        for _ in range(3):
            res.render()
        device.set_profiling(0)
        shaded, covered, scopes = device.fragment_stats()
        assert (shaded, covered, scopes) == (3 * winners, 3 * frags, 3)
        res.render()
        assert device.fragment_stats() == (3 * winners, 3 * frags, 3)         # off again: nothing added
        out = res.read()                             # the statistics pass left the frame as it was
        if want_prim:
            assert np.array_equal(out["prim"], ref["prim"])
        res.destroy()
    # hello triangle: K2's 8192 pixels, every fragment visible; a LOAD scope counts only what it covers itself
    hello = scenes.hello_triangle(256, 256)
    first = mirhi.SceneResources(device, hello, mirhi.Format.B8G8R8A8_SRGB)
    second = mirhi.SceneResources(device, hello, mirhi.Format.B8G8R8A8_SRGB, color_image=first.color, color_load_op=mirhi.LoadOp.LOAD)
    device.reset_kernel_times()
    device.set_profiling(mirhi.Profile.FRAGMENTS | mirhi.Profile.TIMING)
    first.render(); second.render()
    device.set_profiling(0)
    assert device.fragment_stats() == (2 * 8192, 2 * 8192, 2)
    assert device.kernel_time(mirhi.Kernel.FRAGMENT_COUNT)[1] == 2
    second.color = None
    second.destroy(); first.destroy()


def test_command_buffer_destroyed_before_its_fence_is_waited(mirhi, device, scenes):
    """submit(cmd, fence); wait_idle(); destroy(cmd); fence.wait() -- Vulkan allows it and the Rust wrapper's Drop
    order does it; the fence must not look at the freed command buffer, and re-recording must not lose a pending status."""
    scene = scenes.random_triangles(300, 320, 200, seed=2)
    res = mirhi.SceneResources(device, scene, mirhi.Format.B8G8R8A8_SRGB)
    fence = mirhi.Fence(device)
    res.render(fence)
    device.wait_idle()
    color, res.color = res.color, None
    res.destroy()                                    # destroys the command buffer; the fence still lists it
    fence.wait()
    assert fence.is_signaled()
    fence.reset()
    assert not fence.is_signaled()
    fence.destroy()
    color.destroy()
    # re-record while a fence is outstanding: still fine
    res = mirhi.SceneResources(device, scene, mirhi.Format.B8G8R8A8_SRGB)
    fence = mirhi.Fence(device)
    res.render(fence)
    res.record()
    fence.wait()
    res.render(fence)
    fence.wait()
    fence.destroy()
    res.destroy()


def test_comm_single_rank(mirhi, device, scenes):
    """mirhi_comm_* with world = 1 on the one GPU of this box: librccl is found and bound (dlopen), the communicator counts one
    rank, the gather of a single band is a no-op that leaves the frame intact.  (N > 1 needs N GPUs: the band arithmetic is
    covered by the gloo tests, the exchange itself by the driver's multi-GPU run of bench.py.)"""
    dev = mirhi.Device(0)
    uid = mirhi.Comm.unique_id()
    assert len(uid) == mirhi.COMM_ID_BYTES and any(uid)
    comm = mirhi.Comm(dev, uid, 0, 1)
    assert comm.world() == 1 and comm.rank() == 0
    scene = scenes.random_triangles(500, 400, 250, seed=12)
    res = mirhi.SceneResources(dev, scene, mirhi.Format.B8G8R8A8_SRGB)
    res.render()
    before = res.read()["color"]
    for algo in (mirhi.GatherAlgo.DIRECT, mirhi.GatherAlgo.BROADCAST):
        comm.all_gather_bands(res.color, res.cmd, algo)
    assert np.array_equal(res.read()["color"], before)
    with pytest.raises(mirhi.RhiError):
        dev.destroy()                                # the communicator is a live child
    comm.destroy()
    res.destroy()
    dev.destroy()


def test_indirect_draws_and_push_constants(mirhi, oracle, device, scenes):
    """draw_indirect / draw_indexed_indirect (command.rs:630-661) with the arguments in a GPU-only Indirect buffer (buffer.rs:86-100),
    filled through a staging upload: the frame equals the one the same draws give when recorded directly.  push_constants
    (command.rs:732-769) is validated and otherwise without effect: no program of the path reads a push-constant block."""
    import struct
    scene = scenes.random_triangles(900, 400, 260, seed=15)
    d = scene.draws[0]
    ref = oracle.render(scene, want_bgra8=False)
    color = mirhi.Image(device, scene.width, scene.height, mirhi.Format.R32G32B32A32_SFLOAT)
    prim = mirhi.Image(device, scene.width, scene.height, mirhi.Format.R32_UINT)
    pipe = (mirhi.GraphicsPipelineBuilder().vertex_shader(mirhi.Program.TRIANGLE).fragment_shader(mirhi.Program.TRIANGLE)
            .vertex_binding(24).vertex_attributes(mirhi.TRIANGLE_VERTEX_OFFSETS).color_attachment_format(mirhi.Format.R32G32B32A32_SFLOAT)
            .cull_mode(d.cull_mode).depth_test_enable(True).depth_write_enable(True).depth_attachment_format(mirhi.Format.D32_SFLOAT).build(device))
    vb = mirhi.Buffer.new_with_data(device, mirhi.BufferUsage.Vertex, d.vertices)
    # three non-indexed draws (300 triangles each, the second one behind a gap in the buffer: stride 32), then the same as indexed draws
    args = b"".join(struct.pack("<4I", 900, 1, 900 * k, 0) + b"\xEE" * 16 for k in range(3))
    ind = mirhi.Buffer(device, mirhi.BufferUsage.Indirect, len(args) + 64)
    with pytest.raises(mirhi.RhiError):
        ind.write_data(0, args)                          # GPU-only memory is not mapped (buffer.rs:266-268)
    ind.upload_via_staging(args)
    cmd = mirhi.CommandBuffer(device)

    def record(indirect):
        cmd.begin_reusable()
        cmd.begin_rendering(color, clear_color=scene.clear_color, prim_id=prim)
        cmd.set_viewport(0.0, 0.0, float(scene.width), float(scene.height)); cmd.set_scissor(0, 0, scene.width, scene.height)
        cmd.bind_pipeline(pipe); cmd.bind_vertex_buffers(0, [vb], [0])
        cmd.push_constants(1, 16, struct.pack("<4f", 1, 2, 3, 4))
        if indirect:
            cmd.draw_indirect(ind, 0, 3, 32)
        else:
            for k in range(3):
                cmd.draw(900, 1, 900 * k, 0)
        cmd.end_rendering(); cmd.end()

    out = []
    for indirect in (False, True):
        record(indirect)
        device.submit([cmd]); device.wait_idle()
        out.append((color.read(), prim.read()))
    assert np.array_equal(out[0][1], ref["prim"]) and np.array_equal(out[1][1], ref["prim"]) and np.array_equal(out[0][0], out[1][0])
    # validation: unaligned offset, stride too small, range beyond the buffer, push constants beyond 128 bytes / unaligned
    cmd.begin_reusable(); cmd.begin_rendering(color, prim_id=prim)
    cmd.set_viewport(0.0, 0.0, float(scene.width), float(scene.height)); cmd.set_scissor(0, 0, scene.width, scene.height)
    cmd.bind_pipeline(pipe); cmd.bind_vertex_buffers(0, [vb], [0])
    for bad in ((2, 1, 16), (0, 2, 12), (len(args) + 60, 1, 16), (0, 9, 32)):
        with pytest.raises(mirhi.RhiError):
            cmd.draw_indirect(ind, *bad)
    with pytest.raises(mirhi.RhiError):
        cmd.draw_indexed_indirect(ind, 0, 1, 20)          # no index buffer bound
    for off, n in ((126, 4), (0, 132), (4, 6)):
        with pytest.raises(mirhi.RhiError):
            cmd.push_constants(1, off, bytes(n))
    cmd.draw_indirect(ind, 0, 0, 0)                        # draw_count 0: nothing
    cmd.end_rendering(); cmd.end()
    for o in (cmd, ind, vb, pipe, prim, color):
        o.destroy()


def test_max_anisotropy_is_sampler_state_of_a_texture(mirhi, device):
    """mirhi_image_set_max_anisotropy: default 1, [1, 16] accepted (maxSamplerAnisotropy), refused on anything but a sampled
    R8G8B8A8 texture (device.rs:161-165 enables the feature; sampler.rs is a stub)."""
    t = mirhi.Image(device, 8, 8, mirhi.Format.R8G8B8A8_SRGB)
    assert t.max_anisotropy == 1
    t.set_max_anisotropy(16)
    assert t.max_anisotropy == 16
    for bad in (0, 17, 1 << 20):
        with pytest.raises(mirhi.RhiError) as e:
            t.set_max_anisotropy(bad)
        assert "maxAnisotropy" in str(e.value)
    assert t.max_anisotropy == 16
    target = mirhi.Image(device, 8, 8, mirhi.Format.B8G8R8A8_SRGB)
    with pytest.raises(mirhi.RhiError) as e:
        target.set_max_anisotropy(4)
    assert "Invalid handle" in str(e.value)
    target.destroy()
    t.destroy()


def test_instanced_draws_repeat_their_primitives_in_instance_order(mirhi, oracle, device, scenes):
    """draw(vertex_count, instance_count) (command.rs:583-599): no program reads an instance index and binding 0 is per-vertex, so
    instance i is the same primitives again.  Under additive blending every instance adds once more; under LESS the first instance
    keeps the pixel (its primitive ids win); the oracle sees the draw once per instance."""
    sc = scenes.random_triangles(40, 200, 120, seed=77, rmin=6, rmax=40)
    d = sc.draws[0]
    d.instances = 3
    out_less = _both(mirhi, oracle, device, sc)
    assert out_less[0]["prim"][out_less[0]["prim"] != 0xFFFFFFFF].max() < 40           # instance 0 holds every covered pixel
    d.depth_test = False; d.depth_write = False
    d.blend = (scenes.BF_ONE, scenes.BF_ONE, scenes.BO_ADD, scenes.BF_ONE, scenes.BF_ONE, scenes.BO_ADD, 0xF)
    out_add = _both(mirhi, oracle, device, sc)
    covered = out_add[0]["prim"] != 0xFFFFFFFF
    assert out_add[0]["prim"][covered].min() >= 80                                      # the last instance wrote last
    assert sc.num_triangles == 120
    res = mirhi.SceneResources(device, sc)                                              # the stated limit is an error, not a hang
    d.instances = 5000
    with pytest.raises(mirhi.RhiError) as e:
        res.record()
    assert "instance_count" in str(e.value)
    res.destroy()


def _both(mirhi, oracle, device, scene):
    res = mirhi.SceneResources(device, scene, want_prim=True)
    res.render()
    out = res.read()
    res.destroy()
    ref = oracle.render(scene, want_bgra8=False)
    assert np.array_equal(out["prim"], ref["prim"])
    err = np.abs(out["color"][..., :4] - ref["rgba"]) / np.maximum(1.0, np.abs(ref["rgba"]))
    assert float(err.max()) < 1e-4
    return out, ref


def test_two_team_mode_is_dropped_for_a_spread_out_mesh(mirhi, oracle, device, scenes):
    """raster_mode picks the two-team mesh variant from the triangle count alone; a scope that then opens more than two list pages
    per tile was spread over the frame, and the command buffer's plan is rebuilt with one team in front of its next submit -- same
    pixels, and the statistic that showed it (dynamic bin pages) drops to the single-list figure."""
    scene = scenes.heightfield_grid(40, 40, 640, 360)
    res = mirhi.SceneResources(device, scene, want_prim=True)
    res.render(); device.wait_idle()
    first = device.stats().last_bin_pages
    a = res.read()
    res.render(); device.wait_idle()
    second = device.stats().last_bin_pages
    b = res.read()
    ref = oracle.render(scene, want_bgra8=False)
    assert np.array_equal(a["prim"], ref["prim"]) and np.array_equal(b["prim"], ref["prim"])
    tiles = ((640 + 31) // 32) * ((360 + 31) // 32)
    assert first > 2 * tiles and second < first // 4
    res.destroy()


def test_device_is_shared_by_host_threads(mirhi, oracle, device, scenes):
    """`Device` is Send + Sync in the reference (device.rs:379-380) while command buffers are externally synchronised
    (command.rs:48-51): four host threads, each with its own resources, command buffer, fence and queue lane, record, submit,
    wait and read back concurrently on ONE device (ctypes drops the GIL inside every call).  Every frame of every thread must be the
    oracle's, and creating / destroying objects under the other threads' submissions must not disturb them."""
    import threading
    seeds = [11, 12, 13, 14]
    made = [scenes.random_triangles(400 + 50 * k, 320, 200, seed=s, rmin=3, rmax=30) for k, s in enumerate(seeds)]
    refs = [oracle.render(sc, want_bgra8=False)["prim"] for sc in made]
    errors = []

    def worker(k):
        try:
            for it in range(12):
                res = mirhi.SceneResources(device, made[k], want_prim=True)       # objects come and go while others render
                res.cmd.set_queue_lane(k)
                f = mirhi.Fence(device)
                for _ in range(3):
                    res.render(f)
                    f.wait()
                    f.reset()
                    if not np.array_equal(res.read()["prim"], refs[k]):
                        errors.append((k, it, "pixels"))
                f.destroy()
                res.destroy()
        except Exception as e:        # noqa: BLE001
            errors.append((k, repr(e)))

    device.set_queue_lanes(4)
    try:
        threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        device.wait_idle()
    finally:
        device.set_queue_lanes(1)
    assert not errors, errors[:5]


def test_band_exchange_with_n_ranks_on_one_gpu(tmp_path):
    """mirhi_comm_all_gather_bands needs N GPUs under the real RCCL.  Here N devices of one child process stand for the ranks and
    tests/mock/mock_rccl.cpp (an in-process stand-in for the ten entry points the library resolves, loaded through
    MIRHI_RCCL_LIBRARY) moves the bytes: band arithmetic, peers, byte counts, both algorithms, uneven and empty bands -- every
    rank's frame must be the unsplit frame byte for byte.  (A child process: the library keeps the RCCL it loaded first.)"""
    import os
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    lib = str(tmp_path / "libmock_rccl.so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", lib,
                           os.path.join(root, "tests", "mock", "mock_rccl.cpp")])
    env = dict(os.environ)
    env.pop("MIRHI_RCCL_LIBRARY", None)
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "mock", "split_exchange_check.py"), lib], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=180)
    assert p.returncode == 0, p.stdout[-3000:]
    assert p.stdout.count(": ok") == 15, p.stdout[-3000:]      # 7 exchanges x 2 layouts (bands, interleaved rows) + the failing-send case (the group is closed again, the next exchange works)
