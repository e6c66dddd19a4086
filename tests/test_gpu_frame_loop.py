"""GPU tests of the draw-submit loop in the reference's shape (crates/renderer/src/renderer.rs:367-449,452-557,
frame_manager.rs:299-539): every frame waits on its in-flight fence, resets and RE-RECORDS its command buffer, ends it and
submits it with the fence.  Round 3 made that path cheap (plan cache, host-written parameter block, no clears, fence on the
last dispatch); these tests pin that it is still the oracle's frame when the recording changes from frame to frame.
"""
import copy
import os

import numpy as np
import pytest

from test_gpu_parity import _check

pytestmark = pytest.mark.gpu
# A/B runs of the whole suite on the HIP launch path (MIRHI_NATIVE_DISPATCH=0): the tests that assert the native path itself have nothing to check then
native_off = pytest.mark.skipif(os.environ.get("MIRHI_NATIVE_DISPATCH") == "0", reason="native dispatch switched off for this run")


def _read(res):
    out = {"color": res.color.read()}
    if res.prim:
        out["prim"] = res.prim.read()
    if res.depth:
        out["depth"] = res.depth.read()
    return out


def test_rerecorded_frames_with_changing_uniform_and_triangle_count(mirhi, oracle, scenes):
    """Two frames in flight on two queue lanes, eight frames; every frame re-records its command buffer with another object
    transform (its slot's uniform buffer rewritten first) AND another triangle count.  Every frame is checked against the
    oracle's render of exactly that frame: winning primitive ids bit for bit, colour within 1e-4."""
    dev = mirhi.Device(0)
    dev.set_queue_lanes(2)
    base = scenes.displaced_sphere(40, 31, 384, 240, seed=11)
    slots = []
    for k in range(2):
        sc = copy.deepcopy(base)
        res = mirhi.SceneResources(dev, sc, mirhi.Format.R32G32B32A32_SFLOAT, want_prim=True)
        res.cmd.set_queue_lane(k)
        slots.append((sc, res, mirhi.Fence(dev, signaled=True)))
    full = base.draws[0].count
    expected = {}

    def verify(f):
        sc, res, fence = slots[f % 2]
        fence.wait()
        _check(_read(res), expected.pop(f), f"frame {f}")

    for f in range(8):
        sc, res, fence = slots[f % 2]
        if f >= 2:
            verify(f - 2)                                   # wait_for_fence(in_flight_fences[current_frame]) -- and look at that frame
        fence.reset()
        d = sc.draws[0]
        d.count = full - 3 * (37 * f)                        # another triangle count every frame
        d.object = scenes.object_ubo(scenes.trs((1.0, 1.0 - 0.03 * f, 1.0), scenes.quat_axis_angle((0.0, 1.0, 0.0), 0.35 * f), (0.02 * f, 0.0, 0.0)))
        res.draw_state[0]["object"].write_data(0, d.object)      # Buffer::write_data: this slot's uniform buffer, its previous frame has been waited for
        res.cmd.reset()
        res.record()                                         # begin .. end: a new recording of another shape
        res.render(fence)
        expected[f] = oracle.render(sc, want_bgra8=False)
    verify(6)
    verify(7)
    for sc, res, fence in slots:
        res.destroy()
        fence.destroy()
    dev.destroy()


@pytest.mark.parametrize("frames_in_flight,lanes,submit_thread", [(2, 2, False), (2, 1, False), (3, 4, False), (2, 2, True), (4, 4, True)])
def test_native_frame_loop_matches_the_oracle(mirhi, oracle, scenes, frames_in_flight, lanes, submit_thread):
    """libmirhost.so (Renderer::render_frame natively, include/mirhost.h): after n frames the image rendered last is the oracle's
    frame -- with the triangle count changing every frame (vary_triangles = 5: frame f draws count - 3 * (f % 5) vertices), so the
    launch plan is rebuilt in every end(), and without (the plan cache path)."""
    from renderer_rs_amd import frameloop
    dev = mirhi.Device(0)
    dev.set_queue_lanes(lanes)
    for make in (lambda: scenes.random_triangles(3000, 640, 360, seed=21), lambda: scenes.displaced_sphere(30, 23, 320, 200, seed=5)):
        scene = make()
        res = mirhi.SceneResources(dev, scene, mirhi.Format.B8G8R8A8_SRGB)
        images = [mirhi.Image(dev, scene.width, scene.height, mirhi.Format.B8G8R8A8_SRGB) for _ in range(frames_in_flight + 1)]
        for vary, n in ((0, 7), (5, 9), (5, 13)):
            loop = frameloop.FrameLoop(dev, res, images, frames_in_flight=frames_in_flight, vary_triangles=vary, submit_thread=submit_thread)
            loop.run(n)
            img, rendered = loop.last_image()
            assert rendered == n
            sc = copy.deepcopy(scene)
            if vary:
                sc.draws[0].count -= 3 * ((n - 1) % vary)
            ref = oracle.render(sc, want_bgra8=True)
            got = img.read()
            d = np.abs(got.astype(np.int32) - ref["bgra8"].astype(np.int32))
            assert d.max() <= 1, f"{scene.name} vary {vary} frame {n}: sRGB8 differs by {d.max()} LSB"
            # every swapchain image holds one of the last frames_in_flight + 1 frames
            loop.destroy()
        for im in images:
            im.destroy()
        res.destroy()
    dev.destroy()


def test_one_command_buffer_through_many_shapes_without_clears(mirhi, oracle, scenes, monkeypatch):
    """The workspace is no longer cleared when a command buffer is re-recorded: the kernels leave counters and page table re-armed and
    the plan relies on it.  One command buffer is recorded with scene after scene of different size, density and program (so tile
    count, bin layout, per-XCD bins and big-list use all change under it); MIRHI_VERIFY_IDLE makes end() check the idle state on the
    host, and every frame must be the oracle's."""
    monkeypatch.setenv("MIRHI_VERIFY_IDLE", "1")
    dev = mirhi.Device(0)
    shared = mirhi.CommandBuffer(dev)
    makers = [lambda: scenes.random_triangles(4000, 640, 360, seed=3), lambda: scenes.SMALL_CASES["huge"](), lambda: scenes.displaced_sphere(40, 31, 512, 300, seed=2),
              lambda: scenes.random_triangles(9000, 320, 200, seed=8, rmin=3, rmax=14), lambda: scenes.SMALL_CASES["near_clip"](), lambda: scenes.SMALL_CASES["multi_draw"](),
              lambda: scenes.random_triangles(500, 1920, 1080, seed=4), lambda: scenes.SMALL_CASES["pbr"](), lambda: scenes.SMALL_CASES["cull_scissor"](),
              lambda: scenes.displaced_sphere(24, 17, 256, 160, seed=3), lambda: scenes.random_triangles(4000, 640, 360, seed=3)]
    for rep in range(2):
        for make in makers:
            scene = make()
            res = mirhi.SceneResources(dev, scene, mirhi.Format.R32G32B32A32_SFLOAT, want_prim=True)
            own = res.cmd
            res.cmd = shared
            res.record()
            for _ in range(2):                              # (twice: the second submit runs on what the first left behind)
                res.render()
                dev.wait_idle()
            _check(_read(res), oracle.render(scene, want_bgra8=False), f"{scene.name} (shared command buffer, pass {rep})")
            res.cmd = own
            res.destroy()
    shared.destroy()
    dev.destroy()


def test_dependent_command_buffers_of_one_submit_are_not_batched(mirhi, oracle, scenes):
    """mirhi_queue_submit([A, B]) where A clears and draws into a target and B LOADs that target and draws on top: the two
    must run in order, not as one batched launch -- the result is the oracle's two-draw frame."""
    dev = mirhi.Device(0)
    dev.set_queue_lanes(2)
    sa = scenes.random_triangles(1500, 320, 200, seed=31)
    sb = scenes.random_triangles(1500, 320, 200, seed=32)
    both = copy.deepcopy(sa)
    both.draws = [copy.deepcopy(sa.draws[0]), copy.deepcopy(sb.draws[0])]
    for d in both.draws:
        d.depth_test = d.depth_write = False                 # (no depth: B simply paints over A, in primitive order)
    sa.draws[0].depth_test = sa.draws[0].depth_write = False
    sb.draws[0].depth_test = sb.draws[0].depth_write = False
    target = mirhi.Image(dev, 320, 200, mirhi.Format.R32G32B32A32_SFLOAT)
    ra = mirhi.SceneResources(dev, sa, color_image=target)
    rb = mirhi.SceneResources(dev, sb, color_image=target, color_load_op=mirhi.LoadOp.LOAD)
    ra.cmd.set_queue_lane(0)
    rb.cmd.set_queue_lane(1)                                 # another lane: the submit has to order B behind A all the same
    ra.record(); rb.record()
    fence = mirhi.Fence(dev)
    for _ in range(3):
        dev.submit([ra.cmd, rb.cmd], fence)
        fence.wait(); fence.reset()
    got = target.read()
    ref = oracle.render(both, want_bgra8=False)
    err = np.abs(got[..., :3] - ref["rgba"][..., :3]).max()
    assert err < 1e-4, f"LOAD after CLEAR in one submit: max |dRGB| = {err}"
    rb.color = None                                          # (the target is shared: ra's destroy() takes it down)
    ra.destroy(); rb.destroy(); fence.destroy()
    dev.destroy()


def test_submit_thread_keeps_fence_and_error_semantics(mirhi, oracle, scenes):
    """mirhi_device_set_submit_thread: submit() returns before the launches are made.  A fence still waits for its own submission, status
    queries say NOT_READY until then, wait_idle / read-backs see every queued frame, a device-side failure (big-list overflow forced by a
    tiny pool) still surfaces at the fence, and switching the thread off again leaves a working device."""
    dev = mirhi.Device(0)
    dev.set_queue_lanes(2)
    dev.set_submit_thread(True)
    scene = scenes.random_triangles(5000, 640, 360, seed=9)
    ref = oracle.render(scene, want_bgra8=False)
    slots = [mirhi.SceneResources(dev, scene, mirhi.Format.R32G32B32A32_SFLOAT, want_prim=True) for _ in range(2)]
    fences = [mirhi.Fence(dev) for _ in range(2)]
    for it in range(40):
        k = it % 2
        if it >= 2:
            fences[k].wait(); fences[k].reset()
        slots[k].cmd.reset(); slots[k].record()
        slots[k].render(fences[k])
    for k in range(2):
        fences[k].wait()
        assert fences[k].is_signaled()
        assert np.array_equal(slots[k].read()["prim"], ref["prim"])
    # submissions without a fence: wait_idle sees them all
    for it in range(16):
        slots[it % 2].render()
    dev.wait_idle()
    assert dev.stats().frames_submitted >= 56
    dev.set_submit_thread(False)
    slots[0].render(); dev.wait_idle()
    assert np.array_equal(slots[0].read()["prim"], ref["prim"])
    for sl in slots:
        sl.destroy()
    for f in fences:
        f.destroy()
    dev.destroy()


@native_off
def test_native_dispatch_carries_the_frame_loop(mirhi, scenes):
    """The kernels of a plain submit go out as AQL packets on the library's own ROCr queues (csrc/mirhi_native.h): the device counts them.
    A timed (profiling) submit goes through HIP launches and does not."""
    dev = mirhi.Device(0)
    scene = scenes.random_triangles(2000, 640, 360, seed=4)
    res = mirhi.SceneResources(dev, scene, mirhi.Format.B8G8R8A8_SRGB)
    f = mirhi.Fence(dev)
    n0 = dev.stats().native_dispatches
    for _ in range(5):
        res.render(f); f.wait(); f.reset()
    n1 = dev.stats().native_dispatches
    assert n1 - n0 == 10, f"expected 5 frames x (geometry + raster) native dispatches, counted {n1 - n0}"
    dev.set_profiling(mirhi.Profile.TIMING)
    res.render(f); f.wait(); f.reset()
    dev.set_profiling(0)
    assert dev.stats().native_dispatches == n1
    res.destroy(); f.destroy(); dev.destroy()


def test_native_frame_loop_rewrites_its_uniform_block_every_frame(mirhi, oracle, scenes):
    """mirhost_frame_desc.per_frame_uniform: one object block per frame in flight, rewritten (Buffer::write_data, buffer.rs:247-279) by its
    frame before recording.  Uniform buffers live in host-written fine-grained memory; the frames are the oracle's all the same."""
    from renderer_rs_amd import frameloop
    dev = mirhi.Device(0)
    dev.set_queue_lanes(2)
    scene = scenes.displaced_sphere(30, 23, 320, 200, seed=5)
    res = mirhi.SceneResources(dev, scene, mirhi.Format.B8G8R8A8_SRGB)
    images = [mirhi.Image(dev, scene.width, scene.height, mirhi.Format.B8G8R8A8_SRGB) for _ in range(3)]
    ref = oracle.render(scene, want_bgra8=True)["bgra8"].astype(np.int32)
    for vary in (0, 5):
        loop = frameloop.FrameLoop(dev, res, images, frames_in_flight=2, vary_triangles=vary, per_frame_uniform=mirhi.Slot.OBJECT)
        loop.run(11)                                         # (frame 10 draws count - 3 * (10 % 5) = count vertices)
        img, rendered = loop.last_image()
        assert rendered == 11
        d = np.abs(img.read().astype(np.int32) - ref)
        assert d.max() <= 1, f"vary {vary}: sRGB8 differs by {d.max()} LSB"
        loop.destroy()
    for im in images:
        im.destroy()
    res.destroy()
    dev.destroy()


def test_write_data_waits_for_the_frames_that_read_the_buffer_and_no_others(mirhi, oracle, scenes):
    """Buffer::write_data on a uniform buffer that a PENDING frame reads: the write lands behind that frame (it renders with the old
    block), the next submit sees the new one.  A frame in flight on another lane that does not read the buffer is not waited for -- which
    shows as nothing here but is what keeps a frame loop with per-frame uniform blocks overlapped."""
    dev = mirhi.Device(0)
    dev.set_queue_lanes(2)
    base = scenes.displaced_sphere(40, 31, 384, 240, seed=11)
    other = mirhi.SceneResources(dev, scenes.random_triangles(6000, 640, 360, seed=2), mirhi.Format.R32G32B32A32_SFLOAT, want_prim=True)
    other.cmd.set_queue_lane(1)
    other_ref = oracle.render(other.scene, want_bgra8=False)
    sc = copy.deepcopy(base)
    res = mirhi.SceneResources(dev, sc, mirhi.Format.R32G32B32A32_SFLOAT, want_prim=True)
    res.cmd.set_queue_lane(0)
    blocks = [scenes.object_ubo(scenes.trs((1.0, 1.0 - 0.05 * k, 1.0), scenes.quat_axis_angle((0.0, 1.0, 0.0), 0.4 * k), (0.03 * k, 0.0, 0.0))) for k in range(6)]
    res.draw_state[0]["object"].write_data(0, blocks[0])
    for k in range(5):
        res.render()                                         # frame with block k, not waited for
        other.render()                                       # a frame on the other lane that does not read the block
        res.draw_state[0]["object"].write_data(0, blocks[k + 1])      # must land behind the pending frame
        dev.wait_idle()
        sc.draws[0].object = blocks[k]
        _check(_read(res), oracle.render(sc, want_bgra8=False), f"frame rendered before write {k + 1}")
    res.render(); dev.wait_idle()
    sc.draws[0].object = blocks[5]
    _check(_read(res), oracle.render(sc, want_bgra8=False), "frame rendered after the last write")
    _check(_read(other), other_ref, "the other lane's frame")
    res.destroy(); other.destroy(); dev.destroy()


def test_write_data_from_one_thread_while_another_runs_the_frame_loop(mirhi, oracle, scenes):
    """write_data looks through the device's pending command buffers for readers of the buffer while another host thread re-records and
    submits its own (the native loop, every frame another triangle count so that every end() rebuilds its plan): both threads' frames are
    the oracle's, nothing deadlocks, nothing is read while it changes."""
    import threading
    from renderer_rs_amd import frameloop
    dev = mirhi.Device(0)
    dev.set_queue_lanes(3)
    loop_scene = scenes.random_triangles(3000, 640, 360, seed=21)
    loop_res = mirhi.SceneResources(dev, loop_scene, mirhi.Format.B8G8R8A8_SRGB)
    images = [mirhi.Image(dev, loop_scene.width, loop_scene.height, mirhi.Format.B8G8R8A8_SRGB) for _ in range(3)]
    loop = frameloop.FrameLoop(dev, loop_res, images, frames_in_flight=2, vary_triangles=5)
    base = scenes.displaced_sphere(24, 17, 256, 160, seed=3)
    sc = copy.deepcopy(base)
    res = mirhi.SceneResources(dev, sc, mirhi.Format.R32G32B32A32_SFLOAT, want_prim=True)
    res.cmd.set_queue_lane(2)
    errors = []

    def run_loop():
        try:
            for _ in range(12):
                loop.run(50)
        except Exception as e:          # pragma: no cover
            errors.append(e)

    t = threading.Thread(target=run_loop)
    t.start()
    fence = mirhi.Fence(dev)
    for k in range(40):
        block = scenes.object_ubo(scenes.trs((1.0, 1.0 - 0.01 * k, 1.0), scenes.quat_axis_angle((0.0, 1.0, 0.0), 0.1 * k), (0.0, 0.0, 0.0)))
        res.draw_state[0]["object"].write_data(0, block)
        res.cmd.reset(); res.record()
        res.render(fence); fence.wait(); fence.reset()
    t.join()
    assert not errors, errors
    sc.draws[0].object = block
    _check(_read(res), oracle.render(sc, want_bgra8=False), "the writing thread's last frame")
    img, rendered = loop.last_image()
    assert rendered == 600
    lsc = copy.deepcopy(loop_scene)
    lsc.draws[0].count -= 3 * ((rendered - 1) % 5)
    d = np.abs(img.read().astype(np.int32) - oracle.render(lsc, want_bgra8=True)["bgra8"].astype(np.int32))
    assert d.max() <= 1
    loop.destroy(); fence.destroy()
    for im in images:
        im.destroy()
    res.destroy(); loop_res.destroy(); dev.destroy()


@pytest.mark.parametrize("upload", ["direct", "copy"])
def test_buffers_rewritten_between_frames_are_seen_by_the_next_frame(mirhi, oracle, scenes, monkeypatch, upload):
    """Between two submits of one command buffer the host rewrites the vertex buffer (plain device memory behind a copy engine: the next
    submit's first packet acquires at system scope, mirhi_device::foreign_writes) and the object block (host-written fine-grained memory,
    or -- MIRHI_PARAM_UPLOAD=copy, the fallback -- plain memory behind copies, parameter block included).  Every frame is the oracle's frame
    of what the buffers held when it was submitted."""
    if upload == "copy":
        monkeypatch.setenv("MIRHI_PARAM_UPLOAD", "copy")
    dev = mirhi.Device(0)
    base = scenes.displaced_sphere(36, 27, 384, 240, seed=7)
    sc = copy.deepcopy(base)
    res = mirhi.SceneResources(dev, sc, mirhi.Format.R32G32B32A32_SFLOAT, want_prim=True)
    fence = mirhi.Fence(dev)
    verts0 = np.array(base.draws[0].vertices, dtype=np.float32, copy=True)
    for k in range(5):
        v = verts0.copy()
        v[:, 0:3] *= np.float32(1.0 - 0.06 * k)              # positions: another mesh every frame
        sc.draws[0].vertices = v
        sc.draws[0].object = scenes.object_ubo(scenes.trs((1.0, 1.0, 1.0), scenes.quat_axis_angle((0.0, 1.0, 0.0), 0.3 * k), (0.01 * k, 0.0, 0.0)))
        res.draw_state[0]["vb"].write_data(0, v)
        res.draw_state[0]["object"].write_data(0, sc.draws[0].object)
        if k % 2:                                            # re-recorded or resubmitted as it is
            res.cmd.reset(); res.record()
        res.render(fence); fence.wait(); fence.reset()
        _check(_read(res), oracle.render(sc, want_bgra8=False), f"{upload}: frame {k}")
    res.destroy(); fence.destroy(); dev.destroy()


def test_a_deep_queue_of_different_command_buffers_on_one_lane(mirhi, oracle, scenes):
    """A thousand submissions on ONE queue lane without a single wait, alternating between command buffers of different shape (other kernels'
    arguments every packet): the kernel-argument ring of the lane's AQL queue must not hand a slot out again before the packet that reads it has
    finished (round 3's first version could, once more than 819 packets were outstanding).  Both targets end up as the oracle's frames."""
    dev = mirhi.Device(0)
    a = scenes.random_triangles(2500, 640, 360, seed=5)
    b = scenes.displaced_sphere(24, 17, 320, 200, seed=9)
    ra = mirhi.SceneResources(dev, a, mirhi.Format.R32G32B32A32_SFLOAT, want_prim=True)
    rb = mirhi.SceneResources(dev, b, mirhi.Format.R32G32B32A32_SFLOAT, want_prim=True)
    ra.cmd.set_queue_lane(0); rb.cmd.set_queue_lane(0)
    ra.record(); rb.record()
    for k in range(1500):
        (ra if k % 2 == 0 else rb).render()
    dev.wait_idle()
    _check(_read(ra), oracle.render(a, want_bgra8=False), "scene A behind 1,500 queued submissions")
    _check(_read(rb), oracle.render(b, want_bgra8=False), "scene B behind 1,500 queued submissions")
    ra.destroy(); rb.destroy(); dev.destroy()


def test_a_fence_behind_milliseconds_of_work_times_out_and_then_completes(mirhi, oracle, scenes):
    """A fence whose submission sits behind ~10 ms of queued frames: wait(0.5 ms) reports TIMEOUT (the wait polls, then blocks in the runtime in
    slices and still honours the timeout), the status is NOT_READY meanwhile, an unbounded wait then returns with the frame complete and right."""
    import time
    dev = mirhi.Device(0)
    scene = scenes.heightfield_grid(256, 256, 1920, 1080)           # 131k triangles: a few tens of microseconds per frame
    res = mirhi.SceneResources(dev, scene, mirhi.Format.R32G32B32A32_SFLOAT, want_prim=True)
    fence = mirhi.Fence(dev)
    res.render(); dev.wait_idle()
    for _ in range(300):                                             # >= 9 ms of queued work on this command buffer's lane
        res.render()
    res.render(fence)
    t0 = time.perf_counter()
    with pytest.raises(mirhi.RhiError) as e:
        fence.wait(timeout_ns=500_000)
    waited = time.perf_counter() - t0
    assert e.value.code == mirhi.TIMEOUT
    assert 0.0004 < waited < 0.05, f"a 0.5 ms timeout returned after {1e3 * waited:.2f} ms"
    assert not fence.is_signaled()
    fence.wait()
    assert fence.is_signaled()
    _check(_read(res), oracle.render(scene, want_bgra8=False), "frame behind the long queue")
    res.destroy(); fence.destroy(); dev.destroy()


@native_off
def test_a_device_on_the_callers_stream_keeps_stream_order_on_lane_0():
    """mirhi_device_create_on_stream promises that the work is issued on the caller's stream (include/mirhi.h): a producer on that stream (here torch
    filling the wrapped vertex buffer), the submit, and a consumer on the stream (torch copying the wrapped target) need no host synchronisation in
    between.  Lane 0 of such a device therefore stays on HIP launches -- native dispatch (AQL packets on the library's own queues, unordered against
    the stream) is what the lanes the library makes itself get, and lane 0 only after mirhi_device_set_native_dispatch(1).  (A child process that
    imports torch first, as bench.py does.)"""
    import subprocess
    import sys
    script = r'''
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.environ["MIRHI_ROOT"]); sys.path.insert(0, os.path.join(os.environ["MIRHI_ROOT"], "tests"))
import __graft_entry__ as ge
mirhi = ge.load_package()
import oracle_binding as oracle
scenes = mirhi.scenes
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    dev = mirhi.Device(0, stream=stream.cuda_stream)
    assert "lane 0 stays on the caller's HIP stream" in dev.dispatch_path(), dev.dispatch_path()
    a, b = scenes.random_triangles(600, 320, 200, seed=41), scenes.random_triangles(600, 320, 200, seed=42)
    raw = lambda s: np.ascontiguousarray(s.draws[0].vertices).view(np.uint8).reshape(-1).copy()
    verts = torch.from_numpy(raw(a)).cuda()
    frame = torch.zeros((a.height, a.width, 4), dtype=torch.uint8, device="cuda")
    target = mirhi.Image(dev, a.width, a.height, mirhi.Format.B8G8R8A8_SRGB, device_ptr=frame.data_ptr())
    res = mirhi.SceneResources(dev, a, mirhi.Format.B8G8R8A8_SRGB, color_image=target,
                               wrap_buffers=lambda d, usage, arr: mirhi.Buffer.wrap(d, usage, verts.data_ptr(), verts.numel()))
    dev.wait_idle()
    n0 = dev.stats().native_dispatches
    staged = [torch.from_numpy(raw(b)).pin_memory(), torch.from_numpy(raw(a)).pin_memory()]
    outs = []
    for k in range(6):                                   # alternate the two vertex sets; every step: stream copy -> submit -> stream copy, no host wait
        verts.copy_(staged[k % 2], non_blocking=True)    # producer on the caller's stream
        res.render()                                     # lane 0: HIP launches on that stream
        outs.append(frame.clone())                       # consumer on the caller's stream
    stream.synchronize()
    assert dev.stats().native_dispatches == n0, "lane 0 of a device on the caller's stream dispatched natively without being asked to"
    refs = [oracle.render(s, want_bgra8=True)["bgra8"].astype(np.int32) for s in (b, a)]
    for k, out in enumerate(outs):
        d = np.abs(out.cpu().numpy().astype(np.int32) - refs[k % 2])
        assert d.max() <= 1, f"step {k}: the frame does not show the vertices the stream wrote just before the submit (max diff {d.max()})"
    dev.set_native_dispatch(True)                        # opt in: the same lane now dispatches natively
    res.render(); dev.wait_idle()
    assert dev.stats().native_dispatches > n0
    res.color = None
    res.destroy(); target.destroy(); dev.destroy()
print("STREAM ORDER OK")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", script], env=dict(os.environ, MIRHI_ROOT=root), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert p.returncode == 0 and "STREAM ORDER OK" in p.stdout, p.stdout[-3000:]


@native_off
def test_a_queue_that_makes_no_progress_within_the_deadline_is_a_lost_device(tmp_path):
    """Every wait on the library's own AQL queues is bounded (MIRHI_NATIVE_TIMEOUT_MS, default 10 s without progress of the queue's read index): on expiry
    the device is marked lost, the fence wait fails with VulkanError (VK_ERROR_DEVICE_LOST), and so does every later submit -- a host thread never spins
    forever on a GPU, or a tool between the library and its queue, that stopped consuming packets.  Provoked here with a 1 ms deadline and one frame of
    ~40 ms (thousands of full-screen triangles: every tile walks the whole large-triangle list), in a child process."""
    import subprocess
    import sys
    script = r'''
import os, sys, time
sys.path.insert(0, os.environ["MIRHI_ROOT"])
import numpy as np
import __graft_entry__ as ge
m = ge.load_package()
S = m.scenes
W, H, N = 1920, 1080, 6000
tri = np.array([[-1, -1], [3, -1], [-1, 3]], dtype=np.float32)
v = np.zeros((N * 3, 6), dtype=np.float32)
v[:, 0:2] = np.tile(tri, (N, 1)); v[:, 2] = np.repeat(np.linspace(0.9, 0.1, N, dtype=np.float32), 3); v[:, 3:6] = 0.5
scene = S.Scene("overdraw", W, H, [S.DrawSpec(vertices=v, stride=24, count=N * 3, cull_mode=S.CULL_NONE)])
dev = m.Device(0)
res = m.SceneResources(dev, scene, m.Format.B8G8R8A8_SRGB)
f = m.Fence(dev)
t0 = time.time(); res.render(f)
try:
    f.wait()
    print("NOT LOST: the frame took", time.time() - t0)
except m.RhiError as e:
    print("WAIT:", e.variant, "DEVICE_LOST" in str(e))
print("STATS lost", dev.stats().device_lost)
try:
    res.render()
    print("SUBMIT accepted")
except m.RhiError as e:
    print("SUBMIT:", e.variant, "DEVICE_LOST" in str(e))
time.sleep(1.0)                      # the GPU is fine and finishes the frame: nothing is freed under a running kernel
os._exit(0)
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MIRHI_ROOT=root, MIRHI_NATIVE_TIMEOUT_MS="1")
    p = subprocess.run([sys.executable, "-c", script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert "WAIT: VulkanError True" in p.stdout, p.stdout[-2000:]
    assert "STATS lost 1" in p.stdout and "SUBMIT: VulkanError True" in p.stdout, p.stdout[-2000:]
    assert "mirhi: device lost: waiting for a fence: no progress within" in p.stdout, p.stdout[-2000:]


@native_off
def test_the_library_and_its_code_object_carry_the_build_of_these_sources(mirhi):
    """round-3 verdict item 4: libmirhi.so and libmirhi_kernels.hsaco are a pair by content, not by file name -- both carry the hash of the sources they were built
    from (mirhi_build_id; the device symbol mirhi::g_build_id is compared at native_device_open), and a device that dispatches natively has checked it."""
    from renderer_rs_amd import build as mbuild
    lib_id = mirhi.lib().mirhi_build_id().decode()
    assert lib_id == mbuild.source_hash(), f"library built from {lib_id}, sources are {mbuild.source_hash()}: rebuild (python -c 'import __graft_entry__ as g; g.build()')"
    dev = mirhi.Device(0)
    assert dev.dispatch_path().startswith("native"), dev.dispatch_path()
    dev.destroy()


@native_off
def test_round_trip_probe_of_a_queue_lane(mirhi):
    """mirhi_device_measure_roundtrip (bench.py's chain_us): an empty one-wave kernel and a bare barrier packet on the lane's own AQL queue, doorbell -> host sees the
    signal; microseconds, a handful of them on this part, and the empty kernel is not faster than the packet alone by more than noise."""
    dev = mirhi.Device(0)
    kernel_us, barrier_us = dev.measure_roundtrip(0, 100)
    assert 1.0 < barrier_us < 1000.0 and 1.0 < kernel_us < 1000.0, (kernel_us, barrier_us)
    assert kernel_us > 0.5 * barrier_us
    dev.destroy()
