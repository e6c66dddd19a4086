#!/usr/bin/env python3
"""EXPERIMENT (needs tools/experiments/native_overlap.patch applied and the library rebuilt; MIRHI_NATIVE_OVERLAP=1 | 2; not in the product): with MORE command buffers than queue lanes, a lane's next scope belongs to another command
buffer than the one in front of it and its vertex / geometry kernels drop the barrier bit.  Deep-queue and fenced frame loops with 4 and 8 command buffers
on 4 lanes, with and without the overlap.  usage: overlap_probe.py [c2 c3 ...]"""
import os, sys, subprocess, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    m = ge.load_package()
    from renderer_rs_amd import frameloop
    wl, ncmd = sys.argv[2], int(sys.argv[3])
    scene = {"c2": m.scenes.random_triangles, "c3": m.scenes.displaced_sphere, "c4": m.scenes.heightfield_grid, "c5": m.scenes.box_hall}[wl]()
    dev = m.Device(0)
    dev.set_queue_lanes(4)
    slots = [m.SceneResources(dev, scene, m.Format.B8G8R8A8_SRGB) for _ in range(ncmd)]
    for k, s in enumerate(slots):
        s.cmd.set_queue_lane(k % 4); s.record()
    ref = None
    frames = 4000 if wl in ("c2", "c3") else 600
    for _ in range(frames // 8):
        for s in slots: s.render()
    dev.wait_idle()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for i in range(frames):
            slots[i % ncmd].render()
        dev.wait_idle()
        best = min(best, time.perf_counter() - t0)
    imgs = [s.read()["color"] for s in slots]
    same = all((imgs[0] == im).all() for im in imgs[1:])
    # fenced and re-recorded, natively: ncmd frames in flight on the 4 lanes
    images = [m.Image(dev, scene.width, scene.height, m.Format.B8G8R8A8_SRGB) for _ in range(ncmd + 1)]
    loop = frameloop.FrameLoop(dev, slots[0], images, frames_in_flight=ncmd)
    loop.run(frames // 8)
    sec = min(loop.run(frames) for _ in range(3))
    img, _n = loop.last_image()
    same = same and (img.read() == imgs[0]).all()
    print(json.dumps({"deep_us": round(1e6 * best / frames, 3), "deep_mtris": round(scene.num_triangles * frames / best / 1e6, 1), "fenced_us": round(1e6 * sec / frames, 3),
                      "fenced_mtris": round(scene.num_triangles * frames / sec / 1e6, 1), "frames_identical": bool(same)}))
    sys.exit(0)
for wl in sys.argv[1:] or ["c2", "c3"]:
    for ncmd in (4, 8):
        for label, env in (("barriers", {}), ("overlap", {"MIRHI_NATIVE_OVERLAP": "1"}), ("first kernel only", {"MIRHI_NATIVE_OVERLAP": "2"})):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", wl, str(ncmd)], env=dict(os.environ, **env), capture_output=True, text=True)
            print(f"{wl} {ncmd} command buffers on 4 lanes, {label:17s}: {r.stdout.strip() or r.stderr.strip()[-400:]}", flush=True)
