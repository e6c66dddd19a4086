"""Soak of batched submits (several command buffers per mirhi_queue_submit -> vertex / geometry / raster_kernel_batch): random groups of
2..8 scenes of one shape and program family, each frame against the oracle.  usage: soak_batch.py [n_groups] [first_seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
m = ge.load_package()
import oracle_binding as ob
S = m.scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
first = int(sys.argv[2]) if len(sys.argv) > 2 else 900000
dev = m.Device(0)
dev.set_queue_lanes(3)
fence = m.Fence(dev)
bad = 0
t0 = time.time()
for g in range(n):
    rng = np.random.default_rng(first + g)
    k = int(rng.integers(2, 9))
    W, H = [(640, 360), (333, 217), (1280, 720), (96, 700), (1920, 1080)][int(rng.integers(0, 5))]
    family = int(rng.integers(0, 3))
    scenes = []
    for i in range(k):
        seed = int(rng.integers(1, 1 << 30))
        if family == 0:
            scenes.append(S.random_triangles(int(rng.integers(1, 6000)), W, H, seed=seed, rmin=2, rmax=float(rng.uniform(6, 90))))
        elif family == 1:
            scenes.append(S.displaced_sphere(int(rng.integers(8, 60)), int(rng.integers(6, 50)), W, H, seed=seed))
        else:   # dense small triangles (triangle-parallel variants)
            scenes.append(S.random_triangles(int(rng.integers(30000, 90000)), W, H, seed=seed, rmin=1, rmax=5))
    fmt = m.Format.B8G8R8A8_SRGB if family != 1 and rng.integers(0, 2) else m.Format.R32G32B32A32_SFLOAT
    want_prim = fmt != m.Format.B8G8R8A8_SRGB
    res = [m.SceneResources(dev, sc, fmt, want_prim=want_prim) for sc in scenes]
    for r in res:
        r.cmd.set_queue_lane(g % 3)
    for rep in range(2):
        dev.submit([r.cmd for r in res], fence)
        fence.wait()
    for r, sc in zip(res, scenes):
        out = r.read()
        ref = ob.render(sc, want_bgra8=True)
        if want_prim:
            ok = np.array_equal(out["prim"], ref["prim"])
            a, b = out["color"], ref["rgba"]
            ok = ok and float((np.abs(a - b) / np.maximum(1.0, np.abs(b))).max()) < 1e-4
        else:
            ok = int(np.abs(out["color"].astype(np.int32) - ref["bgra8"].astype(np.int32)).max()) <= 1
        if not ok:
            bad += 1
            print(f"MISMATCH group={g} scene={sc.name} k={k} {W}x{H} family={family}", flush=True)
    for r in res:
        r.destroy()
    if g % 20 == 19:
        print(f"{g + 1} groups, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print(f"done: {n} groups, {bad} mismatches")
fence.destroy(); dev.destroy()
sys.exit(1 if bad else 0)
