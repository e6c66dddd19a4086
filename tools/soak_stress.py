"""Stress soak at full frame sizes: dense heaps in a few tiles (bin overflow -> big list), screen-filling triangles (big list),
slivers, odd target sizes, TP path on / off.  usage: soak_stress.py [n_cases]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
m = ge.load_package()
import oracle_binding as ob
S = m.scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = m.Device(0)
bad = 0
for i in range(n):
    rng = np.random.default_rng(800000 + i)
    W, H = [(1920, 1080), (1000, 700), (333, 517), (2560, 1440), (64, 2000)][i % 5]
    kind = i % 4
    if kind == 0:      # a heap of triangles inside a handful of tiles
        nt = int(rng.integers(2000, 12000))
        c = rng.uniform(-0.9, 0.9, (1, 2)) + rng.normal(0, 0.03, (nt, 1, 2))
        p = c + rng.normal(0, 0.02, (nt, 3, 2))
    elif kind == 1:    # screen-filling triangles plus small ones
        nt = int(rng.integers(50, 400))
        p = rng.uniform(-1.5, 1.5, (nt, 3, 2))
    elif kind == 2:    # slivers
        nt = int(rng.integers(500, 5000))
        a = rng.uniform(-1, 1, (nt, 1, 2)); d = rng.normal(0, 0.5, (nt, 1, 2))
        p = np.concatenate([a, a + d, a + d * 0.5 + rng.normal(0, 0.002, (nt, 1, 2))], axis=1)
    else:              # dense small triangles everywhere (triangle-parallel path)
        nt = int(rng.integers(50000, 200000))
        c = rng.uniform(-1, 1, (nt, 1, 2))
        p = c + rng.normal(0, 0.004, (nt, 3, 2))
    z = rng.uniform(0.0, 1.0, (nt, 3, 1)) if rng.random() < 0.5 else np.repeat(rng.uniform(0, 1, (nt, 1, 1)), 3, axis=1)
    col = np.repeat(rng.uniform(0, 1, (nt, 1, 3)), 3, axis=1) if rng.random() < 0.5 else rng.uniform(0, 1, (nt, 3, 3))
    verts = np.concatenate([p, z, col], axis=2).astype(np.float32).reshape(nt * 3, 6)
    d = S.DrawSpec(vertices=verts, stride=24, count=3 * nt, cull_mode=int(rng.integers(0, 3)),
                   depth_compare=[S.CMP_LESS, S.CMP_LESS_OR_EQUAL, S.CMP_GREATER, S.CMP_GREATER_OR_EQUAL][int(rng.integers(0, 4))])
    scene = S.Scene(f"stress-{i}", W, H, [d], clear_color=(0.1, 0.2, 0.3, 1.0), clear_depth=0.0 if d.depth_compare in (S.CMP_GREATER, S.CMP_GREATER_OR_EQUAL) else 1.0)
    fmt = m.Format.B8G8R8A8_SRGB if i % 2 else m.Format.R32G32B32A32_SFLOAT
    t0 = time.time()
    try:
        res = m.SceneResources(dev, scene, fmt, want_prim=True, want_depth=True)
        fence = m.Fence(dev)
        res.render(); res.render(fence); fence.wait(); out = res.read()      # (the stats are taken when a fence completes)
        big = dev.stats().last_big_list
        fence.destroy()
        res.destroy()
        ref = ob.render(scene, nthreads=16, want_bgra8=(i % 2 == 1))
        ok = np.array_equal(out["prim"], ref["prim"])
        cov = ref["prim"] != 0xFFFFFFFF
        ok = ok and np.array_equal(out["depth"].view(np.uint32)[cov], ref["depth"].view(np.uint32)[cov])
        if i % 2: err = int(np.abs(out["color"].astype(np.int32) - ref["bgra8"].astype(np.int32)).max()); ok = ok and err <= 1
        else: err = float(np.abs(out["color"] - ref["rgba"]).max()); ok = ok and err < 1e-4
    except Exception as e:
        ok, err, big = False, repr(e), -1
    print(f"{scene.name} kind {kind} {W}x{H} tris {nt} big_list {big} err {err} {'ok' if ok else 'MISMATCH'} ({time.time() - t0:.1f} s)", flush=True)
    bad += not ok
print(f"done: {n} cases, {bad} mismatches")
dev.destroy()
sys.exit(1 if bad else 0)
