"""Soak of the tile-row split: random scenes rendered as `world` bands on one GPU, assembled, against the oracle.
usage: soak_split.py [n_seeds]"""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
m = ge.load_package()
import oracle_binding as ob
from renderer_rs_amd import multigpu
spec = importlib.util.spec_from_file_location("fz", os.path.join(ROOT, "tests", "test_gpu_fuzz.py"))
fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad = 0
for i in range(n):
    seed = 700000 + i
    rng = np.random.default_rng(seed)
    scene = (fz._random_scene if i % 2 == 0 else fz._random_pbr_scene)(m.scenes, seed)
    world = int(rng.integers(2, 9))
    layout = "bands" if (i // 2) % 2 == 0 else "interleaved"
    ref = ob.render(scene, want_bgra8=False)
    prim = np.full((scene.height, scene.width), 0xFFFFFFFF, dtype=np.uint32)
    color = np.zeros((scene.height, scene.width, 4), dtype=np.float32)
    for rank in range(world):
        dev = m.Device(0)
        dev.set_tile_split(rank, world, layout=layout)
        res = m.SceneResources(dev, scene, want_prim=True)
        res.render(); out = res.read(); res.destroy(); dev.destroy()
        for r0, r1 in multigpu.owned_pixel_rows(scene.height, rank, world, layout):
            prim[r0:r1] = out["prim"][r0:r1]; color[r0:r1] = out["color"][r0:r1]
    ok = np.array_equal(prim, ref["prim"])
    nan = np.isnan(ref["rgba"])
    err = float((np.abs(np.where(nan, 0, color) - np.where(nan, 0, ref["rgba"])) / np.maximum(1.0, np.abs(np.where(nan, 0, ref["rgba"])))).max())
    if not ok or err >= 1e-4:
        bad += 1
        print(f"MISMATCH seed={seed} world={world} layout={layout} prim_ok={ok} err={err}", flush=True)
print(f"done: {n} scenes, {bad} mismatches")
sys.exit(1 if bad else 0)
