"""One-off soak: the differential fuzzers of tests/test_gpu_fuzz.py over many more seeds. usage: soak_fuzz.py [n_seeds] [first_seed]"""
import importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
m = ge.load_package()
import oracle_binding as ob
spec = importlib.util.spec_from_file_location("fz", os.path.join(ROOT, "tests", "test_gpu_fuzz.py"))
fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
first = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
dev = m.Device(0)
bad = 0
t0 = time.time()
for i in range(n):
    seed = first + i
    for kind, make in (("state", fz._random_scene), ("pbr", fz._random_pbr_scene)):
        scene = make(m.scenes, seed)
        if kind == "state" and i % 3 == 0:      # every third one with a predicate depth state
            rng = np.random.default_rng(seed)
            op, write = [(o, False) for o in range(8)][int(rng.integers(0, 8))]
            for d in scene.draws: d.depth_test, d.depth_write, d.depth_compare = True, write, op
        if kind == "state" and i % 3 == 1:      # every third one with a different depth state per draw (scope segments)
            rng = np.random.default_rng(seed)
            fz._randomize_states(m.scenes, scene, rng)
        try:
            res = m.SceneResources(dev, scene, want_prim=True, want_depth=True)
            res.render(); out = res.read(); res.destroy()
            ref = ob.render(scene, want_bgra8=False)
            ok = np.array_equal(out["prim"], ref["prim"])
            cov = ref["prim"] != 0xFFFFFFFF
            ok = ok and np.array_equal(out["depth"].view(np.uint32)[cov], ref["depth"].view(np.uint32)[cov])
            a, b = out["color"], ref["rgba"]
            nan = np.isnan(b)
            ok = ok and np.array_equal(np.isnan(a), nan)
            with np.errstate(invalid="ignore"):      # equal infinities (feedback blend factors overflow on both sides alike) are no error
                err = float(np.where(a == b, 0.0, np.abs(np.where(nan, 0, a) - np.where(nan, 0, b)) / np.maximum(1.0, np.abs(np.where(nan, 0, b)))).max())
            ok = ok and err < 1e-4
        except Exception as e:
            ok, err = False, repr(e)
        if not ok:
            bad += 1
            print(f"MISMATCH kind={kind} seed={seed} err={err}", flush=True)
    if i % 50 == 49: print(f"{i + 1} seeds, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print(f"done: {n} seeds x 2 generators, {bad} mismatches")
dev.destroy()
sys.exit(1 if bad else 0)
