"""Soak of round 3's steady-state path: ONE command buffer (one workspace, never cleared by the host after its first frame) is re-recorded with random scene
after random scene of the differential fuzzers -- other sizes, programs, depth states, segments, blended and masked draws -- and submitted twice each, natively
dispatched, with MIRHI_VERIFY_IDLE checking the idle state of counters and page table on the host at every end().  usage: soak_reuse.py [n_seeds] [first_seed]"""
import importlib.util, os, sys, time
os.environ["MIRHI_VERIFY_IDLE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
m = ge.load_package()
import oracle_binding as ob
spec = importlib.util.spec_from_file_location("fz", os.path.join(ROOT, "tests", "test_gpu_fuzz.py"))
fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
first = int(sys.argv[2]) if len(sys.argv) > 2 else 300000
dev = m.Device(0)
shared = m.CommandBuffer(dev)
fence = m.Fence(dev)
bad = 0
t0 = time.time()
for i in range(n):
    seed = first + i
    for kind, make in (("state", fz._random_scene), ("pbr", fz._random_pbr_scene)):
        scene = make(m.scenes, seed)
        if kind == "state" and i % 3 == 1:
            fz._randomize_states(m.scenes, scene, np.random.default_rng(seed))
        try:
            res = m.SceneResources(dev, scene, want_prim=True, want_depth=True)
            own, res.cmd = res.cmd, shared
            res.record()
            for _ in range(2):
                res.render(fence); fence.wait(); fence.reset()
            out = {"color": res.color.read(), "prim": res.prim.read(), "depth": res.depth.read()}
            res.cmd = own
            res.destroy()
            ref = ob.render(scene, want_bgra8=False)
            ok = np.array_equal(out["prim"], ref["prim"])
            a, b = out["color"][..., :4], ref["rgba"]
            nan = np.isnan(b)
            err = float((np.abs(np.where(nan, 0, a) - np.where(nan, 0, b)) / np.maximum(1.0, np.abs(np.where(nan, 0, b)))).max())
            if not ok or err >= 1e-4 or not np.array_equal(np.isnan(a), nan):
                bad += 1
                print(f"MISMATCH seed {seed} {kind}: prim equal {ok}, max err {err:.3e}", flush=True)
        except Exception as e:
            bad += 1
            print(f"ERROR seed {seed} {kind}: {e!r}", flush=True)
    if (i + 1) % 100 == 0:
        print(f"{i + 1} seeds ({2 * (i + 1)} scenes), {bad} bad, {time.time() - t0:.0f} s, native dispatches so far {dev.stats().native_dispatches}", flush=True)
print(f"done: {n} seeds, {bad} bad, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
