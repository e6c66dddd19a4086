"""Where exactly GPU and oracle differ for one soak-fuzz case: primitive ids, depth bits, colour; per draw state.  usage: fuzz_diff.py <state|pbr> <seed> [first_seed]
env toggles (MIRHI_GEOM_TPW, MIRHI_NATIVE_DISPATCH, ...) apply as usual."""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
m = ge.load_package()
import oracle_binding as ob
spec = importlib.util.spec_from_file_location("fz", os.path.join(ROOT, "tests", "test_gpu_fuzz.py"))
fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
kind, seed = sys.argv[1], int(sys.argv[2])
first = int(sys.argv[3]) if len(sys.argv) > 3 else 500000
scene = (fz._random_scene if kind == "state" else fz._random_pbr_scene)(m.scenes, seed)
if kind == "state" and (seed - first) % 3 == 0:
    rng = np.random.default_rng(seed)
    op, write = [(o, False) for o in range(8)][int(rng.integers(0, 8))]
    for d in scene.draws: d.depth_test, d.depth_write, d.depth_compare = True, write, op
if kind == "state" and (seed - first) % 3 == 1:
    rng = np.random.default_rng(seed)
    fz._randomize_states(m.scenes, scene, rng)
dev = m.Device(0)
ref = ob.render(scene, want_bgra8=False)
for attempt in range(3):
    res = m.SceneResources(dev, scene, want_prim=True, want_depth=True)
    res.render(); out = res.read(); res.destroy()
    dp = out["prim"] != ref["prim"]
    cov = ref["prim"] != 0xFFFFFFFF
    dz = (out["depth"].view(np.uint32) != ref["depth"].view(np.uint32)) & cov
    print(f"attempt {attempt}: {scene.name} {scene.width}x{scene.height}, {scene.num_triangles} triangles in {len(scene.draws)} draws: prim differs at {int(dp.sum())} pixels, depth bits at {int(dz.sum())} covered pixels", flush=True)
    if dp.any():
        ys, xs = np.nonzero(dp)
        for y, x in list(zip(ys, xs))[:6]:
            print(f"   ({x},{y}) tile ({x // 32},{y // 32}): gpu prim {out['prim'][y, x]} depth {out['depth'][y, x]!r}  oracle prim {ref['prim'][y, x]} depth {ref['depth'][y, x]!r}")
        print("   rows", ys.min(), "..", ys.max(), "cols", xs.min(), "..", xs.max())
    elif dz.any():
        ys, xs = np.nonzero(dz)
        for y, x in list(zip(ys, xs))[:6]:
            print(f"   ({x},{y}): prim {ref['prim'][y, x]} gpu depth {out['depth'][y, x]!r} ({out['depth'].view(np.uint32)[y, x]:#x}) oracle {ref['depth'][y, x]!r} ({ref['depth'].view(np.uint32)[y, x]:#x})")
base = 0
for di, d in enumerate(scene.draws):
    print(f" draw {di}: program {d.program} tris {d.num_triangles} prims {base}..{base + d.num_triangles - 1} depth test/write/op {d.depth_test}/{d.depth_write}/{d.depth_compare} cull {d.cull_mode} blend {getattr(d, 'blend', None)} indexed {d.indices is not None} viewport {d.viewport} scissor {d.scissor}")
    base += d.num_triangles
print(" clear", scene.clear_color, scene.clear_depth, "stats", dev.stats().last_status, dev.stats().last_big_list)
dev.destroy()
