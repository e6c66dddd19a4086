"""Re-runs one soak-fuzz case and prints where GPU and oracle differ most. usage: fuzz_case.py <state|pbr> <seed>"""
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
first = int(sys.argv[3]) if len(sys.argv) > 3 else 500000      # the soak run's first seed (decides which cases get extra states)
scene = (fz._random_scene if kind == "state" else fz._random_pbr_scene)(m.scenes, seed)
if kind == "state" and (seed - first) % 3 == 0:
    rng = np.random.default_rng(seed)
    op, write = [(o, False) for o in range(8)][int(rng.integers(0, 8))]
    for d in scene.draws: d.depth_test, d.depth_write, d.depth_compare = True, write, op
if kind == "state" and (seed - first) % 3 == 1:
    rng = np.random.default_rng(seed)
    fz._randomize_states(m.scenes, scene, rng)
dev = m.Device(0)
res = m.SceneResources(dev, scene, want_prim=True, want_depth=True)
res.render(); out = res.read(); res.destroy()
ref = ob.render(scene, want_bgra8=False)
a, b = out["color"], ref["rgba"]
err = np.abs(a - b) / np.maximum(1.0, np.abs(b))
y, x, c = np.unravel_index(np.nanargmax(err), err.shape)
prim = int(ref["prim"][y, x])
base = 0
for di, d in enumerate(scene.draws):
    if prim < base + d.num_triangles: break
    base += d.num_triangles
print(f"{scene.name} {scene.width}x{scene.height}: max err {err[y, x, c]:.3e} at ({x},{y}) channel {c}: gpu {a[y, x]} oracle {b[y, x]}")
print(f"prim {prim} -> draw {di} program {d.program} tri {prim - base}; prim equal everywhere: {np.array_equal(out['prim'], ref['prim'])}")
if d.material is not None:
    mat = np.frombuffer(d.material, dtype=np.float32)
    print("material floats", mat[:12], "flags", np.frombuffer(d.material, dtype=np.int32)[12:17] if len(d.material) >= 68 else None)
print("pixels above 1e-4:", int((err.max(axis=2) > 1e-4).sum()), "of covered", int((ref["prim"] != 0xFFFFFFFF).sum()))
dev.destroy()
nan_a, nan_b = np.isnan(a).any(axis=2), np.isnan(b).any(axis=2)
if (nan_a != nan_b).any():
    ys, xs = np.nonzero(nan_a != nan_b)
    print("NaN pattern differs at", len(ys), "pixels; first", (xs[0], ys[0]), "gpu", a[ys[0], xs[0]], "oracle", b[ys[0], xs[0]], "prim", ref["prim"][ys[0], xs[0]])
    for di, d in enumerate(scene.draws): print(" draw", di, "program", d.program, "tris", d.num_triangles, "depth", d.depth_test, d.depth_write, d.depth_compare, "blend", d.blend)
    print(" clear", scene.clear_color)
