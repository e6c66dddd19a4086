import sys, numpy as np
import __graft_entry__ as ge
sys.path.insert(0, 'tests')
m = ge.load_package()
import oracle_binding as ob
dev = m.Device(0)
scene = m.scenes.SMALL_CASES[sys.argv[1] if len(sys.argv) > 1 else "textured"]()
res = m.SceneResources(dev, scene, want_prim=True)
res.render(); out = res.read(); res.destroy()
ref = ob.render(scene, want_bgra8=False)
a, b = out["color"], ref["rgba"]
err = np.abs(a - b) / np.maximum(1, np.abs(b))
print("prim diff", (out["prim"] != ref["prim"]).sum())
idx = np.unravel_index(np.argmax(err), err.shape)
print("max err", err.max(), "at", idx, "gpu", a[idx[0], idx[1]], "ref", b[idx[0], idx[1]])
print("per-channel max", err.reshape(-1, 4).max(axis=0))
ys, xs = np.where(err.max(axis=2) > 5e-5)
print("n > 5e-5:", len(ys), list(zip(ys[:10], xs[:10])))
