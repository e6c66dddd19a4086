"""Diagnostic (stamps build): renders C2 with the raster kernel cut after stage N (MIRHI_STAGE env: 0 = whole kernel,
1 = prologue, 2 = + fill of the first chunk, 3 = + raster loops).  Run under rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU and
difference the per-launch counts.  usage: stage_counts.py [frames] [workload: c2 c3 c4 c5]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MIRHI_LIB_NAME"] = "libmirhi_stamps.so"
import __graft_entry__ as ge
m = ge.load_package()
L = C.CDLL(m.LIB_PATH)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
wl = sys.argv[2] if len(sys.argv) > 2 else "c2"
scene = {"c2": m.scenes.random_triangles, "c3": m.scenes.displaced_sphere, "c4": m.scenes.heightfield_grid, "c5": m.scenes.box_hall}[wl]()
dev = m.Device(0)
res = m.SceneResources(dev, scene, m.Format.B8G8R8A8_SRGB)
assert L.mirhi_debug_set_stage_limit(C.c_uint32(int(os.environ.get("MIRHI_STAGE", "0")))) == 0
for _ in range(n):
    res.render()
    dev.wait_idle()
print("done")
res.destroy(); dev.destroy()
