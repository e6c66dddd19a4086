"""Renders N frames of one workload (for rocprofv3). usage: prof_run.py <c2|c3|c4|c5> [frames] [bgra8|rgba32f]"""
import sys
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
m = ge.load_package()
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
fmt = m.Format.R32G32B32A32_SFLOAT if (len(sys.argv) > 3 and sys.argv[3] == "rgba32f") else m.Format.B8G8R8A8_SRGB
make = {"tri1": lambda: m.scenes.random_triangles(1), "c2": m.scenes.random_triangles, "c3": m.scenes.displaced_sphere, "c4": m.scenes.heightfield_grid, "c5": m.scenes.box_hall}[wl]
scene = make()
dev = m.Device(0)
res = m.SceneResources(dev, scene, fmt)
for _ in range(n):
    res.render()
dev.wait_idle()
print("done", scene.name, n)
res.destroy(); dev.destroy()
