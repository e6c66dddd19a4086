"""Renders N frames of c2 with K frames in flight (for rocprofv3 --kernel-trace). usage: prof_frames_in_flight.py [frames] [in_flight]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
m = ge.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
k = int(sys.argv[2]) if len(sys.argv) > 2 else 4
scene = m.scenes.random_triangles()
dev = m.Device(0)
dev.set_queue_lanes(k)
res = [m.SceneResources(dev, scene, m.Format.B8G8R8A8_SRGB) for _ in range(k)]
fences = [m.Fence(dev, signaled=True) for _ in range(k)]
for i in range(n):
    f = fences[i % k]
    f.wait(); f.reset()
    res[i % k].render(f)
dev.wait_idle()
print("done", n, k)
for f in fences: f.destroy()
for r in res: r.destroy()
dev.destroy()
