"""Isolated kernel times (HIP event pairs, one frame at a time) for a few triangle counts at 1920x1080.
usage: kernel_times.py [counts...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
m = ge.load_package()
counts = [int(a) for a in sys.argv[1:]] or [1, 1000, 10000, 40000]
dev = m.Device(0)
for n in counts:
    res = m.SceneResources(dev, m.scenes.random_triangles(n), m.Format.B8G8R8A8_SRGB)
    for _ in range(50): res.render()
    dev.wait_idle()
    dev.set_profiling(True); dev.reset_kernel_times()
    for _ in range(500): res.render()
    dev.wait_idle()
    g, gn = dev.kernel_time(m.Kernel.GEOMETRY); r, rn = dev.kernel_time(m.Kernel.RASTER)
    dev.set_profiling(False)
    print(f"{n:7d} triangles: geometry {1e3 * g / max(gn, 1):7.2f} us  raster {1e3 * r / max(rn, 1):7.2f} us")
    res.destroy()
dev.destroy()
