"""Kernel times of an opaque scene plus N alpha-blended triangles on top (ordered segment), 1920x1080. usage: blend_times.py [counts...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
m = ge.load_package()
S = m.scenes
counts = [int(a) for a in sys.argv[1:]] or [100, 1000, 10000]
dev = m.Device(0)
for n in counts:
    opaque = S.random_triangles(10000).draws[0]
    over = S.random_triangles(n, seed=99).draws[0]
    over.blend = S.ALPHA_BLEND
    over.depth_write = False
    scene = S.Scene("blend", 1920, 1080, [opaque, over], clear_color=(0.1, 0.1, 0.15, 1.0))
    res = m.SceneResources(dev, scene, m.Format.B8G8R8A8_SRGB)
    for _ in range(20): res.render()
    dev.wait_idle()
    dev.set_profiling(True); dev.reset_kernel_times()
    for _ in range(200): res.render()
    dev.wait_idle()
    g, gn = dev.kernel_time(m.Kernel.GEOMETRY); r, rn = dev.kernel_time(m.Kernel.RASTER)
    dev.set_profiling(False)
    print(f"10000 opaque + {n:6d} blended triangles: geometry {1e3 * g / max(gn, 1) * 2:7.2f} us  raster+ordered {1e3 * r / max(rn, 1) * 2:7.2f} us per frame (2 segments)")
    res.destroy()
dev.destroy()
