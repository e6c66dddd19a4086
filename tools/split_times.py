"""Per-kernel times of one rank's share of a tile-row split (emulated on one GPU). usage: split_times.py [workload] [world]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
m = ge.load_package()
wl = sys.argv[1] if len(sys.argv) > 1 else "c4"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
scene = {"c2": m.scenes.random_triangles, "c3": m.scenes.displaced_sphere, "c4": m.scenes.heightfield_grid, "c5": m.scenes.box_hall}[wl]()
for rank in (None, 0, world // 2, world - 1):
    dev = m.Device(0)
    if rank is not None: dev.set_tile_split(rank, world)
    res = m.SceneResources(dev, scene, m.Format.B8G8R8A8_SRGB)
    for _ in range(10): res.render()
    dev.wait_idle()
    dev.set_profiling(True); dev.reset_kernel_times()
    for _ in range(100): res.render()
    dev.wait_idle()
    g, gn = dev.kernel_time(m.Kernel.GEOMETRY); r, rn = dev.kernel_time(m.Kernel.RASTER)
    print(f"{wl} rank {rank} of {world}: geometry {1e3 * g / max(gn, 1):7.2f} us  raster {1e3 * r / max(rn, 1):7.2f} us")
    res.destroy(); dev.destroy()
