import sys, time, numpy as np
import __graft_entry__ as ge
sys.path.insert(0, 'tests')
m = ge.load_package()
import oracle_binding as ob
dev = m.Device(0)
print(dev.name())
for fn in (m.scenes.random_triangles, m.scenes.displaced_sphere, m.scenes.heightfield_grid, m.scenes.box_hall):
    scene = fn()
    for fmt in (m.Format.R32G32B32A32_SFLOAT, m.Format.B8G8R8A8_SRGB):
        res = m.SceneResources(dev, scene, fmt, want_prim=(fmt == m.Format.R32G32B32A32_SFLOAT))
        res.render(); dev.wait_idle()
        dev.set_profiling(True); dev.reset_kernel_times()
        t0 = time.perf_counter()
        N = 50
        for _ in range(N): res.render()
        dev.wait_idle()
        t1 = time.perf_counter()
        g, ng = dev.kernel_time(m.Kernel.GEOMETRY); r, nr = dev.kernel_time(m.Kernel.RASTER)
        dev.set_profiling(False)
        t2 = time.perf_counter()
        for _ in range(N): res.render()
        dev.wait_idle()
        t3 = time.perf_counter()
        st = dev.stats()
        print(f"{scene.name} fmt={fmt}: wall/frame prof {1e6*(t1-t0)/N:.1f} us, noprof {1e6*(t3-t2)/N:.1f} us; geometry {1e3*g/ng:.1f} us raster {1e3*r/nr:.1f} us; big_list {st.last_big_list} ws {st.workspace_bytes/1e6:.1f} MB; Mtris/s {scene.num_triangles/((t3-t2)/N)/1e6:.1f}")
        if fmt == m.Format.R32G32B32A32_SFLOAT:
            out = res.read()
            t4 = time.perf_counter(); ref = ob.render(scene, want_bgra8=False); t5 = time.perf_counter()
            diff = (out["prim"] != ref["prim"]).sum()
            err = np.nanmax(np.abs(out["color"][..., :3] - ref["rgba"][..., :3]) / np.maximum(1, np.abs(ref["rgba"][..., :3])))
            print(f"   parity: prim diff {diff}, max rel |dRGB| {err:.3e}, oracle {t5-t4:.2f}s")
        res.destroy()
dev.destroy()
