"""Same triangles as ONE draw and as N draws (own object UBO each): what does the draw count cost? usage: many_draws.py [n_draws] [tris_per_draw]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
m = ge.load_package()
S = m.scenes
nd = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
per = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = m.Device(0)
for program in (S.PROGRAM_TRIANGLE, S.PROGRAM_MODEL):
    if program == S.PROGRAM_TRIANGLE:
        base = S.random_triangles(nd * per, 1920, 1080, seed=5, rmin=2, rmax=12)
        d0 = base.draws[0]
        v = d0.vertices.reshape(-1, 6)
        v[:, 3:6] = np.random.default_rng(1).uniform(0, 1, v[:, 3:6].shape)      # not flat: the fragment program runs
        one = [S.DrawSpec(vertices=v, stride=24, count=v.shape[0], cull_mode=S.CULL_NONE)]
        many = [S.DrawSpec(vertices=v[i * per * 3:(i + 1) * per * 3].copy(), stride=24, count=per * 3, cull_mode=S.CULL_NONE) for i in range(nd)]
    else:
        g = S.heightfield_grid()          # 1M triangles in one indexed draw
        d0 = g.draws[0]
        idx = d0.indices[: nd * per * 3]
        mk = lambda ind: S.DrawSpec(vertices=d0.vertices, stride=48, count=ind.size, indices=ind, program=program, cull_mode=d0.cull_mode,
                                    front_face=d0.front_face, camera=d0.camera, object=d0.object)
        one = [mk(idx)]
        many = [mk(idx[i * per * 3:(i + 1) * per * 3]) for i in range(nd)]
    for name, draws in (("1 draw", one), (f"{nd} draws", many)):
        scene = S.Scene(name, 1920, 1080, draws)
        t0 = time.perf_counter()
        res = m.SceneResources(dev, scene, m.Format.B8G8R8A8_SRGB)
        t_rec = time.perf_counter() - t0
        for _ in range(5): res.render()
        dev.wait_idle()
        dev.set_profiling(True); dev.reset_kernel_times()
        for _ in range(50): res.render()
        dev.wait_idle()
        g_, gn = dev.kernel_time(m.Kernel.GEOMETRY); r_, rn = dev.kernel_time(m.Kernel.RASTER)
        dev.set_profiling(False)
        print(f"program {program} {name:12s} ({nd * per} tris): record+build {1e3 * t_rec:7.1f} ms  geometry {1e3 * g_ / max(gn, 1):8.2f} us  raster {1e3 * r_ / max(rn, 1):8.2f} us")
        res.destroy()
dev.destroy()

# per-frame cost of re-recording a command buffer with many draws (resources already exist)
dev = m.Device(0)
base = S.random_triangles(nd * per, 1920, 1080, seed=5, rmin=2, rmax=12).draws[0].vertices.reshape(-1, 6)
many = [S.DrawSpec(vertices=base[i * per * 3:(i + 1) * per * 3].copy(), stride=24, count=per * 3, cull_mode=S.CULL_NONE) for i in range(nd)]
res = m.SceneResources(dev, S.Scene("many", 1920, 1080, many), m.Format.B8G8R8A8_SRGB)
t0 = time.perf_counter()
for _ in range(5): res.record()
dt = (time.perf_counter() - t0) / 5
print(f"re-recording {nd} draws: {1e3 * dt:.2f} ms per command buffer ({1e6 * dt / nd:.2f} us per draw incl. the Python binding)")
res.destroy(); dev.destroy()
