"""Isolated kernel times of the real asset (tests/golden/dancer, 17,210 triangles, 1920x1080, BGRA8 sRGB): config-3
constants vs the asset's own normal map (1024^2 fixture, mip-mapped, trilinear) under model_full and model_pbr."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
m = ge.load_package()
sc = m.scenes
path = os.path.join(ROOT, "tests", "golden", "dancer", "scene.gltf")
dev = m.Device(0)
for label, kw in (("model_full, constants", dict(program=sc.PROGRAM_MODEL_FULL)),
                  ("model_full, normal map", dict(program=sc.PROGRAM_MODEL_FULL, textures=True)),
                  ("model_pbr,  constants", dict(program=sc.PROGRAM_MODEL_PBR)),
                  ("model_pbr,  normal map", dict(program=sc.PROGRAM_MODEL_PBR, textures=True))):
    t0 = time.perf_counter()
    scene = sc.gltf_model(path, **kw)
    t_load = time.perf_counter() - t0
    res = m.SceneResources(dev, scene, m.Format.B8G8R8A8_SRGB)
    for _ in range(30): res.render()
    dev.wait_idle()
    dev.set_profiling(True); dev.reset_kernel_times()
    for _ in range(300): res.render()
    dev.wait_idle()
    g, gn = dev.kernel_time(m.Kernel.GEOMETRY); r, rn = dev.kernel_time(m.Kernel.RASTER)
    dev.set_profiling(False)
    print(f"{label:24s}: load+decode {1e3 * t_load:6.1f} ms  geometry {1e3 * g / max(gn, 1):7.2f} us  raster {1e3 * r / max(rn, 1):7.2f} us")
    res.destroy()
dev.destroy()
