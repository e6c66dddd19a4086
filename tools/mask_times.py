"""Frame cost of an alpha-masked Cook-Torrance mesh (glTF alphaMode MASK: per-fragment `discard`, model_pbr.hlsl:176-179) on top of an
opaque scene, 1920x1080.  Default: the masked draws keep bins and the raster kernel (alpha tested in front of the depth key);
MIRHI_MASKED_ORDERED=1: the ordered resolve they took before.  usage: mask_times.py [quads per side ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
m = ge.load_package()
S = m.scenes
sides = [int(a) for a in sys.argv[1:]] or [16, 64, 224]
dev = m.Device(0)
n = 64
yy, xx = np.mgrid[0:n, 0:n]
leaf = np.zeros((n, n, 4), dtype=np.uint8)
leaf[..., 0] = 60; leaf[..., 1] = 150 + (xx % 8) * 8; leaf[..., 2] = 40
leaf[..., 3] = np.where(np.hypot((xx % 16) - 7.5, (yy % 16) - 7.5) < 6.5, 255, 0).astype(np.uint8)      # discs: 53 % kept
for side in sides:
    grid = S.heightfield_grid(side, side, 1920, 1080).draws[0]              # 2 * side^2 triangles, MODEL layout, uv = (0..1)^2
    verts = np.ascontiguousarray(grid.vertices).view(np.float32).reshape(-1, 12).copy()
    verts[:, 6:8] *= max(1.0, side / 4.0)                                   # tile the cut-out pattern
    view, proj, cam = S.default_camera(1920, 1080, eye=(0.0, 0.0, 5.0))
    light = S.light_ubo(direction=(0.2, -0.6, -0.8), intensity=1.2, color=(1.0, 1.0, 1.0))
    masked = S.DrawSpec(vertices=verts, stride=48, count=grid.count, indices=grid.indices, program=S.PROGRAM_MODEL_PBR, cull_mode=S.CULL_NONE,
                        camera=cam, object=grid.object, light=light,
                        material=S.pbr_material_ubo((1.0, 1.0, 1.0, 1.0), 0.0, 0.6, 1.0, alpha_cutoff=0.5, has_base_color=True),
                        albedo_map=S.Texture(leaf), alpha_test=True)
    opaque = S.random_triangles(10000).draws[0]
    scene = S.Scene("masked", 1920, 1080, [opaque, masked], clear_color=(0.1, 0.1, 0.15, 1.0))
    res = m.SceneResources(dev, scene, m.Format.B8G8R8A8_SRGB)
    for _ in range(10): res.render()
    dev.wait_idle()
    dev.set_profiling(True); dev.reset_kernel_times()
    for _ in range(50): res.render()
    dev.wait_idle()
    g, gn = dev.kernel_time(m.Kernel.GEOMETRY); r, rn = dev.kernel_time(m.Kernel.RASTER)
    dev.set_profiling(False)
    print(f"10000 opaque + {masked.num_triangles:7d} masked triangles: geometry {1e3 * g / max(gn, 1) * 2:8.2f} us  raster {1e3 * r / max(rn, 1) * 2:9.2f} us per frame (2 segments)"
          f"  [{'ordered' if os.environ.get('MIRHI_MASKED_ORDERED') == '1' else 'bins + alpha test'}]")
    res.destroy()
dev.destroy()
