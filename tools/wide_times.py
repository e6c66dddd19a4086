#!/usr/bin/env python3
"""Raster / geometry kernel durations of the mesh workloads with four and with sixteen waves per tile (MIRHI_RASTER_WIDE=0/1), isolated dispatches."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    m = ge.load_package()
    DANCER = os.path.join(ROOT, "tests", "golden", "dancer", "scene.gltf")
    make = {"c3": m.scenes.displaced_sphere, "dancer": lambda: m.scenes.gltf_model(DANCER), "dancer_tex": lambda: m.scenes.gltf_model(DANCER, textures=True),
            "c4": m.scenes.heightfield_grid, "c5": m.scenes.box_hall, "grid64": lambda: m.scenes.heightfield_grid(64, 64, 1920, 1080)}[sys.argv[2]]
    scene = make()
    dev = m.Device(0)
    res = m.SceneResources(dev, scene, m.Format.B8G8R8A8_SRGB)
    for _ in range(8):
        res.render(); dev.wait_idle()
    dev.reset_kernel_times(); dev.set_profiling(m.Profile.TIMING)
    for _ in range(64):
        res.render(); dev.wait_idle()
    out = {name: dev.kernel_time(k) for k, name in enumerate(m.Kernel.NAMES)}
    print(json.dumps({k: round(1e3 * ms / n, 2) if n else 0 for k, (ms, n) in out.items()}))
    sys.exit(0)
for wl in sys.argv[1:] or ["c3", "dancer", "dancer_tex", "grid64"]:
    for label, env in (("host's choice", {}), ("four waves", {"MIRHI_RASTER_WIDE": "0"}), ("eight waves", {"MIRHI_RASTER_WIDE": "8"}), ("sixteen waves", {"MIRHI_RASTER_WIDE": "16"}),
                       ("four waves, one team", {"MIRHI_RASTER_WIDE": "0", "MIRHI_RASTER_TEAMS": "1"}),
                       ("eight waves, one team", {"MIRHI_RASTER_WIDE": "8", "MIRHI_RASTER_TEAMS": "1"}), ("sixteen waves, one team", {"MIRHI_RASTER_WIDE": "16", "MIRHI_RASTER_TEAMS": "1"})):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", wl], env=dict(os.environ, **env), capture_output=True, text=True)
        print(f"{wl:11s} {label:26s} {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)
