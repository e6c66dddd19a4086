#!/usr/bin/env python3
"""Renders a few frames of one scene so that rocprofv3 --pmc can count its kernels' instructions:
   rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES --output-format csv -d OUT -- python3 tools/inst_probe.py <scene>
scenes: empty (one tiny triangle, 1920x1080: the per-tile fixed cost), c2, c2half (5,000 triangles), c2small (10,000 triangles of radius 2..8)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
m = ge.load_package()
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
make = {"empty": lambda: m.scenes.random_triangles(1, 1920, 1080, seed=1, rmin=1.0, rmax=1.5), "c2": m.scenes.random_triangles,
        "c2half": lambda: m.scenes.random_triangles(5000), "c2small": lambda: m.scenes.random_triangles(10000, rmin=2.0, rmax=8.0)}[which]
dev = m.Device(0)
res = m.SceneResources(dev, make(), m.Format.B8G8R8A8_SRGB)
for _ in range(6):
    res.render(); dev.wait_idle()
res.destroy(); dev.destroy()
