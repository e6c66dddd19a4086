#!/usr/bin/env python3
"""Where mirhi_cmd_end spends its host time in the fenced frame loop (variant build: python renderer-rs_amd/build.py --variant hp -DMIRHI_HOST_PROF)."""
import os, sys, ctypes as C
os.environ["MIRHI_LIB_NAME"] = "libmirhi_hp.so"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
m = ge.load_package()
from renderer_rs_amd import frameloop
scene = m.scenes.random_triangles()
dev = m.Device(0)
dev.set_queue_lanes(2)
res = m.SceneResources(dev, scene, m.Format.B8G8R8A8_SRGB)
images = [m.Image(dev, scene.width, scene.height, m.Format.B8G8R8A8_SRGB) for _ in range(3)]
loop = frameloop.FrameLoop(dev, res, images, frames_in_flight=2)
loop.run(500)
loop.phase_seconds(True)
sec = loop.run(4000)
print("us per frame", 1e6 * sec / 4000, "phases us", [round(1e6 * p / 4000, 3) for p in loop.phase_seconds(False)])
C.CDLL(m.LIB_PATH).mirhi_debug_host_prof()
loop.destroy()
