#!/usr/bin/env python3
"""The reference-shaped frame loop (every frame re-recorded and fenced, natively through libmirhost.so) on a BASELINE scene for a list of
(frames in flight, queue lanes) pairs, on a device with its own stream (native dispatch on every lane).
usage: fif_probe.py [c2|c3|c4|c5] [frames] [fif:lanes,...]        env: MIRHI_GEOM_TPW, ... (read once per process)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
m = ge.load_package()
from renderer_rs_amd import frameloop
import numpy as np


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "c2"
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
    pairs = [tuple(int(x) for x in p.split(":")) for p in (sys.argv[3] if len(sys.argv) > 3 else "1:1,2:1,2:2,4:1,4:4").split(",")]
    scene = {"c2": m.scenes.random_triangles, "c3": m.scenes.displaced_sphere, "c4": m.scenes.heightfield_grid, "c5": m.scenes.box_hall}[which]()
    tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("MIRHI_"))
    for fif, lanes in pairs:
        dev = m.Device(0)
        dev.set_queue_lanes(lanes)
        res = m.SceneResources(dev, scene, m.Format.B8G8R8A8_SRGB)
        res.render(); ref = res.read()["color"]
        images = [m.Image(dev, scene.width, scene.height, m.Format.B8G8R8A8_SRGB) for _ in range(fif + 1)]
        loop = frameloop.FrameLoop(dev, res, images, frames_in_flight=fif)
        loop.run(max(64, frames // 8))
        s0 = dev.stats()
        t = [loop.run(frames) for _ in range(3)]
        s1 = dev.stats()
        loop.phase_seconds(True)
        loop.run(frames)
        ph = loop.phase_seconds(False)
        img, n = loop.last_image()
        same = bool(np.array_equal(img.read(), ref))
        us = 1e6 * min(t) / frames
        print(f"{which} [{tag}] fif {fif} lanes {lanes}: {us:.2f} us/frame = {scene.num_triangles / us:.0f} Mtris/s  (host: wait {1e6 * ph[0] / frames:.2f} record {1e6 * ph[1] / frames:.2f} "
              f"end {1e6 * ph[2] / frames:.2f} submit {1e6 * ph[3] / frames:.2f}; native dispatches {s1.native_dispatches - s0.native_dispatches} for {s1.frames_submitted - s0.frames_submitted} frames; "
              f"last frame == reference: {same}; {dev.dispatch_path()})", flush=True)
        loop.destroy()
        for im in images: im.destroy()
        res.destroy(); dev.destroy()


if __name__ == "__main__":
    main()
