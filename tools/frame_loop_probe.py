#!/usr/bin/env python3
"""The reference-shaped frame loop (every frame re-recorded and fenced) run natively through libmirhost.so, on a BASELINE scene.
usage: frame_loop_probe.py [c2|c3|c4|c5] [frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
m = ge.load_package()
from renderer_rs_amd import frameloop
import numpy as np
import torch

def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "c2"
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
    scene = {"c2": m.scenes.random_triangles, "c3": m.scenes.displaced_sphere, "c4": m.scenes.heightfield_grid, "c5": m.scenes.box_hall}[which]()
    for fif, lanes in ((2, 2), (3, 3), (4, 4), (2, 1)):
        dev = m.Device(0, stream=torch.cuda.current_stream().cuda_stream)
        dev.set_queue_lanes(lanes)
        res = m.SceneResources(dev, scene, m.Format.B8G8R8A8_SRGB)
        res.render(); ref = res.read()["color"]
        images = [m.Image(dev, scene.width, scene.height, m.Format.B8G8R8A8_SRGB) for _ in range(fif + 1)]
        out = []
        for vary, thread in ((0, False), (0, True), (7, True)):
            loop = frameloop.FrameLoop(dev, res, images, frames_in_flight=fif, vary_triangles=vary, submit_thread=thread)
            loop.run(max(64, frames // 8))
            t = [loop.run(frames) for _ in range(3)]
            loop.phase_seconds(True)
            loop.run(frames)
            ph = loop.phase_seconds(False)
            if vary == 0:
                print(f"   submit thread {thread}: host us/frame: fence wait {1e6 * ph[0] / frames:.2f}, record {1e6 * ph[1] / frames:.2f}, end {1e6 * ph[2] / frames:.2f}, submit {1e6 * ph[3] / frames:.2f}", flush=True)
            img, n = loop.last_image()
            same = bool(np.array_equal(img.read(), ref)) if vary == 0 else None
            out.append((f"{vary}{' +thread' if thread else ''}", 1e6 * min(t) / frames, same))
            loop.destroy()
        tris = scene.num_triangles
        print(f"{which} frames in flight {fif} on {lanes} lanes: " + "; ".join(
            f"vary {v}: {us:.2f} us/frame = {tris / us:.0f} Mtris/s" + (f" (last frame == resubmitted frame: {same})" if same is not None else "") for v, us, same in out), flush=True)
        for im in images: im.destroy()
        res.destroy(); dev.destroy()

if __name__ == "__main__":
    main()
