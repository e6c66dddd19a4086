#!/usr/bin/env python3
"""Reference-shaped frame loop driven from Python (wait fence -> reset -> re-record -> end -> submit with fence), C2 scene.
Python's own per-call cost is part of the figure: the native loop is host/frame_loop.cpp (bench.py: rerecorded_submit)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
m = ge.load_package()
import torch

def main():
    scene = m.scenes.random_triangles()
    for fif in (2, 4):
        dev = m.Device(0, stream=torch.cuda.current_stream().cuda_stream)
        dev.set_queue_lanes(fif)
        slots, fences = [], []
        shared = {}
        for i in range(fif):
            sl = m.SceneResources(dev, scene, m.Format.B8G8R8A8_SRGB)
            sl.cmd.set_queue_lane(i)
            slots.append(sl)
            fences.append(m.Fence(dev, signaled=True))
        def loop(n, record):
            t0 = time.perf_counter()
            for i in range(n):
                k = i % fif
                fences[k].wait(); fences[k].reset()
                if record:
                    slots[k].record()
                slots[k].render(fences[k])
            dev.wait_idle()
            return 1e6 * (time.perf_counter() - t0) / n
        loop(200, True)
        print(f"frames in flight {fif}: re-recorded {loop(2000, True):.2f} us/frame, resubmitted with fences {loop(2000, False):.2f} us/frame", flush=True)
        for sl in slots: sl.destroy()
        for f in fences: f.destroy()
        dev.destroy()

if __name__ == "__main__":
    main()
