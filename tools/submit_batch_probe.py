#!/usr/bin/env python3
"""How much of the C2 frame period is host time?  L queue lanes, B command buffers per mirhi_queue_submit call (one ctypes call,
2 x B kernel launches inside).  usage: submit_batch_probe.py [frames]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
m = ge.load_package()
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
scene = m.scenes.random_triangles()
for L, B in ((4, 1), (4, 2), (4, 4), (8, 8), (4, 1)):
    dev = m.Device(0)
    dev.set_queue_lanes(min(L, 4))
    shared = {}

    def wrap(device, usage, arr):
        key = (usage, arr.size, arr.ctypes.data)
        if key not in shared:
            shared[key] = m.Buffer.new_with_data(device, usage, arr)
        return shared[key]
    slots = [m.SceneResources(dev, scene, m.Format.B8G8R8A8_SRGB, wrap_buffers=wrap) for _ in range(L)]
    groups = [[s.cmd for s in slots[g:g + B]] for g in range(0, L, B)]
    for i in range(2000):
        dev.submit(groups[i % len(groups)])
    dev.wait_idle()
    n = frames // B
    t0 = time.perf_counter()
    for i in range(n):
        dev.submit(groups[i % len(groups)])
    host = time.perf_counter() - t0
    dev.wait_idle()
    dt = time.perf_counter() - t0
    print(json.dumps({"cmd_buffers": L, "per_submit": B, "us_per_frame": round(1e6 * dt / (n * B), 3), "host_us_per_frame": round(1e6 * host / (n * B), 3),
                      "mtris_per_s": round(scene.num_triangles * n * B / dt / 1e6, 1)}), flush=True)
    seen = set()
    for s in slots:
        s.objs = [o for o in s.objs if not (id(o) in seen or seen.add(id(o)))]
        s.destroy()
    dev.destroy()
