#!/usr/bin/env python3
"""Is the C2 line bound by the GPU or by one host thread's submit rate?  T host threads, each with its own Device and L queue
lanes, submit frames for a fixed number of iterations; the aggregate rate over T x L tells (ctypes releases the GIL inside
mirhi_queue_submit).  usage: saturation_probe.py [workload] [frames_per_thread]   (env GPU_MAX_HW_QUEUES is read by the HIP
runtime at start-up: run once without it and once with GPU_MAX_HW_QUEUES=8)"""
import os, sys, threading, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
m = ge.load_package()
workload = sys.argv[1] if len(sys.argv) > 1 else "c2"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
scene = {"c2": m.scenes.random_triangles, "c3": m.scenes.displaced_sphere, "c4": m.scenes.heightfield_grid, "c5": m.scenes.box_hall}[workload]()


def make(lanes):
    dev = m.Device(0)
    dev.set_queue_lanes(lanes)
    shared = {}

    def wrap(device, usage, arr):
        key = (usage, arr.size, arr.ctypes.data)
        if key not in shared:
            shared[key] = m.Buffer.new_with_data(device, usage, arr)
        return shared[key]
    slots = [m.SceneResources(dev, scene, m.Format.B8G8R8A8_SRGB, wrap_buffers=wrap) for _ in range(lanes)]
    return dev, slots


def run(T, L):
    rigs = [make(L) for _ in range(T)]
    host = [0.0] * T

    def work(t, n):
        dev, slots = rigs[t]
        t0 = time.perf_counter()
        for i in range(n):
            slots[i % L].render()
        host[t] = time.perf_counter() - t0
        dev.wait_idle()
    for t in range(T):
        work(t, 4000)
    th = [threading.Thread(target=work, args=(t, frames)) for t in range(T)]
    t0 = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    dt = time.perf_counter() - t0
    out = {"threads": T, "lanes": L, "mtris_per_s": round(scene.num_triangles * frames * T / dt / 1e6, 1), "us_per_frame": round(1e6 * dt / (frames * T), 3),
           "host_loop_us_per_frame": round(1e6 * max(host) / frames, 3)}
    for dev, slots in rigs:
        seen = set()
        for s in slots:
            s.objs = [o for o in s.objs if not (id(o) in seen or seen.add(id(o)))]
            s.destroy()
        dev.destroy()
    return out


print(json.dumps({"workload": workload, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")}))
for T, L in ((1, 2), (1, 3), (1, 4), (1, 4), (1, 6), (1, 8), (1, 8), (2, 4), (1, 4)):
    print(json.dumps(run(T, L)), flush=True)
