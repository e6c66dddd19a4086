#!/usr/bin/env python3
"""What a frame's kernels look like INSIDE the 4-lane frame loop: only lane 0's dispatches carry event pairs (MIRHI_PROFILE_ONE_LANE),
lanes 1-3 run untimed.  Prints the mean duration of lane 0's geometry and raster kernels, the gap between them and the lane's
frame period -- beside the same figures with one lane alone.  usage: one_lane_timeline.py [workload] [frames]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
m = ge.load_package()
workload = sys.argv[1] if len(sys.argv) > 1 else "c2"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
scene = {"c2": m.scenes.random_triangles, "c3": m.scenes.displaced_sphere}[workload]()
for L in (1, 2, 4):
    dev = m.Device(0)
    dev.set_queue_lanes(L)
    shared = {}

    def wrap(device, usage, arr):
        key = (usage, arr.size, arr.ctypes.data)
        if key not in shared:
            shared[key] = m.Buffer.new_with_data(device, usage, arr)
        return shared[key]
    slots = [m.SceneResources(dev, scene, m.Format.B8G8R8A8_SRGB, wrap_buffers=wrap) for _ in range(L)]
    for i in range(2000):
        slots[i % L].render()
    dev.wait_idle()
    dev.reset_kernel_times()
    dev.set_profiling(m.Profile.TIMING | (1 << 8))          # lane 0 only
    for i in range(frames):
        slots[i % L].render()
    dev.wait_idle()
    dev.set_profiling(0)
    tl = dev.timeline()
    geo = [(b, e) for k, lane, b, e in tl if k == m.Kernel.GEOMETRY]
    ras = [(b, e) for k, lane, b, e in tl if k == m.Kernel.RASTER]
    n = min(len(geo), len(ras))
    geo, ras = geo[n // 4:n], ras[n // 4:n]
    out = {"lanes": L, "timed_frames": len(ras),
           "geometry_us": round(sum(e - b for b, e in geo) / len(geo), 3), "raster_us": round(sum(e - b for b, e in ras) / len(ras), 3),
           "geometry_to_raster_gap_us": round(sum(r[0] - g[1] for g, r in zip(geo, ras)) / len(ras), 3),
           "raster_to_next_geometry_gap_us": round(sum(g2[0] - r[1] for r, g2 in zip(ras, geo[1:])) / (len(ras) - 1), 3),
           "lane_frame_period_us": round((ras[-1][1] - ras[0][1]) / (len(ras) - 1), 3)}
    out["frame_period_all_lanes_us"] = round(out["lane_frame_period_us"] / L, 3)
    print(json.dumps(out), flush=True)
    seen = set()
    for s in slots:
        s.objs = [o for o in s.objs if not (id(o) in seen or seen.add(id(o)))]
        s.destroy()
    dev.destroy()
