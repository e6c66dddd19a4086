"""Host cost of one mirhi_queue_submit (2 launches): a frame so small that the GPU is never the bottleneck.
usage: host_submit_cost.py [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
m = ge.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
dev = m.Device(0)
for lanes in (1, 4):
    dev.set_queue_lanes(lanes)
    res = [m.SceneResources(dev, m.scenes.hello_triangle(64, 64), m.Format.B8G8R8A8_SRGB) for _ in range(lanes)]
    for i in range(200): res[i % lanes].render()
    dev.wait_idle()
    t0 = time.perf_counter()
    for i in range(n): res[i % lanes].render()
    t1 = time.perf_counter()
    dev.wait_idle()
    t2 = time.perf_counter()
    print(f"lanes {lanes}: host {1e6 * (t1 - t0) / n:.2f} us per submit (enqueue only), {1e6 * (t2 - t0) / n:.2f} us per frame incl. drain")
    for r in res: r.destroy()
dev.destroy()

# where the host time goes: a trivial ctypes call, the raw submit call with a prebuilt argument array, the wrapper
import ctypes as C
L = m.lib()
dev = m.Device(0)
dev.set_queue_lanes(4)
res = [m.SceneResources(dev, m.scenes.hello_triangle(64, 64), m.Format.B8G8R8A8_SRGB) for _ in range(4)]
t0 = time.perf_counter()
for i in range(n): L.mirhi_result_name(0)
t1 = time.perf_counter()
print(f"trivial ctypes call: {1e6 * (t1 - t0) / n:.2f} us")
arrs = [(C.c_void_p * 1)(r.cmd.handle) for r in res]
f = L.mirhi_queue_submit; h = dev.handle
for i in range(200): f(h, 1, arrs[i % 4], None)
dev.wait_idle()
t0 = time.perf_counter()
for i in range(n): f(h, 1, arrs[i % 4], None)
t1 = time.perf_counter()
dev.wait_idle()
print(f"raw mirhi_queue_submit with a prebuilt array: {1e6 * (t1 - t0) / n:.2f} us")
for r in res: r.destroy()
dev.destroy()
