import os, sys, ctypes as C, numpy as np
os.environ["MIRHI_LIB_NAME"] = "libmirhi_stamps.so"
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
m = ge.load_package()
L = C.CDLL(m.LIB_PATH)
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
DANCER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "dancer", "scene.gltf")
make = {"dancer": lambda: m.scenes.gltf_model(DANCER), "dancer-tex": lambda: m.scenes.gltf_model(DANCER, textures=True), "tri1": lambda: m.scenes.random_triangles(1), "c2": m.scenes.random_triangles, "c3": m.scenes.displaced_sphere, "c4": m.scenes.heightfield_grid, "c5": m.scenes.box_hall, "grid16": lambda: m.scenes.heightfield_grid(16, 16, 1920, 1080), "grid64": lambda: m.scenes.heightfield_grid(64, 64, 1920, 1080)}[wl]
scene = make()
dev = m.Device(0)
res = m.SceneResources(dev, scene, m.Format.B8G8R8A8_SRGB)
for _ in range(3):
    res.render(); dev.wait_idle()
L.mirhi_debug_clear_stamps()
res.render(); dev.wait_idle()
dev.wait_idle()
def grab(kernel_waves):
    buf = np.zeros(32768 * 8, dtype=np.uint64)
    L.mirhi_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), buf.size)
    return buf.reshape(-1, 8)
# the stamp buffer is shared by both kernels: raster runs last, so it holds raster stamps for wave ids < raster waves
st = grab(0).astype(np.int64)
wpt = 16 if os.environ.get("MIRHI_RASTER_WIDE") == "1" else 4
nw = min(32768, ((scene.width + 31) // 32) * ((scene.height + 31) // 32) * wpt)
s = st[:nw]
ok = s[:, 4] > 0
s = s[ok]
t0 = s[:, 0].min()
busy = s[(s[:, 4] - s[:, 0]) > 2400 * 3]          # waves that lived longer than 3 us: busy tiles
print(f"busy waves {len(busy)}: first start -> last end of busy waves {((busy[:,4].max() - busy[:,0].min()) / 2400.0) if len(busy) else 0:.2f} us; start spread of busy waves p50 {np.median(busy[:,0] - t0) / 2400.0 if len(busy) else 0:.2f} max {((busy[:,0] - t0).max() / 2400.0) if len(busy) else 0:.2f} us")
def us(c): return c / 2400.0   # s_memtime ticks = shader cycles (~2.4 GHz)
print(f"raster waves {len(s)}: kernel span {us(s[:,4].max() - t0):.2f} us (first start -> last end)")
for name, a, b in (("prologue (counters)", 0, 1), ("bin list (fill+raster)", 1, 2), ("  of which fill of first chunk", 1, 5), ("big list", 2, 3), ("resolve", 3, 4), ("whole wave", 0, 4)):
    d = s[:, b] - s[:, a]
    d = d[(s[:, b] > 0) & (s[:, a] > 0)]
    if len(d): print(f"  {name:32s} mean {us(d.mean()):7.2f} us  p50 {us(np.median(d)):7.2f}  max {us(d.max()):7.2f}")
# s_memtime cannot be compared between waves; launch ramp and drain are read on the device-wide 100 MHz clock (slots 6, 7: 10 ns steps)
if (s[:, 6] > 0).all() and (s[:, 7] > 0).all():
    r0 = s[:, 6].min()
    starts = (s[:, 6] - r0) / 100.0; ends = (s[:, 7] - r0) / 100.0
    print(f"  on the device clock: first wave start -> last wave start {starts.max():.2f} us (p50 {np.median(starts):.2f}, p90 {np.percentile(starts, 90):.2f}); first start -> last end {ends.max():.2f} us; "
          f"waves resident at once: max {max(int(((starts <= t) & (ends > t)).sum()) for t in np.linspace(0, ends.max(), 64))}")
print(f"  wave start spread: p50 {us(np.median(s[:,0]-t0)):.2f} us, max {us((s[:,0]-t0).max()):.2f} us")
buf = np.zeros(16384 * 8, dtype=np.uint64)
L.mirhi_debug_read_geo_stamps(buf.ctypes.data_as(C.c_void_p), buf.size)
g = buf.reshape(-1, 8).astype(np.int64)
waves64 = (scene.num_triangles + 63) // 64
tpw = int(os.environ.get("MIRHI_GEOM_TPW", "0")) or (16 if waves64 <= 256 else (32 if waves64 <= 512 else 64))      # (the host's choice: submit_now)
nb = min(16384, (scene.num_triangles + tpw - 1) // tpw)
g = g[:nb]; g = g[g[:, 3] > 0]
print(f"geometry waves sampled {len(g)} ({tpw} triangles per wave): kernel span {us(g[:,3].max() - g[:,0].min()):.2f} us (first start -> last end), wave start spread p50 {us(np.median(g[:,0] - g[:,0].min())):.2f} max {us((g[:,0] - g[:,0].min()).max()):.2f} us")
rt = np.zeros(16384 * 2, dtype=np.uint64)
L.mirhi_debug_read_geo_clock(rt.ctypes.data_as(C.c_void_p), rt.size)
rt = rt.reshape(-1, 2).astype(np.int64)[:nb]; rt = rt[(rt[:, 0] > 0) & (rt[:, 1] > 0)]
if len(rt):
    r0 = rt[:, 0].min()
    print(f"  on the device clock: first wave start -> last wave start {(rt[:, 0].max() - r0) / 100.0:.2f} us (p50 {np.median(rt[:, 0] - r0) / 100.0:.2f}, p90 {np.percentile(rt[:, 0] - r0, 90) / 100.0:.2f}); first start -> last end {(rt[:, 1].max() - r0) / 100.0:.2f} us; wave life mean {(rt[:, 1] - rt[:, 0]).mean() / 100.0:.2f} max {(rt[:, 1] - rt[:, 0]).max() / 100.0:.2f} us")
for name, a, b in (("  entry -> draw descriptor in registers", 0, 4), ("  -> indices + vertices in registers", 4, 5), ("  -> setup done", 5, 6), ("  -> flat colour stored", 6, 1),
                   ("fetch + vs + setup", 0, 1), ("  pairs enumerated, reservations issued", 1, 7), ("  -> records stored (issued)", 7, 2), ("binning (atomics + record copies)", 1, 2), ("clip", 2, 3), ("whole wave", 0, 3)):
    d = g[:, b] - g[:, a]
    print(f"  {name:36s} mean {us(d.mean()):7.2f} us  p50 {us(np.median(d)):7.2f}  max {us(d.max()):7.2f}")
res.destroy(); dev.destroy()
