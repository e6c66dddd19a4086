"""PCIe-inclusive rate of the C2 workload: the figure DESIGN.md quotes next to bench.py's resident-input `value`.
Three loops over the same command buffer (BASELINE configs[1], 1920x1080 B8G8R8A8_SRGB), one frame at a time:
  resident      submit + wait                                   (inputs and target stay in HBM)
  upload        vertex buffer written from host memory (mirhi_buffer_write, 0.72 MB H2D) + submit + wait
  upload+read   as above + the finished frame copied back to host memory (mirhi_image_read, 8.3 MB D2H)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
m = ge.load_package()
scene = m.scenes.random_triangles()
dev = m.Device(0)
res = m.SceneResources(dev, scene, m.Format.B8G8R8A8_SRGB)
vb = res.draw_state[0]["vb"]
host = np.ascontiguousarray(np.frombuffer(np.asarray(scene.draws[0].vertices).tobytes(), dtype=np.uint8))
fence = m.Fence(dev)
def loop(n, upload, read):
    t0 = time.perf_counter()
    for _ in range(n):
        if upload: vb.write_data(0, host)
        fence.reset(); res.render(fence); fence.wait()
        if read: res.color.read()
    return (time.perf_counter() - t0) / n
for name, up, rd in (("resident", False, False), ("upload", True, False), ("upload+read", True, True)):
    loop(20, up, rd)
    dt = loop(300, up, rd)
    print(f"{name:12s} {1e6 * dt:8.1f} us/frame  {scene.num_triangles / dt / 1e6:8.1f} Mtris/s")
fence.destroy(); res.destroy(); dev.destroy()
