import sys, time, numpy as np
import __graft_entry__ as ge
m = ge.load_package()
S = m.scenes
scene = S.random_triangles()
for lanes in (3, 4):
    for batch in (1, 4, 16, 64):
        dev = m.Device(0); dev.set_queue_lanes(lanes)
        frames = [m.SceneResources(dev, scene, m.Format.B8G8R8A8_SRGB) for _ in range(lanes)]
        cmds = [frames[i % lanes].cmd for i in range(batch)]
        for _ in range(5): dev.submit(cmds)
        dev.wait_idle()
        N = 1920 // batch
        t0 = time.perf_counter()
        for _ in range(N): dev.submit(cmds)
        t_host = time.perf_counter() - t0
        dev.wait_idle()
        t1 = time.perf_counter() - t0
        print(f"lanes {lanes} batch {batch}: host submit {1e6*t_host/(N*batch):.2f} us/frame, total {1e6*t1/(N*batch):.2f} us/frame = {scene.num_triangles*N*batch/t1/1e6:.0f} Mtris/s", flush=True)
        for f in frames: f.destroy()
        dev.destroy()
