import sys, time, numpy as np
import __graft_entry__ as ge
m = ge.load_package()
S = m.scenes
def run(scene, lanes, nframes_in_flight, N=400, fmt=m.Format.B8G8R8A8_SRGB):
    dev = m.Device(0)
    dev.set_queue_lanes(lanes)
    frames = [m.SceneResources(dev, scene, fmt) for _ in range(nframes_in_flight)]
    for i in range(10): frames[i % len(frames)].render()
    dev.wait_idle()
    t0 = time.perf_counter()
    for i in range(N): frames[i % len(frames)].render()
    dev.wait_idle()
    t1 = time.perf_counter()
    print(f"{scene.name}: lanes {lanes} frames-in-flight {nframes_in_flight}: {1e6*(t1-t0)/N:.1f} us/frame  {scene.num_triangles*N/(t1-t0)/1e6:.1f} Mtris/s", flush=True)
    for f in frames: f.destroy()
    dev.destroy()
for scene in (S.random_triangles(), S.displaced_sphere(), S.heightfield_grid()):
    run(scene, 1, 1); run(scene, 1, 2); run(scene, 2, 2); run(scene, 3, 3); run(scene, 4, 4)
