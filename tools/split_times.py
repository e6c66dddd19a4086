"""Per-rank kernel times of a screen-tile-row split, EMULATED ON ONE GPU: every rank of world 2 / 4 / 8 renders its band in turn (isolated
dispatches, event pair on each), and the 1 -> N curve is PREDICTED from them: a frame's time on N GPUs = the slowest rank's kernels, or the
band exchange if that is longer (direct sends over a full xGMI mesh: every link carries one band, frame_bytes / N at 153 GB/s; the exchange
of frame f overlaps the kernels of frame f + 1 with two frames in flight).  At world 8 also each rank's frame PERIOD in the reference-shaped loop (re-recorded, fenced, 2 and 4
frames in flight): what a rank delivers when its geometry of the next frame runs under its raster kernel of this one.  Nothing here has run on more than one GPU.
usage: split_times.py [--json out.json] [--layout bands|interleaved|both] [workloads...]      (default: both layouts, c4 c5)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
m = ge.load_package()
args = sys.argv[1:]
out_path = None
if "--json" in args:
    i = args.index("--json"); out_path = args[i + 1]; del args[i:i + 2]
layouts = ["bands", "interleaved"]
if "--layout" in args:
    i = args.index("--layout"); layouts = layouts if args[i + 1] == "both" else [args[i + 1]]; del args[i:i + 2]
LINK_GBS = 153.0
result = {"note": "emulated on one MI355X: per-rank kernel times are measured (isolated dispatches), the curve is predicted from them", "link_GB_per_s": LINK_GBS, "workloads": {}}
for wl in args or ["c4", "c5"]:
    scene = {"c2": m.scenes.random_triangles, "c3": m.scenes.displaced_sphere, "c4": m.scenes.heightfield_grid, "c5": m.scenes.box_hall}[wl]()
    frame_bytes = scene.width * scene.height * 4
    table = {}

    def measure(rank, world, layout="bands"):
        dev = m.Device(0)
        if world > 1:
            dev.set_tile_split(rank, world, layout=layout)
        res = m.SceneResources(dev, scene, m.Format.B8G8R8A8_SRGB)
        for _ in range(6):
            res.render()
        dev.wait_idle()
        dev.set_profiling(m.Profile.TIMING); dev.reset_kernel_times()
        for _ in range(24):
            res.render(); dev.wait_idle()
        t = {name: dev.kernel_time(k) for k, name in enumerate(m.Kernel.NAMES)}
        res.destroy(); dev.destroy()
        return {k: round(1e3 * ms / n, 2) if n else 0.0 for k, (ms, n) in t.items() if k != "fragment_count"}

    def period(rank, world, layout="bands", fif=2, frames=300):
        """The rank's frame period in the reference-shaped loop (re-recorded, fenced, `fif` frames in flight on as many lanes): its vertex / geometry kernels of frame
        f + 1 run under the raster kernel of frame f, so a rank delivers a frame in less than the sum of its isolated kernels."""
        from renderer_rs_amd import frameloop
        dev = m.Device(0)
        dev.set_queue_lanes(fif)
        if world > 1:
            dev.set_tile_split(rank, world, layout=layout)
        res = m.SceneResources(dev, scene, m.Format.B8G8R8A8_SRGB)
        res.render()
        images = [m.Image(dev, scene.width, scene.height, m.Format.B8G8R8A8_SRGB) for _ in range(fif + 1)]
        loop = frameloop.FrameLoop(dev, res, images, frames_in_flight=fif)
        loop.run(max(16, frames // 8))
        t = min(loop.run(frames) for _ in range(2))
        loop.destroy()
        for im in images: im.destroy()
        res.destroy(); dev.destroy()
        return round(1e6 * t / frames, 2)

    one = measure(0, 1)
    t1 = sum(one.values())
    p1 = {f: period(0, 1, fif=f) for f in (2, 4)}
    table["1"] = {"ranks": [one], "frame_us": round(t1, 2), "speedup": 1.0, "period_us_2_in_flight": p1[2], "period_us_4_in_flight": p1[4]}
    print(f"{wl} world 1: {one} = {t1:.1f} us; frame period in the fenced loop {p1[2]} us (2 in flight), {p1[4]} us (4)", flush=True)
    for layout, world in [(lay, w) for lay in layouts for w in (2, 4, 8)]:
        ranks = [measure(r, world, layout) for r in range(world)]
        slow = max(sum(r.values()) for r in ranks)
        exch = 1e6 * (frame_bytes / world) / (LINK_GBS * 1e9)
        frame = max(slow, exch)
        per2 = [period(r, world, layout, fif=2) for r in range(world)] if world == 8 else None      # (world 8 only: the size the BASELINE configs name)
        per4 = [period(r, world, layout, fif=4) for r in range(world)] if world == 8 else None
        fixed = min(r["geometry"] for r in ranks) / one["geometry"] if one["geometry"] else 0.0
        table[f"{world} {layout}"] = {"layout": layout, "ranks": ranks, "slowest_rank_kernels_us": round(slow, 2), "exchange_us_modelled": round(exch, 2), "frame_us_predicted": round(frame, 2),
                             "speedup_predicted": round(t1 / frame, 2), "geometry_share_of_cheapest_rank": round(fixed, 3)}
        if per2:
            table[f"{world} {layout}"].update({"rank_period_us_2_in_flight": per2, "rank_period_us_4_in_flight": per4,
                                               "speedup_predicted_2_in_flight": round(p1[2] / max(max(per2), exch), 2), "speedup_predicted_4_in_flight": round(p1[4] / max(max(per4), exch), 2)})
            print(f"{wl} world {world} {layout}: a rank's frame period in the fenced loop: 2 in flight {min(per2):.1f}-{max(per2):.1f} us, 4 in flight {min(per4):.1f}-{max(per4):.1f} us "
                  f"-> predicted x{p1[2] / max(max(per2), exch):.2f} / x{p1[4] / max(max(per4), exch):.2f} against one GPU's {p1[2]} / {p1[4]} us", flush=True)
        print(f"{wl} world {world} {layout}: slowest rank {slow:.1f} us (vertex / geometry / raster per rank: " +
              " | ".join(f"{r['vertex']:.1f}/{r['geometry']:.1f}/{r['raster']:.1f}" for r in ranks) + f"), exchange {exch:.1f} us -> predicted x{t1 / frame:.2f}", flush=True)
    result["workloads"][wl] = {"triangles": scene.num_triangles, "width": scene.width, "height": scene.height, "worlds": table}
if out_path:
    json.dump(result, open(out_path, "w"), indent=1)
