#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X compute rasterizer (contract in the task statement).

A "step" is one frame: one pass of the hot path (vertex transform -> setup/binning -> tile raster
-> shading -> final colour store) over one synthetic scene whose inputs are already resident in HBM.
Default workload = BASELINE.json configs[1]: 10k random flat-shaded triangles at 1920x1080,
B8G8R8A8_SRGB target (the reference's swapchain format).  Metric: Mtris/s (input triangles / s).

N > 1 (torchrun, one rank per GPU):
  default        every rank renders whole frames (alternate-frame rendering, no data-path
                 collective): weak scaling, value = N * K * tris / max-over-ranks time
  --split rows   one frame is split by screen-tile rows across the ranks and the bands are
                 all-gathered over RCCL/xGMI every frame: strong scaling of a single frame
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)

WORKLOADS = {
    "c2": ("BASELINE configs[1]: 10k random triangles, flat shade, 1920x1080", lambda s: s.random_triangles()),
    "c3": ("BASELINE configs[2] stand-in: 70,312-tri displaced sphere, Phong + 1 point light, 1920x1080", lambda s: s.displaced_sphere()),
    "c4": ("BASELINE configs[3]: 1M-triangle grid, 3840x2160", lambda s: s.heightfield_grid()),
    "c5": ("BASELINE configs[4] stand-in: 262,144-tri box hall, 4 lights + textures, 3840x2160", lambda s: s.box_hall()),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--split", default="none", choices=["none", "rows"])
    ap.add_argument("--format", default="bgra8", choices=["bgra8", "rgba32f"])
    ap.add_argument("--frames-in-flight", type=int, default=4,
                    help="independent frames (own command buffer + target) overlapped on the GPU; the reference keeps 2 "
                         "(MAX_FRAMES_IN_FLIGHT) + 1 swapchain image")
    ap.add_argument("--profile-pass-only", action="store_true",
                    help="only the isolated per-kernel timing pass (one frame at a time): the command profiled with rocprofv3")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo only for single-box rehearsals")
    ap.add_argument("--no-split-extra", action="store_true", help="N>1: skip the secondary tile-row-split measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def cpu_baseline(scene, seconds: float):
    """The oracle (a C port of the same pipeline; the reference has no CPU path and cannot be built
    here) timed on this host's cores on whole frames of the same workload for ~`seconds`."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    cores = max(1, min(os.cpu_count() or 1, 64))
    ob.render(scene, nthreads=cores, want_bgra8=True)       # warm-up, page-in
    t0 = time.perf_counter()
    frames = 0
    while True:
        ob.render(scene, nthreads=cores, want_bgra8=True)
        frames += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or frames >= 2000:
            break
    return {"value": round(scene.num_triangles * frames / dt / 1e6, 4), "unit": "Mtris/s", "cores": cores, "kind": "port",
            "sample": f"{frames} whole frames of the same workload in {dt:.1f} s (oracle/mirhi_oracle.c, {cores} row-band threads)"}


def tile_split_measurement(m, multigpu, torch, dist, dev0, rank, world, local_rank, args, barrier):
    """SURVEY 8e / BASELINE configs[3]: ONE frame (1M triangles, 3840x2160) split by screen-tile rows across the ranks,
    the bands all-gathered over RCCL/xGMI every frame.  Strong scaling of a single frame; reported beside the primary."""
    import time as _t
    scene = m.scenes.heightfield_grid()
    dev = m.Device(local_rank, stream=torch.cuda.current_stream().cuda_stream)
    dev.set_tile_split(rank, world)
    keep = []

    def wrap(device, usage, arr):
        t = torch.from_numpy(arr.copy()).cuda()
        keep.append(t)
        return m.Buffer.wrap(device, usage, t.data_ptr(), t.numel())

    frame = torch.zeros((multigpu.padded_rows(scene.height, world), scene.width, 4), dtype=torch.uint8, device="cuda")
    target = m.Image(dev, scene.width, scene.height, m.Format.B8G8R8A8_SRGB, device_ptr=frame.data_ptr())
    res = m.SceneResources(dev, scene, m.Format.B8G8R8A8_SRGB, color_image=target, wrap_buffers=wrap)
    steps = max(10, min(200, args.steps))

    def step():
        res.render()
        multigpu.all_gather_bands(frame, rank, world, via_host=(args.backend == "gloo"))

    for _ in range(5):
        step()
    barrier()
    t0 = _t.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    dt = _t.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    res.destroy()
    dev.destroy()
    return {"workload": "c4: 1M-triangle grid, 3840x2160, tile-row split + in-place all-gather of BGRA8 bands",
            "value": round(scene.num_triangles * steps / dt / 1e6, 3), "unit": "Mtris/s", "ms_per_frame": round(1e3 * dt / steps, 4),
            "steps": steps, "scaling": "strong", "n_gpus": world}


def measured_traffic(workload: str):
    """HBM bytes per raster_kernel launch from the latest committed rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE collected
    in separate runs and corrected as MI355X_MICROARCH.md prescribes; see profiles/README.md).  Counters cannot be read
    from inside this process, so the figure is the one measured for the same kernel build and workload, or null."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_" + workload + "_hbm_traffic.json"))):
        try:
            best = json.load(open(path))["raster_kernel"]["hbm_bytes_per_launch"]
        except Exception:
            pass
    return best


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    m = ge.load_package()
    from renderer_rs_amd import multigpu

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X: there is no CPU fallback for the measured path")
    if "MIRHI_BENCH_FORCE_DEVICE" in os.environ:          # rehearsal of N > 1 on a one-GPU box (gloo)
        local_rank = int(os.environ["MIRHI_BENCH_FORCE_DEVICE"])
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    desc, make = WORKLOADS[args.workload]
    scene = make(m.scenes)
    split = args.split == "rows" and world > 1
    fmt = m.Format.B8G8R8A8_SRGB if args.format == "bgra8" else m.Format.R32G32B32A32_SFLOAT
    bpp = 4 if args.format == "bgra8" else 16

    stream = torch.cuda.current_stream().cuda_stream
    dev = m.Device(local_rank, stream=stream)
    nfif = 1 if split else max(1, min(8, args.frames_in_flight))
    dev.set_queue_lanes(nfif)
    if split:
        dev.set_tile_split(rank, world)

    # inputs and the render target live in HBM as torch tensors; the rasterizer wraps the pointers
    keep = []

    def wrap(device, usage, arr):
        t = torch.from_numpy(arr.copy()).cuda()
        keep.append(t)
        return m.Buffer.wrap(device, usage, t.data_ptr(), t.numel())

    rows = multigpu.padded_rows(scene.height, world) if split else scene.height
    frames, slots = [], []
    shared = {}

    def wrap_shared(device, usage, arr):     # geometry / uniforms are uploaded once and shared by all frames in flight
        key = (usage, arr.ctypes.data, arr.size)
        if key not in shared:
            shared[key] = wrap(device, usage, arr)
        return shared[key]

    for _ in range(nfif):                    # one colour target + command buffer per frame in flight (swapchain images)
        frame = torch.zeros((rows, scene.width, 4), dtype=torch.uint8 if bpp == 4 else torch.float32, device="cuda")
        target = m.Image(dev, scene.width, scene.height, fmt, device_ptr=frame.data_ptr())
        frames.append(frame)
        slots.append(m.SceneResources(dev, scene, fmt, color_image=target, wrap_buffers=wrap_shared))
    res = slots[0]
    counter = [0]

    def step():
        i = counter[0] % nfif
        counter[0] += 1
        slots[i].render()
        if split:
            multigpu.all_gather_bands(frames[i], rank, world, via_host=(args.backend == "gloo"))

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(0 if args.profile_pass_only else args.steps):
        step()
    barrier()
    dt = max(time.perf_counter() - t0, 1e-9)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # per-kernel device time: HIP event pairs recorded on the submit stream around each kernel, K more steps with
    # ONE frame in flight (frames that overlap on the GPU would stretch each other's kernel durations)
    dev.set_profiling(True)
    dev.reset_kernel_times()
    for i in range(args.steps):
        slots[0].render()
    torch.cuda.synchronize()
    geo_ms, geo_n = dev.kernel_time(m.Kernel.GEOMETRY)
    ras_ms, ras_n = dev.kernel_time(m.Kernel.RASTER)
    ov_ms, ov_n = dev.event_overhead()       # already subtracted per launch from the two figures above
    dev.set_profiling(False)

    split_extra = None
    if world > 1 and not split and not args.no_split_extra and not args.profile_pass_only:
        try:
            split_extra = tile_split_measurement(m, multigpu, torch, dist, dev, rank, world, local_rank, args, barrier)
        except Exception as e:          # never let the secondary measurement take the primary line down
            split_extra = {"error": repr(e)}

    frames_total = args.steps * (1 if split else world)
    tris = scene.num_triangles
    value = tris * frames_total / dt / 1e6
    stats = dev.stats()

    if rank == 0:
        alg_bytes = scene.algorithmic_bytes(bpp_out=bpp)
        if split:   # all geometry + this rank's share of the frame buffer
            alg_bytes = alg_bytes - scene.width * scene.height * bpp + scene.width * scene.height * bpp // world
        ras_us = 1e3 * ras_ms / max(1, ras_n)
        geo_us = 1e3 * geo_ms / max(1, geo_n)
        achieved = alg_bytes / (ras_us * 1e-6) / 1e9 if ras_us > 0 else 0.0
        out = {
            "metric": "Mtris/s at 1920x1080 (input triangles per second, whole frame incl. shading + store)"
                      if scene.height == 1080 else "Mtris/s (input triangles per second, whole frame incl. shading + store)",
            "value": round(value, 3), "unit": "Mtris/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 6), "higher_is_better": True,
            "scaling": "strong" if split else "weak", "vs_baseline": None, "dtype": "f32+i32",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {desc}", "triangles": tris, "width": scene.width, "height": scene.height,
                       "target_format": "B8G8R8A8_SRGB" if bpp == 4 else "R32G32B32A32_SFLOAT",
                       "parallelism": (f"tile-row split x{world} + RCCL all-gather" if split else (f"afr{world}" if world > 1 else "single")),
                       "frames_per_step": 1, "frames_in_flight": nfif},
            "shaded_mpix_per_s": round(scene.width * scene.height * frames_total / dt / 1e6, 1),
            "roofline": {"bound": "hbm", "kernel": "raster_kernel", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None if split else measured_traffic(args.workload),
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_kernel_us": round(ras_us, 3),
                         "geometry_kernel_us": round(geo_us, 3),
                         "event_pair_overhead_us": round(1e3 * ov_ms, 3), "event_pair_samples": ov_n,
                         "how": "hipEvent pairs on the submit stream around every launch, minus the mean of one EMPTY pair recorded behind each frame's raster pair; K extra steps after the timed region, one frame in flight so kernels of different frames do not overlap"},
            "workspace_mb": round(stats.workspace_bytes / 1e6, 1), "big_list": stats.last_big_list,
        }
        if split_extra is not None:
            out["tile_split"] = split_extra
        if not args.no_cpu_baseline and world == 1 and not args.profile_pass_only:
            try:
                out["cpu_baseline"] = cpu_baseline(scene, args.cpu_seconds)
            except Exception as e:  # the oracle is a reported baseline, never the measured path
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)

    seen = set()
    for sl in slots:                          # shared buffers are destroyed once
        sl.objs = [o for o in sl.objs if not (id(o) in seen or seen.add(id(o)))]
        sl.destroy()
    dev.destroy()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
