#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X compute rasterizer (contract in the task statement).

A "step" is one BATCH of `config.frames_per_step` frames; a frame is one pass of the hot path (vertex transform ->
setup / binning -> tile raster -> shading -> final colour store) over one synthetic scene whose inputs are already
resident in HBM.  value = triangles x frames_per_step x steps / time.  Batching makes the timed region tens of
milliseconds whatever --steps says (the driver runs --steps 20 --warmup 5), so the figure is the steady state with
`frames_in_flight` command buffers in flight and not the ramp-up of a few frames.

N = 1   BASELINE.json configs[1]: 10k random flat-shaded triangles at 1920x1080, B8G8R8A8_SRGB target (the reference's
        swapchain format).  Metric: Mtris/s (input triangles / s); shaded Mpix/s and overdraw beside it.
N > 1   one rank per GPU.  `--gpus N` starts the N ranks itself (torch.distributed.run as a CHILD process, before
        anything touches the GPU) unless a launcher already did (WORLD_SIZE set).  Primary measurement: BASELINE
        configs[3] -- ONE 1M-triangle 3840x2160 frame split by screen-tile rows across the ranks, the finished BGRA8
        bands exchanged over RCCL / xGMI through the C ABI (mirhi_comm_all_gather_bands): strong scaling of a frame.
        `--split none` measures alternate-frame rendering instead (whole frames per rank, no collective, weak scaling).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)
SIMDS, CLOCK_GHZ = 1024, 2.4     # 256 CUs x 4 SIMDs; shader clock of the issue bound below
# cycles a SIMD needs per wave64 instruction, measured on this part (tools/microbench/valu_rate*.hip, salu_rate.hip): VALU 2.5 (add / mul / fma / logic) to 4.3
# (v_mad_i32_i24, compares, selects, shifts, conversions); SALU 4.3; a VALU and a SALU instruction of different waves issue side by side
VALU_CYCLES, SALU_CYCLES = (2.5, 4.3), 4.3


def issue_bound(insts, kernel_us):
    """roofline.issue_*: the kernel's instruction-issue bound from its measured instruction mix (committed SQ counters of the same build) --
    max(VALU x cycles per VALU, SALU x cycles per SALU) / (SIMDs x clock), for the cheapest and the dearest VALU mix -- and the fraction of it the
    kernel's measured duration reaches.  SURVEY 8(d): 'report VALU utilisation too: the byte floor is tiny'."""
    if not insts or not kernel_us:
        return None
    valu, salu = insts.get("valu", 0), insts.get("salu", 0)
    lo = max(valu * VALU_CYCLES[0], salu * SALU_CYCLES) / (SIMDS * CLOCK_GHZ * 1e3)
    hi = max(valu * VALU_CYCLES[1], salu * SALU_CYCLES) / (SIMDS * CLOCK_GHZ * 1e3)
    return {"valu_insts": valu, "salu_insts": salu, "issue_bound_us": [round(lo, 2), round(hi, 2)], "issue_frac": [round(lo / kernel_us, 3), round(hi / kernel_us, 3)],
            "how": "wave-instructions per launch from rocprofv3 --pmc SQ_INSTS_VALU / SQ_INSTS_SALU of this build (profiles/*_hbm_traffic.json); bound = max(VALU x 2.5..4.3, SALU x 4.3) cycles "
                   "/ (1024 SIMDs x 2.4 GHz); frac = bound / the kernel's isolated duration"}

WORKLOADS = {
    # name: (description, scene factory, default frames per step)
    "c2": ("BASELINE configs[1]: 10k random triangles, flat shade, 1920x1080", lambda s: s.random_triangles(), 512),
    "c3": ("BASELINE configs[2] stand-in: 70,312-tri displaced sphere, Phong + 1 point light, 1920x1080", lambda s: s.displaced_sphere(), 256),
    "c4": ("BASELINE configs[3]: 1M-triangle grid, 3840x2160", lambda s: s.heightfield_grid(), 32),
    "c5": ("BASELINE configs[4] stand-in: 262,144-tri box hall, 4 lights + textures, 3840x2160", lambda s: s.box_hall(), 32),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="ranks = GPUs; > 1 without a launcher: bench.py starts them itself")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS), help="default: c2 on one GPU, c4 on several")
    ap.add_argument("--frames-per-step", type=int, default=0, help="frames in one step (0 = the workload's default)")
    ap.add_argument("--split", default=None, choices=["none", "rows"], help="N > 1: rows (default) = tile-row split of one frame + "
                    "RCCL band exchange; none = alternate-frame rendering")
    ap.add_argument("--split-layout", default="interleaved", choices=["bands", "interleaved"],
                    help="N > 1, rows: which tile rows a rank rasterizes -- one contiguous band, or rows rank, rank + N, ... (load balance: SURVEY 8e); the "
                         "torch.distributed fallback of the exchange needs bands")
    ap.add_argument("--gather", default="abi", choices=["abi", "torch"], help="band exchange: mirhi_comm_* (RCCL through the C ABI) or "
                    "torch.distributed.all_gather_into_tensor")
    ap.add_argument("--gather-algo", default="direct", choices=["direct", "broadcast"])
    ap.add_argument("--format", default="bgra8", choices=["bgra8", "rgba32f"])
    ap.add_argument("--frames-in-flight", type=int, default=4,
                    help="independent frames (own command buffer + target) overlapped on the GPU; the reference keeps 2 "
                         "(MAX_FRAMES_IN_FLIGHT) + 1 swapchain image")
    ap.add_argument("--frames-per-submit", type=int, default=1,
                    help="command buffers handed to one mirhi_queue_submit call (vkQueueSubmit with several command buffers): frames of "
                         "equal shape then share one batch of kernel launches; frames in flight = this x --frames-in-flight")
    ap.add_argument("--record-each-frame", action="store_true",
                    help="(the default on one GPU since round 4) the timed region is the reference-shaped loop: every frame waits on its in-flight fence, "
                         "resets and RE-RECORDS its command buffer, ends and submits it with the fence (renderer.rs:367-449,452-557), natively through "
                         "libmirhost.so; the resubmitted rate is measured beside it (resubmitted_submit)")
    ap.add_argument("--resubmit", action="store_true",
                    help="the timed region resubmits command buffers recorded once (the headline of rounds 1-3) instead of re-recording every frame")
    ap.add_argument("--other-workloads", default="c3,c4,c5", help="N = 1: short passes of these workloads after the headline (workloads{} in the line); '' = none")
    ap.add_argument("--profile-pass-only", action="store_true",
                    help="skip the timed region: only the per-dispatch timing passes (the command profiled with rocprofv3 --pmc)")
    ap.add_argument("--timeline-out", default=None, help="write the per-dispatch timeline of the in-flight pass to this JSON file")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo only for rehearsals of N > 1 on a one-GPU box")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (N = 1: batched submits; N > 1: one-GPU reference)")
    ap.add_argument("--prewarm-seconds", type=float, default=1.0, help="untimed frame loop in front of the warm-up steps (GPU clock ramp)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=16.0)
    return ap.parse_args()


def spawn_ranks(args) -> int:
    """--gpus N without a launcher: N fresh rank processes under torch.distributed.run, started as a child BEFORE this process
    has made any HIP call (never an exec of a process that has touched the GPU).  Returns the child's exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def cpu_baseline(scene, seconds: float):
    """The oracle (a C port of the same pipeline; the reference has no CPU path and cannot be built here) timed on this host's
    cores on whole frames of the same workload: all cores (row-band threads) for half of `seconds`, one thread for the rest."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob

    def run(threads, budget):
        ob.render(scene, nthreads=threads, want_bgra8=True)       # warm-up, page-in
        t0 = time.perf_counter()
        frames = 0
        while True:
            ob.render(scene, nthreads=threads, want_bgra8=True)
            frames += 1
            dt = time.perf_counter() - t0
            if dt >= budget or frames >= 2000:
                break
        return round(scene.num_triangles * frames / dt / 1e6, 4), frames, dt

    cores = max(1, min(os.cpu_count() or 1, 64))
    v_all, f_all, t_all = run(cores, seconds / 2)
    v_one, f_one, t_one = run(1, seconds / 2)
    return {"value": v_all, "unit": "Mtris/s", "cores": cores, "kind": "port",
            "sample": f"{f_all} whole frames of the same workload in {t_all:.1f} s (oracle/mirhi_oracle.c, {cores} row-band threads)",
            "single_thread": {"value": v_one, "unit": "Mtris/s", "cores": 1, "sample": f"{f_one} whole frames in {t_one:.1f} s"}}


def measured_traffic(workload: str, source_hash: str):
    """HBM bytes per FRAME (every kernel of the frame: vertex + geometry + raster) from the committed rocprofv3 PMC passes
    (FETCH_SIZE and WRITE_SIZE in separate runs, corrected as MI355X_MICROARCH.md prescribes; profiles/README.md).  Counters
    cannot be read from inside this process, so the figure is quoted only if it was taken on THIS kernel build (source hash)."""
    import glob
    best, note = None, "no counters committed for this workload"
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_" + workload + "_hbm_traffic.json"))):
        try:
            j = json.load(open(path))
        except Exception:
            continue
        if j.get("kernel_source_sha16") == source_hash and "frame_hbm_bytes" in j:
            best, note = j, os.path.basename(path)
        elif best is None:
            note = f"{os.path.basename(path)} was measured on build {j.get('kernel_source_sha16', '?')}, this is {source_hash}: not quoted"
    return best, note


def summarize_timeline(tl, kernel_names):
    """tl: [(kernel, lane, begin_us, end_us)] in submission order.  Overlap between consecutive raster kernels (by begin time),
    the gap between a frame's geometry and raster kernel, and the steady-state frame period."""
    if not tl:
        return None
    by_kernel = {}
    for k, lane, b, e in tl:
        by_kernel.setdefault(k, []).append((b, e, lane))
    out = {"dispatches": len(tl)}
    for k, v in by_kernel.items():
        out[kernel_names[k] + "_us"] = round(sum(e - b for b, e, _ in v) / len(v), 3)
    ras = sorted(by_kernel.get(1, []))
    if len(ras) > 2:
        ov = [max(0.0, min(ras[i][1], ras[i + 1][1]) - ras[i + 1][0]) for i in range(len(ras) - 1)]
        out["raster_overlap_us"] = round(sum(ov) / len(ov), 3)                        # mean overlap with the next raster kernel
        out["raster_begin_to_begin_us"] = round((ras[-1][0] - ras[0][0]) / (len(ras) - 1), 3)
        out["frame_period_us"] = round((ras[-1][1] - ras[0][1]) / (len(ras) - 1), 3)  # raster end to raster end
    gaps, last_geo = [], {}
    for k, lane, b, e in tl:                                                            # submission order: geometry, then its raster
        if k == 0:
            last_geo[lane] = e
        elif k == 1 and lane in last_geo:
            gaps.append(b - last_geo.pop(lane))
    if gaps:
        out["geometry_to_raster_gap_us"] = round(sum(gaps) / len(gaps), 3)
    return out


def main():
    args = parse_args()
    launched = "WORLD_SIZE" in os.environ
    if not launched and args.gpus > 1:
        raise SystemExit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; refusing to report a line "
                         f"whose n_gpus would not be what was asked for")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import numpy as np
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    m = ge.load_package()
    from renderer_rs_amd import build as mbuild
    from renderer_rs_amd import multigpu

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

    idle = []                                 # the rig's device, once it exists: its kernels run on queues of the library's own (native dispatch),

    def barrier():                            # which torch.cuda.synchronize() knows nothing about -- the timed region ends when THEY are empty
        for d in idle:
            d.wait_idle()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(dt: float) -> float:
        if world == 1:
            return dt
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    split = world > 1 and (args.split or "rows") == "rows"
    wname = args.workload or ("c4" if split else "c2")
    desc, make, default_fps = WORKLOADS[wname]
    fps = args.frames_per_step if args.frames_per_step > 0 else default_fps
    scene = make(m.scenes)
    fmt = m.Format.B8G8R8A8_SRGB if args.format == "bgra8" else m.Format.R32G32B32A32_SFLOAT
    bpp = 4 if args.format == "bgra8" else 16
    tris = scene.num_triangles
    use_abi_gather = split and args.gather == "abi" and args.backend == "nccl"

    keep = []                                 # torch tensors behind the wrapped buffers / targets

    def make_wrap():
        shared = {}

        def wrap_shared(device, usage, arr):  # geometry / uniforms are uploaded once and shared by all frames in flight
            key = (usage, arr.size, hashlib.sha1(arr.tobytes()).digest())
            if key not in shared:
                t = torch.from_numpy(arr.copy()).cuda()
                keep.append(t)
                shared[key] = m.Buffer.wrap(device, usage, t.data_ptr(), t.numel())
            return shared[key]
        return wrap_shared

    class Rig:
        """A device with `lanes` frames in flight: one colour target + command buffer per frame (swapchain images)."""

        def __init__(self, lanes, band=None, rows=None, per_submit=1, dev=None, scene=scene):
            self.owns_dev = dev is None
            # one GPU: a device with a stream of its own (every queue lane dispatches natively); several: on torch's stream, whose order the
            # torch.distributed fallback of the band exchange relies on (include/mirhi.h, mirhi_device_create_on_stream)
            self.dev = dev or (m.Device(local_rank) if world == 1 else m.Device(local_rank, stream=torch.cuda.current_stream().cuda_stream))
            self.scene = scene
            if self.owns_dev:
                self.dev.set_queue_lanes(lanes)
            if band is not None:
                self.dev.set_tile_split(*band)
            wrap = make_wrap()
            self.frames, self.slots, self.counter, self.per_submit = [], [], 0, per_submit
            for i in range(lanes * per_submit):
                frame = torch.zeros((rows or scene.height, scene.width, 4), dtype=torch.uint8 if bpp == 4 else torch.float32, device="cuda")
                target = m.Image(self.dev, scene.width, scene.height, fmt, device_ptr=frame.data_ptr())
                self.frames.append(frame)
                self.slots.append(m.SceneResources(self.dev, scene, fmt, color_image=target, wrap_buffers=wrap))
                self.slots[-1].cmd.set_queue_lane((i // per_submit) % lanes)      # group g = slots [g * per_submit, ...) on lane g
            self.groups = [self.slots[g * per_submit:(g + 1) * per_submit] for g in range(lanes)]

        def swapchain(self, count):
            """`count` more colour targets of this rig's scene: the swapchain images of the reference-shaped loop"""
            out = []
            for _ in range(count):
                frame = torch.zeros((self.scene.height, self.scene.width, 4), dtype=torch.uint8 if bpp == 4 else torch.float32, device="cuda")
                self.frames.append(frame)
                out.append(m.Image(self.dev, self.scene.width, self.scene.height, fmt, device_ptr=frame.data_ptr()))
            self.extra_images = getattr(self, "extra_images", []) + out
            return out

        def next_slot(self):
            i = self.counter % len(self.slots)
            self.counter += 1
            return i

        def add_groups(self, per_submit):
            """one more group of `per_submit` frames (own targets and command buffers) per queue lane of this device"""
            lanes = len(self.groups)
            wrap = make_wrap()
            added = []
            for g in range(lanes):
                grp = []
                for _ in range(per_submit):
                    frame = torch.zeros((self.scene.height, self.scene.width, 4), dtype=torch.uint8 if bpp == 4 else torch.float32, device="cuda")
                    target = m.Image(self.dev, self.scene.width, self.scene.height, fmt, device_ptr=frame.data_ptr())
                    self.frames.append(frame)
                    sl = m.SceneResources(self.dev, self.scene, fmt, color_image=target, wrap_buffers=wrap)
                    sl.cmd.set_queue_lane(g)
                    self.slots.append(sl)
                    grp.append(sl)
                added.append(grp)
            return added

        def submit_group(self):
            """per_submit frames in one mirhi_queue_submit call; returns the slots rendered"""
            g = self.groups[(self.counter // self.per_submit) % len(self.groups)]
            self.counter += self.per_submit
            self.dev.submit([sl.cmd for sl in g])
            return g

        def destroy(self, comm=None):
            self.dev.wait_idle()
            if comm is not None:
                comm.destroy()
            seen = set()
            for sl in self.slots:                     # shared buffers are destroyed once
                sl.objs = [o for o in sl.objs if not (id(o) in seen or seen.add(id(o)))]
                sl.destroy()
            for im in getattr(self, "extra_images", []):
                im.destroy()
            if self.owns_dev:
                self.dev.destroy()

    nfif = max(1, min(8, args.frames_in_flight))
    per_submit = max(1, min(8, args.frames_per_submit))
    if fps % per_submit:
        fps = (fps // per_submit + 1) * per_submit
    if split:
        nfif = min(nfif, 2)
        rows = multigpu.padded_rows(scene.height, world)      # (padding is only needed by the torch all-gather, kept as the way out)
        split_layout = args.split_layout if use_abi_gather else "bands"
        rig = Rig(nfif, band=(rank, world, split_layout), rows=rows, per_submit=per_submit)
    else:
        rig = Rig(nfif, per_submit=per_submit)
    dev = rig.dev
    idle.append(dev)

    comm, rccl_ranks = None, world
    if use_abi_gather:
        # the 128-byte RCCL id travels over the launcher's control channel; the data path is the C ABI alone
        uid = torch.zeros(m.COMM_ID_BYTES, dtype=torch.uint8, device="cuda")
        ok = torch.ones(1, dtype=torch.int32, device="cuda")
        try:
            if rank == 0:
                uid.copy_(torch.frombuffer(bytearray(m.Comm.unique_id()), dtype=torch.uint8))
        except Exception as e:
            print(f"bench.py: rank {rank}: mirhi_comm_unique_id failed: {e!r}", file=sys.stderr, flush=True)
            ok.zero_()
        dist.broadcast(uid, src=0)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()):
            try:
                comm = m.Comm(dev, bytes(uid.cpu().numpy().tobytes()), rank, world)
                rccl_ranks = comm.world()
            except Exception as e:
                print(f"bench.py: rank {rank}: mirhi_comm_create failed: {e!r}", file=sys.stderr, flush=True)
                ok.zero_()
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if not int(ok.item()):
            # every rank agrees: the exchange goes through torch.distributed's RCCL all-gather instead (same wire, other caller); the
            # line says so in config.parallelism
            if comm is not None:
                comm.destroy()
            comm, rccl_ranks = None, world
            if split_layout != "bands":          # torch's in-place all-gather moves one contiguous band per rank
                split_layout = "bands"
                dev.wait_idle()
                dev.set_tile_split(rank, world, layout="bands")
                for sl in rig.slots:
                    sl.record()
    if comm is not None:
        # the frames of the split leave the library as AQL packets on every lane (lane 0 too: nothing here relies on torch's stream order -- the exchange is the
        # C ABI's, ordered behind each frame by the library itself; the timed region ends with wait_idle)
        dev.set_native_dispatch(True)
    algo = m.GatherAlgo.DIRECT if args.gather_algo == "direct" else m.GatherAlgo.BROADCAST

    def frames():
        """per_submit frames: one submit call, then (split) the band exchange of each"""
        for sl in rig.submit_group():
            if split:
                if comm is not None:
                    comm.all_gather_bands(sl.color, sl.cmd, algo)
                else:
                    multigpu.all_gather_bands(rig.frames[rig.slots.index(sl)], rank, world, via_host=(args.backend == "gloo"))

    floop = None
    record_each = args.record_each_frame or (world == 1 and not split and per_submit == 1 and not args.resubmit and not args.profile_pass_only)
    if record_each:
        if split or world > 1 or per_submit != 1:
            raise SystemExit("bench.py: --record-each-frame measures the one-GPU frame loop (no split, one frame per submit)")
        from renderer_rs_amd import frameloop
        # Renderer::render_frame natively: nfif frames in flight, nfif + 1 swapchain images (swapchain.rs:228-236), every frame re-recorded
        floop = frameloop.FrameLoop(dev, rig.slots[0], rig.swapchain(nfif + 1), frames_in_flight=nfif)

    def step():
        if floop is not None:
            floop.run(fps)
            return
        for _ in range(fps // per_submit):
            frames()

    if args.profile_pass_only:
        for _ in range(8):                    # (every dispatch of this process is then an isolated one: what rocprofv3 averages)
            rig.slots[0].render()
            dev.wait_idle()
    else:
        # The W warm-up steps of the contract are 20 ms of work for C2 -- not enough for a freshly acquired GPU to reach its sustained
        # clocks (the first run on a fresh box read 5-11 % low, kernels included: raster 11.8 us against 10.6 us a minute later).  So the
        # frame loop first runs untimed for a fixed wall-clock spell, then the W steps, then the K timed ones.
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < args.prewarm_seconds:
            step()
        for _ in range(args.warmup):
            step()
    dev.wait_idle()                           # also reports (and acts on) the device status of the warm-up frames: a bin pool that
    barrier()                                 # turned out too small is grown before the timed region, not inside it
    stats_before = dev.stats()
    t0 = time.perf_counter()
    for _ in range(0 if args.profile_pass_only else args.steps):
        step()
    barrier()
    dt = max_over_ranks(max(time.perf_counter() - t0, 1e-9))
    stats_after = dev.stats()

    # ---- the frames the timed region produced, against the oracle (the checker; never inside the timed region) ------------------------
    # every colour target the loop rendered into (the swapchain images of the re-recorded loop / each lane's target of the resubmitted one) is read
    # back and compared with the frame the CPU oracle renders of the same scene: <= 1 code per channel of the sRGB8 target, 1e-4 on a float one.
    verify = None
    if world == 1 and not args.profile_pass_only and not split:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_binding as ob
            ref = ob.render(scene, nthreads=max(1, min(os.cpu_count() or 1, 64)), want_bgra8=(bpp == 4))
            ref_img = ref["bgra8"] if bpp == 4 else ref["rgba"]
            targets = getattr(rig, "extra_images", []) if floop is not None else [sl.color for sl in rig.slots[:nfif * per_submit]]
            worst, checked = 0.0, 0
            for img in targets:
                got = img.read()
                d = np.abs(got.astype(np.int32) - ref_img.astype(np.int32)) if bpp == 4 else np.abs(got[..., :3] - ref_img[..., :3])
                worst = max(worst, float(d.max()))
                checked += 1
            tol = 1.0 if bpp == 4 else 1e-4
            verify = {"frames_verified": checked if worst <= tol else 0, "targets": checked, "max_abs_diff": worst, "tolerance": tol,
                      "unit": "sRGB8 codes" if bpp == 4 else "linear float", "against": "oracle/mirhi_oracle.c frame of the same scene"}
            if worst > tol:
                print(f"bench.py: the timed region's frames differ from the oracle's by {worst} (> {tol}): the line is not to be trusted", file=sys.stderr, flush=True)
        except Exception as e:
            verify = {"frames_verified": 0, "error": repr(e)}

    if split and not args.profile_pass_only:
        # every rank must now hold the same, complete frame
        chk = torch.stack([f[:scene.height].to(torch.int64).sum() for f in rig.frames]).cpu() if bpp == 4 else None
        if chk is not None and world > 1:
            lo, hi = chk.clone(), chk.clone()
            if args.backend == "nccl":
                lo, hi = lo.cuda(), hi.cuda()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            if not torch.equal(lo.cpu(), hi.cpu()):
                raise SystemExit("bench.py: the ranks disagree about the gathered frame (checksums differ): the band exchange is broken")

    # ---- per-dispatch device time (event pair attached to each dispatch: begin -> end on the GPU clock) -------------------
    prof_frames = max(64, min(1024, fps))
    # (a) one frame in flight: isolated kernel durations (kernels of different frames cannot stretch each other)
    dev.wait_idle()
    dev.reset_kernel_times()
    dev.set_profiling(m.Profile.TIMING)
    for _ in range(prof_frames):
        rig.slots[0].render()
        if args.profile_pass_only:
            dev.wait_idle()                   # (under a tracer / counter collection: strictly one frame on the GPU at a time)
    dev.wait_idle()
    iso = {name: dev.kernel_time(k) for k, name in enumerate(m.Kernel.NAMES)}
    iso_tl = summarize_timeline(dev.timeline(), m.Kernel.NAMES)
    # (b) the timed region's conditions: all lanes in flight (no band exchange: kernels only)
    dev.reset_kernel_times()
    flight_tl_raw = []
    if not args.profile_pass_only:
        for _ in range(prof_frames):
            rig.slots[rig.next_slot()].render()
        dev.wait_idle()
        flight_tl_raw = dev.timeline()
    flight_tl = summarize_timeline(flight_tl_raw, m.Kernel.NAMES)
    dev.set_profiling(0)
    if args.timeline_out and rank == 0:
        with open(args.timeline_out, "w") as f:
            json.dump({"kernels": m.Kernel.NAMES, "frames_in_flight": nfif, "dispatches": flight_tl_raw}, f)
    # (c) fragment statistics: never part of a timed frame
    dev.reset_kernel_times()
    shaded, covered, stat_frames = 0, 0, 2
    if not args.profile_pass_only:
        dev.set_profiling(m.Profile.FRAGMENTS)
        for _ in range(stat_frames):
            rig.slots[0].render()
        dev.set_profiling(0)
        shaded, covered, _scopes = dev.fragment_stats()
    dev.reset_kernel_times()
    stats = dev.stats()
    dispatch_path = dev.dispatch_path()
    build_id = m.lib().mirhi_build_id().decode()
    extras = {}
    if verify is not None:
        extras["frames_verified"] = verify.pop("frames_verified")
        extras["verification"] = verify
    if world == 1 and per_submit == 1 and not args.no_extras and not args.profile_pass_only:
        # the same frame loop with 8 command buffers per mirhi_queue_submit call (vkQueueSubmit with several command buffers): the
        # frames of a call share one batch of launches.  Same device and queue lanes (a second device would be dealt other hardware
        # queues); reported beside the headline, which keeps one frame per submit.
        try:
            groups = rig.add_groups(8)
            for i in range(max(8, fps // 8 // 4)):
                dev.submit([sl.cmd for sl in groups[i % len(groups)]])
            dev.wait_idle()
            nb = max(16, 4 * fps // 8)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(nb):
                dev.submit([sl.cmd for sl in groups[i % len(groups)]])
            dev.wait_idle()
            d1 = time.perf_counter() - t1
            extras["batched_submit"] = {"value": round(tris * nb * 8 / d1 / 1e6, 3), "unit": "Mtris/s", "frames_per_submit": 8, "queue_lanes": nfif,
                                        "frames": nb * 8, "us_per_frame": round(1e6 * d1 / (nb * 8), 4)}
        except Exception as e:
            extras["batched_submit"] = {"error": repr(e)}
    def rerecorded(rig_, fif, frames_, vary=0, phases=False, uniform=None):
        """the reference-shaped loop (wait fence -> reset -> re-record -> end -> submit with fence) on `rig_`'s device, natively"""
        from renderer_rs_amd import frameloop
        loop = frameloop.FrameLoop(rig_.dev, rig_.slots[0], rig_.swapchain(fif + 1), frames_in_flight=fif, vary_triangles=vary, per_frame_uniform=uniform)
        try:
            loop.run(max(32, frames_ // 4))
            sec = min(loop.run(frames_) for _ in range(2))
            out_ = {"value": round(rig_.scene.num_triangles * frames_ / sec / 1e6, 3), "unit": "Mtris/s", "us_per_frame": round(1e6 * sec / frames_, 4),
                    "frames_in_flight": fif, "fence_gated": True, "frames": frames_}
            if phases:
                loop.phase_seconds(True)
                loop.run(frames_)
                ph = loop.phase_seconds(False)
                out_["host_us_per_frame"] = {k: round(1e6 * v / frames_, 3) for k, v in zip(("fence_wait", "record", "end", "submit"), ph)}
            return out_
        finally:
            loop.destroy()

    def resubmitted(rig_, slots_, frames_):
        for i in range(max(16, frames_ // 8)):
            slots_[i % len(slots_)].render()
        rig_.dev.wait_idle()
        t1 = time.perf_counter()
        for i in range(frames_):
            slots_[i % len(slots_)].render()
        rig_.dev.wait_idle()
        d1 = time.perf_counter() - t1
        return {"value": round(rig_.scene.num_triangles * frames_ / d1 / 1e6, 3), "unit": "Mtris/s", "us_per_frame": round(1e6 * d1 / frames_, 4),
                "command_buffers": len(slots_), "queue_lanes": len(slots_), "fence_gated": False, "frames": frames_,
                "note": "recorded once, resubmitted round-robin without waiting on fences (the lanes queue up)"}

    if world == 1 and per_submit == 1 and not args.no_extras and not args.profile_pass_only:
        try:
            if floop is not None:
                floop.destroy()
                floop = None
            n_re = max(512, 4 * fps)
            # the reference's frame loop on the headline's device and queue lanes: every frame re-recorded and fenced (renderer.rs:367-557)
            extras["rerecorded_submit"] = rerecorded(rig, nfif, n_re, phases=True)
            # ... and with a triangle count that changes from frame to frame (another launch plan every frame)
            extras["rerecorded_submit"]["changing_triangle_count"] = rerecorded(rig, nfif, n_re, vary=7)
        except Exception as e:
            extras["rerecorded_submit"] = {"error": repr(e)}
        try:
            # the headline of rounds 1-3: the same command buffers recorded once and resubmitted round-robin, no fences (the lanes queue up)
            extras["resubmitted_submit"] = resubmitted(rig, rig.slots[:nfif], n_re)
        except Exception as e:
            extras["resubmitted_submit"] = {"error": repr(e)}
        try:
            if os.environ.get("MIRHI_BENCH_SKIP_FIF2"):
                raise RuntimeError("skipped (MIRHI_BENCH_SKIP_FIF2)")
            # the reference's MAX_FRAMES_IN_FLIGHT = 2 (crates/renderer/src/lib.rs:43): two queue lanes, two command buffers
            dev.wait_idle()
            dev.set_queue_lanes(2)
            two = rig.slots[:2]
            for i, sl in enumerate(two):
                sl.cmd.set_queue_lane(i)
            n2 = max(512, 2 * fps)
            extras["frames_in_flight_2"] = {"resubmitted": resubmitted(rig, two, n2), "rerecorded": rerecorded(rig, 2, n2, phases=True)}
            # Where a fence-gated frame's microseconds go: ONE frame in flight is the whole chain per frame (host -> doorbell -> geometry -> raster ->
            # signal -> host); its parts measured separately: host phases (clock reads in the native loop), the kernels alone (event pairs), the round trip
            # of an empty one-wave kernel on the same queue (doorbell -> packet processor -> wave -> release -> signal -> host: everything around the
            # kernels that is not the host's) and of a barrier packet (no wave).  What is left is the boundary between the two kernels plus whatever the
            # parts cost more when chained than alone.
            one = rerecorded(rig, 1, max(256, n2 // 2), phases=True)
            rt_kernel, rt_barrier = dev.measure_roundtrip(0, 300)
            g_us = 1e3 * iso["geometry"][0] / max(1, iso["geometry"][1]); r_us = 1e3 * iso["raster"][0] / max(1, iso["raster"][1]); v_us = 1e3 * iso["vertex"][0] / max(1, iso["vertex"][1])
            hostp = one["host_us_per_frame"]
            host_us = hostp["record"] + hostp["end"] + hostp["submit"]
            extras["frames_in_flight_2"]["chain_us"] = {
                "frame_latency_one_in_flight": one["us_per_frame"], "host": round(host_us, 3), "host_phases": {k: hostp[k] for k in ("record", "end", "submit")},
                "doorbell_to_wave_plus_signal_to_host": round(rt_kernel, 3), "barrier_packet_round_trip": round(rt_barrier, 3),
                "vertex": round(v_us, 3), "geometry": round(g_us, 3), "raster": round(r_us, 3),
                "boundary_and_rest": round(one["us_per_frame"] - host_us - rt_kernel - g_us - r_us - v_us, 3),
                "how": "frame_latency = us per frame of the re-recorded loop with ONE frame in flight; host = its record + end + submit phases; doorbell..host = round trip "
                       "of an empty one-wave kernel on the same AQL queue (mirhi_device_measure_roundtrip); kernels = isolated durations (event pairs); the rest by difference"}
            dev.wait_idle()
            dev.set_queue_lanes(nfif)
            for i, sl in enumerate(rig.slots[:nfif]):
                sl.cmd.set_queue_lane(i % nfif)
        except Exception as e:
            extras["frames_in_flight_2"] = {"error": repr(e)}
    if world == 1 and per_submit == 1 and not args.no_extras and not args.profile_pass_only and args.other_workloads:
        # The other single-GPU BASELINE workloads, one short pass each on the same device and queue lanes (the headline fields above stay
        # the C2 figures): frame loop of pre-recorded command buffers with `nfif` frames in flight, the reference-shaped re-recorded loop,
        # and the isolated kernel durations behind their own roofline.
        wl_out = {}
        for other in [w_ for w_ in args.other_workloads.split(",") if w_ and w_ in WORKLOADS and w_ != wname]:
            try:
                odesc, omake, _ofps = WORKLOADS[other]
                osc = omake(m.scenes)
                orig = Rig(nfif, dev=dev, scene=osc)
                t1 = time.perf_counter()
                nwarm = 0
                while time.perf_counter() - t1 < 0.15:
                    orig.slots[nwarm % nfif].render()
                    nwarm += 1
                dev.wait_idle()
                us_guess = 1e6 * (time.perf_counter() - t1) / max(1, nwarm)
                nfr = int(max(64, min(4096, 0.3e6 / max(1.0, us_guess))))
                res_o = resubmitted(orig, orig.slots, nfr)
                rer_o = rerecorded(orig, nfif, nfr)
                # ... and with the frame's own object block rewritten first (Buffer::write_data, buffer.rs:247-279; one block per frame in flight)
                rer_u = rerecorded(orig, nfif, nfr, uniform=int(m.Slot.OBJECT)) if orig.slots[0].draw_state[0]["object"] is not None else None
                dev.wait_idle(); dev.reset_kernel_times(); dev.set_profiling(m.Profile.TIMING)
                for _ in range(32):
                    orig.slots[0].render()
                dev.wait_idle()
                kt = {name: dev.kernel_time(k) for k, name in enumerate(m.Kernel.NAMES)}
                dev.set_profiling(0); dev.reset_kernel_times()
                k_us = {name: (1e3 * ms / n_ if n_ else 0.0) for name, (ms, n_) in kt.items()}
                oalg = osc.algorithmic_bytes(bpp_out=bpp)
                otraffic, onote = measured_traffic(other, mbuild.source_hash())
                ach = oalg / (k_us["raster"] * 1e-6) / 1e9 if k_us["raster"] > 0 else 0.0
                fus = k_us["raster"] + k_us["geometry"] + k_us["vertex"]
                wl_out[other] = {"workload": f"{other}: {odesc}", "triangles": osc.num_triangles, "width": osc.width, "height": osc.height,
                                 "value": res_o["value"], "unit": "Mtris/s", "us_per_frame": res_o["us_per_frame"], "frames": nfr, "frames_in_flight": nfif,
                                 "rerecorded_submit": {"value": rer_o["value"], "us_per_frame": rer_o["us_per_frame"],
                                                       "uniform_write_per_frame": {"value": rer_u["value"], "us_per_frame": rer_u["us_per_frame"]} if rer_u else None},
                                 "roofline": {"bound": "hbm", "kernel": "raster_kernel", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                              "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": otraffic["frame_hbm_bytes"] if otraffic else None,
                                              "traffic_structural": otraffic.get("structural_bytes") if otraffic else None,
                                              "traffic_source": onote, "algorithmic_bytes_per_launch": oalg, "avg_kernel_us": round(k_us["raster"], 3),
                                              "geometry_kernel_us": round(k_us["geometry"], 3), "vertex_kernel_us": round(k_us["vertex"], 3),
                                              "frame_frac": round(oalg / (fus * 1e-6) / 1e9 / HBM_PEAK_GBS, 5) if fus > 0 else None,
                                              "issue": issue_bound((otraffic or {}).get("instructions", {}).get("raster_kernel"), k_us["raster"])}}
                orig.destroy()
            except Exception as e:
                wl_out[other] = {"error": repr(e)}
        extras["workloads"] = wl_out
    if world > 1 and split and not args.no_extras and not args.profile_pass_only:
        try:
            # the same workload, whole frame on ONE GPU (every rank renders its own unsplit copy, no exchange): the N = 1 point of the
            # curve.  Same device and queue lanes, re-recorded without the split.
            dev.wait_idle()
            dev.set_tile_split(0, 1)
            for sl in rig.slots:
                sl.record()
            n = max(8, fps)
            for _ in range(max(1, n // 4 // per_submit)):
                rig.submit_group()
            barrier()
            t1 = time.perf_counter()
            for _ in range(n // per_submit):
                rig.submit_group()
            barrier()
            d1 = max_over_ranks(time.perf_counter() - t1)
            extras["one_gpu_same_workload"] = {"value": round(tris * (n // per_submit) * per_submit / d1 / 1e6, 3), "unit": "Mtris/s", "frames": (n // per_submit) * per_submit,
                                               "note": "whole frame per GPU, no split, no exchange (all ranks at once, slowest rank)"}
        except Exception as e:          # never let a secondary measurement take the primary line down
            extras["one_gpu_same_workload"] = {"error": repr(e)}
    if floop is not None:
        floop.destroy()
    rig.destroy(comm)        # (frees its queue lanes: a second device beside it would share the 4 hardware queues with it)

    if rank == 0:
        frames_total = args.steps * fps * (1 if split or world == 1 else world)
        value = tris * frames_total / dt / 1e6
        alg_bytes = scene.algorithmic_bytes(bpp_out=bpp)
        if split:   # all geometry + this rank's share of the frame buffer
            alg_bytes = alg_bytes - scene.width * scene.height * bpp + scene.width * scene.height * bpp // world
        ras_ms, ras_n = iso["raster"]
        geo_ms, geo_n = iso["geometry"]
        vs_ms, vs_n = iso["vertex"]
        ras_us = 1e3 * ras_ms / max(1, ras_n)
        geo_us = 1e3 * geo_ms / max(1, geo_n)
        vs_us = 1e3 * vs_ms / max(1, vs_n)
        achieved = alg_bytes / (ras_us * 1e-6) / 1e9 if ras_us > 0 else 0.0
        frame_us = ras_us + geo_us + vs_us
        src_hash = mbuild.source_hash()
        traffic, traffic_note = (None, "split run") if split else measured_traffic(wname, src_hash)
        shaded_per_frame = shaded / max(1, stat_frames)
        out = {
            "metric": "Mtris/s at 1920x1080 (input triangles per second, whole frame incl. shading + store)"
                      if scene.height == 1080 else "Mtris/s (input triangles per second, whole frame incl. shading + store)",
            "value": round(value, 3), "unit": "Mtris/s", "n_gpus": rccl_ranks, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / max(1, args.steps), 6), "higher_is_better": True,
            "scaling": "strong" if split else "weak", "vs_baseline": None, "dtype": "f32+i32",
            "data": "synthetic",
            "config": {"workload": f"{wname}: {desc}", "triangles": tris, "width": scene.width, "height": scene.height,
                       "target_format": "B8G8R8A8_SRGB" if bpp == 4 else "R32G32B32A32_SFLOAT",
                       "parallelism": (f"tile-row split x{world} ({split_layout if split else ''}) + band exchange ({'RCCL through the C ABI, ' + args.gather_algo if comm is not None else 'torch.distributed ' + args.backend})" if split
                                       else (f"afr{world}" if world > 1 else "single")),
                       "frames_per_step": fps, "prewarm_seconds": args.prewarm_seconds, "frames_in_flight": nfif * per_submit, "queue_lanes": nfif, "frames_per_submit": per_submit,
                       "command_buffers": "re-recorded every frame (wait fence, reset, record, end, submit with fence: renderer.rs:367-557), native loop"
                                          if record_each else "recorded once, resubmitted"},
            "timed_region_s": round(dt, 6), "us_per_frame": round(1e6 * dt / max(1, args.steps * fps), 4),
            # how the timed region's kernels left the library (include/mirhi.h, mirhi_device_dispatch_path): hand-written AQL packets or HIP launches --
            # a silent fallback would show here -- and how many of each the timed region made
            "dispatch_path": dispatch_path, "native_dispatches": int(stats_after.native_dispatches - stats_before.native_dispatches) & 0xFFFFFFFF,
            "frames_submitted": int(stats_after.frames_submitted - stats_before.frames_submitted),
            "build_id": build_id, "build_matches_sources": build_id == src_hash,
            "shaded_mpix_per_s": round(shaded_per_frame * frames_total / dt / 1e6, 1),
            "overdraw": round(covered / shaded, 4) if shaded else None,
            "shaded_pixels_per_frame": int(shaded_per_frame), "covered_fragments_per_frame": int(covered / max(1, stat_frames)),
            "target_mpix_per_s": round(scene.width * scene.height * frames_total / dt / 1e6, 1),
            "roofline": {"bound": "hbm", "kernel": "raster_kernel", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": traffic["frame_hbm_bytes"] if traffic else None, "traffic_scope": "whole frame: vertex + geometry + raster kernels",
                         "traffic_structural": traffic.get("structural_bytes") if traffic else None,   # algorithmic bytes + the intermediate streams (shaded vertices, bin records) written once and read once
                         "traffic_source": traffic_note, "kernel_source_sha16": src_hash,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_kernel_us": round(ras_us, 3),
                         "geometry_kernel_us": round(geo_us, 3), "vertex_kernel_us": round(vs_us, 3),
                         "frame_kernels_us": round(frame_us, 3),
                         "frame_frac": round(alg_bytes / (frame_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5) if frame_us > 0 else None,
                         "issue": issue_bound((traffic or {}).get("instructions", {}).get("raster_kernel"), ras_us),
                         "timed_launches": ras_n,
                         "how": "event pair attached to every dispatch (hipExtLaunchKernelGGL start/stop events = the dispatch's begin and end on "
                                "the GPU clock, as rocprofv3 --kernel-trace reports them; nothing subtracted); a pass of frames behind the timed "
                                "region with ONE frame in flight, so kernels of different frames do not stretch each other"},
            "timeline_isolated": iso_tl, "timeline_in_flight": flight_tl,
            "workspace_mb": round(stats.workspace_bytes / 1e6, 1), "big_list": stats.last_big_list,
        }
        out.update(extras)
        if not args.no_cpu_baseline and world == 1 and not args.profile_pass_only:
            try:
                out["cpu_baseline"] = cpu_baseline(scene, args.cpu_seconds)
            except Exception as e:  # the oracle is a reported baseline, never the measured path
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
