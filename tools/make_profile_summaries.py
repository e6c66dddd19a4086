"""Turns gpurun_out/prof_<tag>/ (tools/collect_profiles.sh) into the per-round files under profiles/.
usage: make_profile_summaries.py <tag>"""
import collections, csv, glob, json, os, shutil, sys
# options: --workloads c2,c3 (the per-workload part for these only), --no-trace (skip the timed-region trace part), --summary-only (only rNN_SUMMARY.md, from the
# files already in the destination): a collection may run as several gpurun calls (20 minutes each at most) whose summaries are merged here
opts = [a for a in sys.argv[1:] if a.startswith("--")]
sys.argv = [sys.argv[0]] + [a for a in sys.argv[1:] if not a.startswith("--")]
only = next((o.split("=", 1)[1].split(",") for o in opts if o.startswith("--workloads=")), ["c2", "c3", "c4", "c5"])
no_trace, summary_only = "--no-trace" in opts or "--summary-only" in opts, "--summary-only" in opts
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
# second argument: where to write (default profiles/).  On the GPU box the raw traces are too big to travel back (gpurun merges <= 64 MiB), so
# tools/collect_profiles.sh runs this there into gpurun_out/prof_<tag>/summary and drops the raw trace; the files are then copied into profiles/ here.
dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def one(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    if not hits:
        raise SystemExit("missing " + pattern)
    return max(hits, key=os.path.getmtime)


def last_json(path):
    lines = [l for l in open(path) if l.startswith("{")]
    return json.loads(lines[-1]) if lines else None


def short(kern):
    for k in ("raster_kernel", "geometry_kernel", "vertex_kernel", "fragment_count_kernel", "winner_count_kernel", "ordered_kernel"):
        if k in kern:
            return k
    return None


def means(pass_dir):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(one(pass_dir + "/**/*counter_collection.csv"))):
        acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v)) for k, v in acc.items()}


for w in ([] if summary_only else only):
    shutil.copy(one(f"stats_{w}/**/*kernel_stats.csv"), os.path.join(dst, f"{tag}_{w}_rocprofv3_kernel_stats.csv"))
    line = last_json(os.path.join(src, f"bench_profile_pass_{w}.json"))
    json.dump(line, open(os.path.join(dst, f"{tag}_{w}_bench_profile_pass.json"), "w"), indent=1)
    rows, per_kernel = [], collections.defaultdict(dict)
    for pass_dir in (f"pmc_sq_{w}", f"pmc_fetch_{w}", f"pmc_write_{w}", f"pmc_rdreq_{w}"):
        if not glob.glob(os.path.join(src, pass_dir) + "/**/*counter_collection.csv", recursive=True):
            continue
        for (kern, ctr), (n, mean) in sorted(means(pass_dir).items()):
            if "mirhi::" not in kern:
                continue
            rows.append((pass_dir, kern, ctr, n, round(mean, 1)))
            if short(kern):
                per_kernel[short(kern)][ctr] = mean
    with open(os.path.join(dst, f"{tag}_{w}_rocprofv3_pmc_summary.csv"), "w") as f:
        cw = csv.writer(f)
        cw.writerow(["pass", "kernel", "counter", "launches", "mean_per_launch"])
        cw.writerows(rows)
    traffic = {"workload": w, "kernel_source_sha16": line["roofline"]["kernel_source_sha16"],
               "source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), python3 bench.py --workload {w} --profile-pass-only",
               "unit_note": "FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B "
                            "(MI355X_MICROARCH.md, HBM section), so the read side is doubled.  The guide calibrates that for 16 B/lane streaming reads; "
                            "read_bytes_exact (the L2's read requests by size, TCC_EA0_RDREQ_{64B,128B}, and in 32-byte units to DRAM, "
                            "TCC_EA0_RDREQ_DRAM_32B: a separate pass) shows it holds for these kernels' gathers as well: the L2 fetches whole 128-byte lines",
               "algorithmic_bytes": line["roofline"]["algorithmic_bytes_per_launch"]}
    # the SQ instruction mix per launch (pmc_sq pass): what the issue bound of a kernel is computed from (bench.py: roofline.issue_frac)
    for k, d in per_kernel.items():
        if "SQ_INSTS_VALU" in d:
            traffic.setdefault("instructions", {})[k] = {c.replace("SQ_INSTS_", "").lower(): int(round(d[c])) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM") if c in d}
            if "SQ_WAVES" in d:
                traffic["instructions"][k]["waves"] = int(round(d["SQ_WAVES"]))
    frame = 0
    for k, d in per_kernel.items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            b = int(round((2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024))
            traffic[k] = {"fetch_size_kb": round(d["FETCH_SIZE"], 2), "write_size_kb": round(d["WRITE_SIZE"], 2), "hbm_bytes_per_launch": b}
            if "TCC_EA0_RDREQ_sum" in d:
                n, n64, n128 = d["TCC_EA0_RDREQ_sum"], d.get("TCC_EA0_RDREQ_64B_sum", 0.0), d.get("TCC_EA0_RDREQ_128B_sum", 0.0)
                traffic[k]["read_bytes_exact"] = {"requests": round(n), "of_64B": round(n64), "of_128B": round(n128), "by_size": int(round(32 * (n - n64 - n128) + 64 * n64 + 128 * n128)),
                                                  "dram_32B_units": int(round(32 * d.get("TCC_EA0_RDREQ_DRAM_32B_sum", 0.0))), "doubled_fetch_size": int(round(2048 * d["FETCH_SIZE"]))}
            frame += b
    traffic["frame_hbm_bytes"] = frame
    traffic["frame_over_algorithmic"] = round(frame / traffic["algorithmic_bytes"], 3)
    # what a binned rasterizer moves by construction: the algorithmic bytes plus every intermediate stream written once and read once (shaded vertices: the
    # vertex kernel's writes; bin records + flat colours: the geometry kernel's writes)
    inter = sum(int(round(traffic[k]["write_size_kb"] * 1024)) for k in ("vertex_kernel", "geometry_kernel") if k in traffic)
    traffic["structural_bytes"] = traffic["algorithmic_bytes"] + 2 * inter
    traffic["frame_over_structural"] = round(frame / traffic["structural_bytes"], 3)
    json.dump(traffic, open(os.path.join(dst, f"{tag}_{w}_hbm_traffic.json"), "w"), indent=1)
    print(json.dumps(traffic, indent=1))

if not summary_only and os.path.exists(os.path.join(src, "bench_default.json")):
    json.dump(last_json(os.path.join(src, "bench_default.json")), open(os.path.join(dst, f"{tag}_c2_bench_default.json"), "w"), indent=1)
for w in ([] if summary_only else ("c3", "c4", "c5")):
    j = last_json(os.path.join(src, f"bench_{w}.json")) if os.path.exists(os.path.join(src, f"bench_{w}.json")) else None
    if j:
        json.dump(j, open(os.path.join(dst, f"{tag}_{w}_bench.json"), "w"), indent=1)

summary = None
if not no_trace:
    # ---- the 4-lane timed region as the tracer sees it: overlap of consecutive raster kernels, geometry -> raster gaps ----------
    trace = one("trace/**/*kernel_trace.csv")
    rows = []
    for r in csv.DictReader(open(trace)):
        k = short(r["Kernel_Name"])
        if k in ("raster_kernel", "geometry_kernel"):
            rows.append((k, int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")))
    rows.sort(key=lambda x: x[1])
    t0 = rows[0][1]
    ras = [x for x in rows if x[0] == "raster_kernel"]
    geo = [x for x in rows if x[0] == "geometry_kernel"]
    # steady state: the middle half of the trace
    lo, hi = len(ras) // 4, 3 * len(ras) // 4
    mid = ras[lo:hi]
    period = (mid[-1][2] - mid[0][2]) / (len(mid) - 1) / 1e3
    overlap = sum(max(0, min(mid[i][2], mid[i + 1][2]) - mid[i + 1][1]) for i in range(len(mid) - 1)) / (len(mid) - 1) / 1e3
    dur_r = sum(x[2] - x[1] for x in mid) / len(mid) / 1e3
    gmid = [x for x in geo if mid[0][1] <= x[1] <= mid[-1][1]]
    dur_g = sum(x[2] - x[1] for x in gmid) / max(1, len(gmid)) / 1e3
    # union of raster-busy time / wall time in the window
    busy, cur_b, cur_e = 0, None, None
    for _, b, e, _q in mid:
        if cur_e is None or b > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_b
            cur_b, cur_e = b, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_b
    wall = mid[-1][2] - mid[0][1]
    queues = sorted(set(x[3] for x in rows))
    summary = {
        "command": "rocprofv3 --kernel-trace -- python3 bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline --no-extras (C2, 4 queue lanes, 512 frames per step)",
        "raster_dispatches": len(ras), "geometry_dispatches": len(geo), "hardware_queues_seen": queues,
        "window": f"raster dispatches {lo}..{hi} of {len(ras)} (steady state)",
        "raster_us": round(dur_r, 3), "geometry_us": round(dur_g, 3),
        "frame_period_us": round(period, 3), "raster_overlap_with_next_us": round(overlap, 3),
        "raster_busy_fraction_of_wall": round(busy / wall, 4),
        "traced_bench_value_mtris_per_s": (last_json(os.path.join(src, "bench_traced.json")) or {}).get("value"),
        "untraced_bench_value_mtris_per_s": (last_json(os.path.join(src, "bench_default.json")) or {}).get("value"),
    }
    json.dump(summary, open(os.path.join(dst, f"{tag}_c2_timed_region_trace_summary.json"), "w"), indent=1)
    print(json.dumps(summary, indent=1))
    with open(os.path.join(dst, f"{tag}_c2_timed_region_trace_excerpt.csv"), "w") as f:
        cw = csv.writer(f)
        cw.writerow(["kernel", "begin_us", "end_us", "queue"])
        for k, b, e, q in [x for x in rows if mid[0][1] <= x[1]][:64]:
            cw.writerow([k, round((b - t0) / 1e3, 3), round((e - t0) / 1e3, 3), q])
    tl = os.path.join(src, "timeline_in_flight.json")
    if os.path.exists(tl):
        j = json.load(open(tl))
        j["dispatches"] = j["dispatches"][:256]
        json.dump(j, open(os.path.join(dst, f"{tag}_c2_timeline_in_flight_excerpt.json"), "w"))


# ---- native dispatch under counter collection (collect_profiles.sh, pmc_native): kernels per queue, counters of the natively dispatched ones -------------
native = None
if summary_only:
    native = json.load(open(os.path.join(dst, f"{tag}_c2_pmc_native_dispatch.json"))) if os.path.exists(os.path.join(dst, f"{tag}_c2_pmc_native_dispatch.json")) else None
    summary = json.load(open(os.path.join(dst, f"{tag}_c2_timed_region_trace_summary.json"))) if os.path.exists(os.path.join(dst, f"{tag}_c2_timed_region_trace_summary.json")) else None
nat_trace = [] if summary_only else glob.glob(os.path.join(src, "pmc_native") + "/**/*kernel_trace.csv", recursive=True)
nat_ctr = [] if summary_only else glob.glob(os.path.join(src, "pmc_native") + "/**/*counter_collection.csv", recursive=True)
if nat_trace and nat_ctr:
    per_queue = collections.Counter()
    for r in csv.DictReader(open(nat_trace[0])):
        if "mirhi::" in r["Kernel_Name"]:
            per_queue[(r["Queue_Id"], short(r["Kernel_Name"]))] += 1
    waves = collections.defaultdict(list)
    for r in csv.DictReader(open(nat_ctr[0])):
        if "mirhi::" in r["Kernel_Name"] and r["Counter_Name"] == "SQ_WAVES":
            waves[(r["Queue_Id"], short(r["Kernel_Name"]))].append(float(r["Counter_Value"]))
    line = last_json(os.path.join(src, "bench_pmc_native.json")) or {}
    native = {"command": "MIRHI_NATIVE_DISPATCH=2 rocprofv3 --kernel-trace --pmc SQ_WAVES -- python3 bench.py --workload c2 --profile-pass-only --no-cpu-baseline",
              "dispatch_path": line.get("dispatch_path"), "completed": bool(line),
              "dispatches_per_queue_and_kernel": {f"queue {q} {k}": n for (q, k), n in sorted(per_queue.items())},
              "sq_waves_mean_per_queue_and_kernel": {f"queue {q} {k}": round(sum(v) / len(v), 1) for (q, k), v in sorted(waves.items())},
              "note": "the queue with 8 geometry + 8 raster dispatches is the library's own AQL queue (the warm-up frames of the profile pass, dispatched natively through the tool's "
                      "proxy queue); the other carries the timed pass, which launches through HIP"}
    json.dump(native, open(os.path.join(dst, f"{tag}_c2_pmc_native_dispatch.json"), "w"), indent=1)

# ---- SUMMARY: every figure quoted anywhere about this collection comes out of the files written above (no hand-copied numbers) --------------------------
def load(name):
    path = os.path.join(dst, name)
    return json.load(open(path)) if os.path.exists(path) else None

md = [f"# {tag}: figures of this collection (GENERATED by tools/make_profile_summaries.py from the files beside it -- do not edit)", ""]
d = load(f"{tag}_c2_bench_default.json")
if d:
    md += [f"Build `{d.get('build_id')}` (matches the sources: {d.get('build_matches_sources')}); dispatch path: {d.get('dispatch_path')}.", "",
           "## C2 driver line (`python3 bench.py --gpus 1 --steps 20 --warmup 5`)", "",
           "| figure | value |", "|---|---|",
           f"| headline: {d['config']['command_buffers'][:40]}..., {d['config']['frames_in_flight']} frames in flight | **{d['value']} Mtris/s** ({d['us_per_frame']} us per frame) |",
           f"| frames verified against the oracle | {d.get('frames_verified')} (max |diff| {(d.get('verification') or {}).get('max_abs_diff')} {(d.get('verification') or {}).get('unit')}) |",
           f"| native dispatches in the timed region | {d.get('native_dispatches')} for {d.get('frames_submitted')} frames |"]
    rs, rr, f2 = d.get("resubmitted_submit") or {}, d.get("rerecorded_submit") or {}, d.get("frames_in_flight_2") or {}
    md += [f"| resubmitted (recorded once, no fences) | {rs.get('value')} Mtris/s |",
           f"| re-recorded, changing triangle count | {(rr.get('changing_triangle_count') or {}).get('value')} Mtris/s |",
           f"| host us per frame (re-recorded loop) | {rr.get('host_us_per_frame')} |",
           f"| 2 frames in flight, re-recorded + fenced | **{(f2.get('rerecorded') or {}).get('value')} Mtris/s** ({(f2.get('rerecorded') or {}).get('us_per_frame')} us) |",
           f"| 2 frames in flight, resubmitted | {(f2.get('resubmitted') or {}).get('value')} Mtris/s |",
           f"| chain of one fenced frame (us) | {json.dumps({k: v for k, v in (f2.get('chain_us') or {}).items() if k != 'how'})} |",
           f"| batched submit (8 frames per call) | {(d.get('batched_submit') or {}).get('value')} Mtris/s |",
           f"| shaded Mpix/s, overdraw | {d.get('shaded_mpix_per_s')}, {d.get('overdraw')} |",
           f"| CPU oracle | {(d.get('cpu_baseline') or {}).get('value')} Mtris/s on {(d.get('cpu_baseline') or {}).get('cores')} threads, {((d.get('cpu_baseline') or {}).get('single_thread') or {}).get('value')} on one |", ""]
    wl = d.get("workloads") or {}
    if wl:
        md += ["## other workloads in the same line", "", "| workload | Mtris/s (4 in flight) | re-recorded | us per frame | raster / geometry / vertex us | HBM frac (raster) |", "|---|---|---|---|---|---|"]
        for k, v in wl.items():
            if "error" in v:
                md.append(f"| {k} | error: {v['error']} | | | | |"); continue
            r = v["roofline"]
            md.append(f"| {k} | {v['value']} | {v['rerecorded_submit']['value']} | {v['us_per_frame']} | {r['avg_kernel_us']} / {r['geometry_kernel_us']} / {r['vertex_kernel_us']} | {r['frac']} |")
        md.append("")
md += ["## isolated pass under rocprofv3 (`--kernel-trace --stats`, `bench.py --workload cN --profile-pass-only`) and the PMC passes", "",
       "| workload | kernel | tracer avg us | event-pair avg us of the same (traced) run | VALU / SALU wave-instructions per launch | HBM bytes per launch |", "|---|---|---|---|---|---|"]
for w in ("c2", "c3", "c4", "c5"):
    stats_path, tr, pp = os.path.join(dst, f"{tag}_{w}_rocprofv3_kernel_stats.csv"), load(f"{tag}_{w}_hbm_traffic.json"), load(f"{tag}_{w}_bench_profile_pass.json")
    if not os.path.exists(stats_path):
        continue
    rows = {}
    for r in csv.DictReader(open(stats_path)):
        k = short(r.get("Name", ""))
        if k in ("raster_kernel", "geometry_kernel", "vertex_kernel") and k not in rows:
            rows[k] = float(r["AverageNs"]) / 1e3
    ev = {"raster_kernel": (pp or {}).get("roofline", {}).get("avg_kernel_us"), "geometry_kernel": (pp or {}).get("roofline", {}).get("geometry_kernel_us"), "vertex_kernel": (pp or {}).get("roofline", {}).get("vertex_kernel_us")}
    for k, us in rows.items():
        ins = ((tr or {}).get("instructions") or {}).get(k, {})
        md.append(f"| {w} | {k} | {us:.2f} | {ev.get(k)} | {ins.get('valu')} / {ins.get('salu')} | {((tr or {}).get(k) or {}).get('hbm_bytes_per_launch')} |")
md.append("")
md += ["## HBM traffic per frame (FETCH_SIZE x 2 + WRITE_SIZE, separate PMC passes)", "", "| workload | frame bytes | algorithmic | x algorithmic | structural | x structural |", "|---|---|---|---|---|---|"]
for w in ("c2", "c3", "c4", "c5"):
    tr = load(f"{tag}_{w}_hbm_traffic.json")
    if tr:
        md.append(f"| {w} | {tr['frame_hbm_bytes']} | {tr['algorithmic_bytes']} | {tr['frame_over_algorithmic']} | {tr['structural_bytes']} | {tr['frame_over_structural']} |")
md.append("")
if summary:
    md += ["## C2 deep-queue loop under the tracer (`--resubmit`, 4 lanes)", "", "```json", json.dumps(summary, indent=1), "```", ""]
if native:
    md += ["## native dispatch under `rocprofv3 --pmc`", "", "```json", json.dumps(native, indent=1), "```", ""]
sp = load(f"{tag}_split_times_one_gpu_emulation.json")
if sp:
    md += ["## tile split, per-rank kernel times EMULATED ON ONE GPU (`tools/split_times.py`; nothing here ran on more than one GPU)", "",
           "| workload | world, layout | slowest rank kernels us | vertex / geometry / raster per rank | predicted speed-up |", "|---|---|---|---|---|"]
    for wname, wv in sp["workloads"].items():
        for key, e in wv["worlds"].items():
            if key == "1":
                md.append(f"| {wname} | 1 | {e['frame_us']} | " + " / ".join(str(e['ranks'][0][k]) for k in ('vertex', 'geometry', 'raster')) + " | 1.0 |"); continue
            md.append(f"| {wname} | {key} | {e['slowest_rank_kernels_us']} | " + " ; ".join("/".join(str(r[k]) for k in ('vertex', 'geometry', 'raster')) for r in e['ranks']) + f" | {e['speedup_predicted']} |")
    md.append("")
    if any("rank_period_us_2_in_flight" in e for wv in sp["workloads"].values() for e in wv["worlds"].values()):
        md += ["A rank's frame PERIOD in the reference-shaped loop (re-recorded, fenced; its next frame's vertex / geometry kernels run under this frame's raster kernel), same emulation:", "",
               "| workload | world, layout | period per rank us, 2 in flight | 4 in flight | one GPU's period 2 / 4 in flight | predicted speed-up 2 / 4 in flight |", "|---|---|---|---|---|---|"]
        for wname, wv in sp["workloads"].items():
            one = wv["worlds"]["1"]
            for key, e in wv["worlds"].items():
                if "rank_period_us_2_in_flight" not in e: continue
                md.append(f"| {wname} | {key} | " + " ; ".join(str(x) for x in e["rank_period_us_2_in_flight"]) + " | " + " ; ".join(str(x) for x in e["rank_period_us_4_in_flight"]) +
                          f" | {one.get('period_us_2_in_flight')} / {one.get('period_us_4_in_flight')} | {e['speedup_predicted_2_in_flight']} / {e['speedup_predicted_4_in_flight']} |")
        md.append("")
open(os.path.join(dst, f"{tag}_SUMMARY.md"), "w").write("\n".join(md) + "\n")
print("wrote", f"{tag}_SUMMARY.md")
