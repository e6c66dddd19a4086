"""Turns gpurun_out/prof_<tag>/ (tools/collect_profiles.sh) into the per-round files under profiles/.
usage: make_profile_summaries.py <tag>"""
import collections, csv, glob, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")

def one(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    if not hits:
        raise SystemExit("missing " + pattern)
    return max(hits, key=os.path.getmtime)      # gpurun merges into gpurun_out/: earlier runs' files may still be there

shutil.copy(one("stats/**/*kernel_stats.csv"), os.path.join(dst, f"{tag}_c2_rocprofv3_kernel_stats.csv"))
for name in ("bench_profile_pass", "bench_default"):
    line = [l for l in open(os.path.join(src, name + ".json")) if l.startswith("{")][-1]
    json.dump(json.loads(line), open(os.path.join(dst, f"{tag}_c2_{name}.json"), "w"), indent=1)
for w in ("c3", "c4", "c5"):
    line = [l for l in open(os.path.join(src, f"bench_{w}.json")) if l.startswith("{")][-1]
    json.dump(json.loads(line), open(os.path.join(dst, f"{tag}_{w}_bench.json"), "w"), indent=1)

def means(pass_dir):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(one(pass_dir + "/**/*counter_collection.csv"))):
        acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v)) for k, v in acc.items()}

rows = []
per_kernel = collections.defaultdict(dict)
for pass_dir in ("pmc_sq", "pmc_fetch", "pmc_write"):
    for (kern, ctr), (n, mean) in sorted(means(pass_dir).items()):
        if "mirhi::" not in kern:
            continue
        rows.append((pass_dir, kern, ctr, n, round(mean, 1)))
        short = "raster_kernel" if "raster_kernel" in kern else ("geometry_kernel" if "geometry_kernel" in kern else "vertex_kernel")
        per_kernel[short][ctr] = mean
with open(os.path.join(dst, f"{tag}_c2_rocprofv3_pmc_summary.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["pass", "kernel", "counter", "launches", "mean_per_launch"])
    w.writerows(rows)
traffic = {"workload": "c2",
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), python3 bench.py --profile-pass-only --steps 50",
           "unit_note": "FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B for wide "
                        "coalesced streams (MI355X_MICROARCH.md, HBM section), so the read side is doubled; that correction is calibrated "
                        "for 16 B/lane streaming reads and is an upper bound for the 48-B record gathers here"}
for k, d in per_kernel.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        traffic[k] = {"fetch_size_kb": d["FETCH_SIZE"], "write_size_kb": d["WRITE_SIZE"],
                      "hbm_bytes_per_launch": int(round((2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024))}
json.dump(traffic, open(os.path.join(dst, f"{tag}_c2_hbm_traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
for r in rows:
    print(*r)
