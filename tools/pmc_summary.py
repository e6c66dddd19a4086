"""Mean per-launch counter values per kernel from a rocprofv3 --pmc counter_collection CSV.
usage: pmc_summary.py <counter_collection.csv> [more.csv ...]"""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        name = r.get("Kernel_Name") or r.get("kernel_name")
        short = "raster" if "raster_kernel" in name else ("geometry" if "geometry_kernel" in name else ("vertex" if "vertex_kernel" in name else None))
        if short is None: continue
        acc[short][r.get("Counter_Name") or r.get("counter_name")].append(float(r.get("Counter_Value") or r.get("counter_value")))
for k, d in acc.items():
    for c, v in sorted(d.items()):
        v = v[len(v) // 5:]
        print(f"{k:9s} {c:28s} mean {sum(v) / len(v):14.1f}  n={len(v)}")
