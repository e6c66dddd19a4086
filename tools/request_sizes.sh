#!/bin/bash
# Exact HBM-side bytes of each kernel from the L2's request-size counters (gfx950: 32 / 64 / 128-byte read requests, 32-byte units to DRAM), as a check of the
# FETCH_SIZE figure (which tallies a 128-byte request at 64: MI355X_MICROARCH.md, HBM).  usage: tools/request_sizes.sh [workloads...] -> gpurun_out/reqsize/<w>.txt
set -e
out=$GRAFT_REPO_ROOT/gpurun_out/reqsize; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for w in ${@:-c2 c3 c4 c5}; do
  rm -rf $out/$w.a $out/$w.b
  rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $out/$w.a -- python3 bench.py --workload $w --profile-pass-only --no-cpu-baseline > /dev/null 2> $out/$w.a.err
  rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum --output-format csv -d $out/$w.b -- python3 bench.py --workload $w --profile-pass-only --no-cpu-baseline > /dev/null 2> $out/$w.b.err
  python3 - $out $w <<'PY' | tee $out/$w.txt
import csv, glob, sys, collections, json
out, w = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(f"{out}/{w}.a/**/*counter_collection.csv", recursive=True) + glob.glob(f"{out}/{w}.b/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        if "mirhi" not in k: continue
        k = "vertex" if "vertex" in k else "geometry" if "geometry" in k else "raster" if "raster" in k else k
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    n32, n64, n128, tot = m.get("TCC_EA0_RDREQ_32B_sum", 0), m.get("TCC_EA0_RDREQ_64B_sum", 0), m.get("TCC_EA0_RDREQ_128B_sum", 0), m.get("TCC_EA0_RDREQ_sum", 0)
    res[k] = {"launches": len(next(iter(c.values()))), "read_requests": round(tot), "read_32B": round(n32), "read_64B": round(n64), "read_128B": round(n128),
              "read_bytes_by_size": round(32 * n32 + 64 * n64 + 128 * n128), "fetch_size_formula_bytes": round(32 * n32 + 64 * (tot - n32)),
              "read_bytes_dram_32B_units": round(32 * m.get("TCC_EA0_RDREQ_DRAM_32B_sum", 0)),
              "write_requests": round(m.get("TCC_EA0_WRREQ_sum", 0)), "write_64B": round(m.get("TCC_EA0_WRREQ_64B_sum", 0)),
              "write_bytes_dram_32B_units": round(32 * m.get("TCC_EA0_WRREQ_WRITE_DRAM_32B_sum", 0))}
print(json.dumps({"workload": w, "per_launch_means": res}, indent=1))
PY
  rm -rf $out/$w.a $out/$w.b
done
