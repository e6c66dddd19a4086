#!/bin/bash
# VALU / SALU wave-instructions of the raster kernel per stage (diagnostic build): stage_counts.sh <workload>   (run on the GPU box)
w=${1:-c5}
out=$GRAFT_REPO_ROOT/gpurun_out/stages_$w
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for st in 1 2 3 0; do
  MIRHI_STAGE=$st rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $out/s$st -- python3 tools/stage_counts.py 6 $w > /dev/null 2> $out/s$st.err
  python3 - $out/s$st $st <<'PY'
import sys, glob, csv, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "raster_kernel" in k: rows["raster"][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("stage", sys.argv[2], {c: round(sum(v) / len(v)) for c, v in sorted(rows["raster"].items())})
PY
done
