# instruction counts of the raster kernel cut after each stage (stamps build, HIP launches): see tools/stage_counts.py
set -e
out=$GRAFT_REPO_ROOT/gpurun_out/r3b/stages; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export MIRHI_NATIVE_DISPATCH=0
for w in ${1:-c2}; do
for s in 1 2 3 0; do
  MIRHI_STAGE=$s rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES --output-format csv -d $out/$w$s -- python3 tools/stage_counts.py 6 $w > /dev/null 2> $out/$w$s.err
  python3 - $out/$w$s "$w stage $s" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "raster" in r["Kernel_Name"]: acc[r["Kernel_Name"].split("(")[0][-34:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items(): print(sys.argv[2], k, {n: round(sum(v[-3:]) / len(v[-3:])) for n, v in sorted(c.items())})
PY
done; done
rm -rf $out/*/
