set -e
out=$GRAFT_REPO_ROOT/gpurun_out/r3b/inst; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for s in empty c2 c2half c2small; do
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES --output-format csv -d $out/$s -- python3 tools/inst_probe.py $s > /dev/null 2> $out/$s.err
  python3 - $out/$s $s <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    print(sys.argv[2], k, {n: round(sum(v[-3:]) / len(v[-3:])) for n, v in c.items()})
PY
done
find $out -name "*.db" -delete
