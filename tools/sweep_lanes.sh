# Deep-queue rate by number of queue lanes (= frames in flight) and raster variant, C3 and C2: more than four lanes are slower (the command processor has four pipes).
mkdir -p gpurun_out/r3b
for w in c3 c2; do
for f in 2 3 4 6 8; do
for wide in auto 0 8; do
  if [ $wide = auto ]; then unset MIRHI_RASTER_WIDE; else export MIRHI_RASTER_WIDE=$wide; fi
  v=$(python bench.py --workload $w --frames-in-flight $f --no-extras --no-cpu-baseline --other-workloads '' 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['us_per_frame'])")
  echo "$w fif=$f wide=$wide: $v" | tee -a gpurun_out/r3b/sweep.txt
done; done; done
