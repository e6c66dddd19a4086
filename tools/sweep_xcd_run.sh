# Tile-to-XCD run length of a 1-D raster grid (MIRHI_XCD_RUN) on C5 / C4 / C3: rate and isolated raster time.
mkdir -p gpurun_out/r3b
for w in c5 c4 c3; do
for g in 1 2 4 5 15; do
  v=$(MIRHI_XCD_RUN=$g python bench.py --workload $w --no-extras --no-cpu-baseline --other-workloads '' 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['us_per_frame'], 'raster', d['roofline']['avg_kernel_us'])")
  echo "$w run=$g: $v" | tee -a gpurun_out/r3b/xcd_run.txt
done; done
