#!/bin/bash
# A/B of the triangle-parallel threshold (MIRHI_TP_MAX_AREA, 0 = off) on the mesh workloads; run on the GPU box.
for a in 0 64 96 128; do
  echo "== MIRHI_TP_MAX_AREA=$a"; export MIRHI_TP_MAX_AREA=$a
  python tools/dancer_times.py | head -2
  for w in c3 c4 c5; do python bench.py --workload $w --no-cpu-baseline --steps 300 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"$w\", d[\"value\"], \"raster\", d[\"roofline\"][\"avg_kernel_us\"], \"geometry\", d[\"roofline\"][\"geometry_kernel_us\"])"; done
done
