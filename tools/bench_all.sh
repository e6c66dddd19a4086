#!/bin/bash
# all four BASELINE workloads, one JSON summary line each (value, ms/step, raster us, geometry us)
out=${1:-gpurun_out/bench_all.log}
: > "$out"
for w in c2 c3 c4 c5; do
  steps=2000; [ $w = c4 ] && steps=300; [ $w = c5 ] && steps=1000
  python bench.py --workload $w --steps $steps --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$w', round(d['value'],1), d['unit'], 'ms/step', d['ms_per_step'], 'raster_us', r['avg_kernel_us'], 'geom_us', r['geometry_kernel_us'], 'frac', r['frac'])" >> "$out" || exit 1
done
cat "$out"
