#!/bin/bash
# A/B of two builds of libmirhi.so on ONE box, interleaved (boxes differ by +-10 %): tools/ab_bench.sh libA.so libB.so [rounds] [bench args...]
# prints value (Mtris/s), raster / geometry kernel us (isolated), us per frame
A=$1; B=$2; R=${3:-3}; shift 3
for r in $(seq 1 $R); do
  for L in $A $B; do
    MIRHI_LIB_NAME=$L python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); rf=j['roofline']
print('$L', 'round $r', 'value', j['value'], 'us/frame', j['us_per_frame'], 'raster', rf['avg_kernel_us'], 'geometry', rf['geometry_kernel_us'], 'vertex', rf['vertex_kernel_us'])"
  done
done
