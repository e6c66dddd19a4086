#!/bin/bash
# A/B of two environments on ONE box, interleaved: tools/ab_env.sh "MIRHI_X=0" "MIRHI_X=1" rounds [bench args...]
A=$1; B=$2; R=${3:-3}; shift 3
for r in $(seq 1 $R); do
  for E in "$A" "$B"; do
    env $E python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); rf=j['roofline']
print('$E', 'round $r', 'value', j['value'], 'us/frame', j['us_per_frame'], 'raster', rf['avg_kernel_us'], 'geometry', rf['geometry_kernel_us'], 'vertex', rf['vertex_kernel_us'])"
  done
done
