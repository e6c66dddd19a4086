#!/bin/bash
# A/B of builds of libmirhi.so on ONE box, interleaved, natively dispatched, with the frame-loop extras: tools/ab_full.sh "libA.so libB.so ..." [rounds] [bench args...]
# (variants: python renderer-rs_amd/build.py --variant NAME -DX=1 -> libmirhi_NAME.so + its code object)
LIBS=$1; R=${2:-3}; shift 2
for r in $(seq 1 $R); do
  for L in $LIBS; do
    MIRHI_LIB_NAME=$L python bench.py --no-cpu-baseline --other-workloads c3 "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); rf=j['roofline']; w=j.get('workloads',{}).get('c3',{})
print('%-20s' % '$L', 'round $r', 'c2', j['value'], 'rerec4', j['rerecorded_submit']['value'], 'rerec2', j['frames_in_flight_2']['rerecorded']['value'], 'lanes2', j['frames_in_flight_2']['resubmitted']['value'],
      '| c3', w.get('value'), 'rerec4', (w.get('rerecorded_submit') or {}).get('value'), '| iso raster', rf['avg_kernel_us'], 'geometry', rf['geometry_kernel_us'], flush=True)"
  done
done
