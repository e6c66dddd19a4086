#!/bin/bash
# the mesh workloads in one go: dancer (isolated kernel times) + c3/c4/c5 bench lines; run on the GPU box
python tools/dancer_times.py || exit 1
for w in c3 c4 c5; do
  python bench.py --workload $w --no-cpu-baseline --steps 500 > gpurun_out/m_$w.json 2> gpurun_out/m_$w.err || exit 1
  python -c "import json; d=json.load(open('gpurun_out/m_$w.json')); print('$w', d['value'], 'raster', d['roofline']['avg_kernel_us'], 'geometry', d['roofline']['geometry_kernel_us'])"
done
