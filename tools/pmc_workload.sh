#!/bin/bash
# SQ instruction counters per kernel for one workload (run on the GPU box): pmc_workload.sh c4
w=${1:-c4}
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$w
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES --output-format csv -d $out/a -- python3 bench.py --workload $w --profile-pass-only --steps 20 --warmup 3 > /dev/null 2> $out/a.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/b -- python3 bench.py --workload $w --profile-pass-only --steps 20 --warmup 3 > /dev/null 2> $out/b.err
python3 tools/pmc_summary.py $(find $out -name "*counter_collection.csv")
