#!/bin/bash
# SQ counters per kernel for one workload (run on the GPU box): pmc_workload.sh c4 [extra bench args]
# Three separate passes (instruction mix; memory instructions + wave cycles; issue-cycle accounting), each its own run.
w=${1:-c4}; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$w
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
args="--workload $w --profile-pass-only --steps 2 --warmup 1 --frames-per-step 16 --no-cpu-baseline $@"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES --output-format csv -d $out/a -- python3 bench.py $args > /dev/null 2> $out/a.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/b -- python3 bench.py $args > /dev/null 2> $out/b.err
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $out/c -- python3 bench.py $args > /dev/null 2> $out/c.err
python3 tools/pmc_summary.py $(find $out -name "*counter_collection.csv")
tail -2 $out/c.err
