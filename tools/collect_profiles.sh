#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's C2 line into gpurun_out/prof_<tag>/ (run on the GPU box):
#   kernel-trace --stats of the isolated timing pass, three separate PMC passes (SQ instruction counters, FETCH_SIZE,
#   WRITE_SIZE), the default bench line, and the other workloads' bench lines.  tools/make_profile_summaries.py turns
#   the raw CSVs into the files committed under profiles/.
set -e
tag=${1:-r01}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --profile-pass-only --steps 1000 --warmup 20 > $out/bench_profile_pass.json 2> $out/stats.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES --output-format csv -d $out/pmc_sq -- python3 bench.py --profile-pass-only --steps 50 --warmup 5 > /dev/null 2> $out/pmc_sq.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py --profile-pass-only --steps 50 --warmup 5 > /dev/null 2> $out/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --profile-pass-only --steps 50 --warmup 5 > /dev/null 2> $out/pmc_write.err
python3 bench.py --steps 3000 --warmup 100 > $out/bench_default.json 2> $out/bench_default.err
for w in c3 c4 c5; do python3 bench.py --workload $w --steps 500 --no-cpu-baseline > $out/bench_$w.json 2>/dev/null; done
find $out -name "*.csv" | head -20
tail -c 600 $out/bench_default.json
