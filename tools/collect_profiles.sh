#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's lines into gpurun_out/prof_<tag>/ (run on the GPU box), in two parts of under twenty minutes each:
#   tools/collect_profiles.sh <tag> a     C2, C3: the passes below; the traced deep-queue loop; native dispatch under --pmc; the driver's bench line
#   tools/collect_profiles.sh <tag> b     C4, C5: the passes below; the bench lines of C3 / C4 / C5; the tile-split emulation
#   stats_*     rocprofv3 --kernel-trace --stats of the isolated pass (bench.py --profile-pass-only: one frame on the GPU at a time,
#               every dispatch of the process isolated) -- the tracer's average durations must agree with roofline.avg_kernel_us
#   pmc_*       separate PMC passes of the same command: SQ instruction mix, FETCH_SIZE, WRITE_SIZE, read requests by size (never combined with a trace domain
#               other than --kernel-trace)
#   trace/      rocprofv3 --kernel-trace of the deep-queue loop (--resubmit, 4 queue lanes in flight): per-dispatch begin / end
#   pmc_native  hand-written AQL packets under counter collection (MIRHI_NATIVE_DISPATCH=2)
#   bench_*     plain bench lines (no tool attached): the driver's command, and the other BASELINE configs
# Each part runs tools/make_profile_summaries.py for its share into gpurun_out/prof_<tag>/summary (the raw traces are too big to travel back); at home the
# summaries are copied into profiles/ and `tools/make_profile_summaries.py <tag> profiles --summary-only` writes profiles/<tag>_SUMMARY.md.
set -e
tag=${1:-r04}
part=${2:-a}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out/summary
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
if [ "$part" = a ]; then loads="c2 c3"; else loads="c4 c5"; fi
for w in $loads; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$w -- python3 bench.py --workload $w --profile-pass-only --no-cpu-baseline > $out/bench_profile_pass_$w.json 2> $out/stats_$w.err
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES --output-format csv -d $out/pmc_sq_$w -- python3 bench.py --workload $w --profile-pass-only --no-cpu-baseline > /dev/null 2> $out/pmc_sq_$w.err
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_$w -- python3 bench.py --workload $w --profile-pass-only --no-cpu-baseline > /dev/null 2> $out/pmc_fetch_$w.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_$w -- python3 bench.py --workload $w --profile-pass-only --no-cpu-baseline > /dev/null 2> $out/pmc_write_$w.err
  # the L2's read requests by size and in 32-byte units to DRAM (gfx950 counters): the exact byte count FETCH_SIZE's doubling is checked against
  rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_DRAM_32B_sum --output-format csv -d $out/pmc_rdreq_$w -- python3 bench.py --workload $w --profile-pass-only --no-cpu-baseline > /dev/null 2> $out/pmc_rdreq_$w.err
  echo "collected $w"
done
if [ "$part" = a ]; then
  # the deep-queue loop (command buffers recorded once, resubmitted on four lanes) under the tracer: overlap of consecutive raster kernels
  rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline --no-extras --resubmit > $out/bench_traced.json 2> $out/trace.err
  # native dispatch (hand-written AQL packets) under counter collection: the queue the library gets is a tool's proxy (mirhi_api.hip, native_queue_is_proxy); with
  # MIRHI_NATIVE_DISPATCH=2 it dispatches natively all the same -- every wait bounded (MIRHI_NATIVE_TIMEOUT_MS) -- and the frames' kernels show up on a queue of their own
  MIRHI_NATIVE_DISPATCH=2 MIRHI_NATIVE_TIMEOUT_MS=5000 timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_WAVES --output-format csv -d $out/pmc_native -- python3 bench.py --workload c2 --profile-pass-only --no-cpu-baseline > $out/bench_pmc_native.json 2> $out/pmc_native.err || echo "pmc_native run failed: rc $?"
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --timeline-out $out/timeline_in_flight.json > $out/bench_default.json 2> $out/bench_default.err
  python3 tools/make_profile_summaries.py $tag $out/summary --workloads=c2,c3 > $out/summary_a.log 2>&1 || tail -5 $out/summary_a.log
else
  for w in c3 c4 c5; do python3 bench.py --workload $w --cpu-seconds 8 > $out/bench_$w.json 2>/dev/null; echo "bench $w done"; done
  # the tile split, every rank of world 2 / 4 / 8 in turn on this one GPU, both layouts (an emulation: tools/split_times.py)
  python3 tools/split_times.py --json $out/summary/${tag}_split_times_one_gpu_emulation.json > $out/summary/${tag}_split_times_one_gpu_emulation.txt 2>&1 || tail -3 $out/summary/${tag}_split_times_one_gpu_emulation.txt
  python3 tools/make_profile_summaries.py $tag $out/summary --workloads=c4,c5 --no-trace > $out/summary_b.log 2>&1 || tail -5 $out/summary_b.log
fi
# the bulky raw files are dropped (the summaries were made from them above)
find $out -name "*kernel_trace.csv" -size +2M -delete
find $out -name "*.db" -delete
du -sh $out
ls $out/summary | head -40
