#!/bin/bash
# tools/collect_profiles.sh <tag> -- copy the judged summaries of tools/round_profiles.sh <tag> from gpurun_out/ (scratch) into profiles/<tag>/
TAG=$1
mkdir -p profiles/$TAG
for wl in 1080p_batch 4k_batch 8192_single; do
  d=gpurun_out/prof_${TAG}_$wl
  cp $d/trace_kernel_stats.csv profiles/$TAG/${wl}_kernel_stats.csv
  cp $d/summary.json profiles/$TAG/${wl}_pmc_summary.json
  cp $d/bench_trace.json profiles/$TAG/${wl}_bench_under_rocprof.json
done
cp gpurun_out/traffic_$TAG.json profiles/traffic.json
cp gpurun_out/bench_$TAG.json profiles/$TAG/default_bench.json
for wl in 1080p_single 4k_single 512_single 2048_batch; do cp gpurun_out/bench_${TAG}_$wl.json profiles/$TAG/other_${wl}_bench.json; done
ls -la profiles/$TAG
