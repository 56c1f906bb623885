#!/bin/bash
# tools/round_profiles.sh <tag> -- everything profiles/ holds for a round, in one gpurun call: rocprofv3 kernel trace + PMC passes
# (tools/prof.sh) of the default workload, of the 4K batch and of 8192^2 (BASELINE configs[4] asks for that capture), the traffic
# table bench.py replays, and the default bench line itself (with cpu_baseline, parity and other_workloads).
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for spec in "1080p_batch:--steps 10 --warmup 3" "4k_batch:--workload 4k_batch --steps 5 --warmup 2" "8192_single:--workload 8192_single --steps 5 --warmup 2"; do
  wl=${spec%%:*}; args=${spec#*:}
  bash tools/prof.sh ${TAG}_$wl $args || exit 1
  python3 tools/prof_traffic.py gpurun_out/prof_${TAG}_$wl/summary.json $wl "profiles/$TAG/${wl}_pmc_summary.json" gpurun_out/traffic_$TAG.json || exit 1
done
cp gpurun_out/traffic_$TAG.json profiles/traffic.json     # on the GPU box, so that the bench line below replays this round's counters
python3 bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err || { tail -5 gpurun_out/bench_$TAG.err; exit 1; }
for wl in 1080p_single 4k_single 512_single 2048_batch; do
  python3 bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-others > gpurun_out/bench_${TAG}_$wl.json 2> gpurun_out/bench_${TAG}_$wl.err || exit 1
done
ls gpurun_out | grep $TAG
