#!/bin/bash
# tools/round_profiles.sh <tag> -- everything profiles/ holds for a round, in one gpurun call:
#   the rocprofv3 trace + PMC passes of the default bench (tools/prof.sh), the default bench line itself
#   (with cpu_baseline) and the bench lines of the other BASELINE configurations.
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
bash tools/prof.sh $TAG --steps 10 --warmup 3 || exit 1
python3 tools/prof_summary.py gpurun_out/prof_$TAG > gpurun_out/prof_$TAG/summary.txt || exit 1
python3 tools/prof_traffic.py gpurun_out/prof_$TAG/summary.json 1080p_batch gpurun_out/prof_$TAG/traffic.json || exit 1
cp gpurun_out/prof_$TAG/traffic.json profiles/traffic.json   # on the GPU box; copy gpurun_out/prof_<tag>/traffic.json back by hand afterwards
python3 bench.py > gpurun_out/prof_$TAG/bench_default.json 2> gpurun_out/prof_$TAG/bench_default.err || exit 1
for wl in 1080p_single 4k_single 4k_batch 8192_single 512_single; do
  python3 bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_$TAG/other_$wl.json 2> gpurun_out/prof_$TAG/other_$wl.err || exit 1
done
ls gpurun_out/prof_$TAG | wc -l
