#!/bin/bash
# tools/trace_wl.sh <outdir> <workload> [ENV=value ...] -- kernel trace of the batched steps of one workload; prints one embed+extract step's launches
OUT=$1; WL=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out/$OUT
for kv in "$@"; do export "$kv"; done
TAG=$(echo "$WL $*" | tr ' =' '__')
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $ROOT/gpurun_out/$OUT/$TAG -o t -- python3 $ROOT/bench.py --workload $WL --no-cpu-baseline --batched-only --steps 6 --warmup 2 > $ROOT/gpurun_out/$OUT/$TAG.json 2> $ROOT/gpurun_out/$OUT/$TAG.err || { tail -3 $ROOT/gpurun_out/$OUT/$TAG.err; exit 1; }
echo "== $TAG"
python3 $ROOT/tools/step_timeline.py $ROOT/gpurun_out/$OUT/$TAG/t_results.db k_gather_bits 4
