#!/bin/bash
# tools/tile_ab.sh <outdir> -- the statistics inside the last forward column step (default) against their other forms, batch workloads:
# sample step 16, |F|^2 planes (TFFT_STATS_TILE=0), no statistics; then a kernel trace of the default 1080p step
OUT=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out/$OUT
cd $ROOT
run() {  # name workload [flags]   (environment from the caller)
  python3 bench.py --workload $2 --no-cpu-baseline --batched-only --steps 10 --warmup 3 $3 > gpurun_out/$OUT/$2_$1.json 2> gpurun_out/$OUT/$2_$1.err || { tail -3 gpurun_out/$OUT/$2_$1.err; exit 1; }
  python3 - gpurun_out/$OUT/$2_$1.json $2 $1 <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], sys.argv[3], 'MPix/s', d['value'], 'ms', d['ms_per_step'], 'embed_only', d['path']['embed_only']['ms_per_step'])
PY
}
for wl in 1080p_batch 4k_batch; do
  run tile $wl
  TFFT_STATS_TILE_STEP=16 run step16 $wl
  TFFT_STATS_TILE=0 run planes $wl
  run nostats $wl --no-stats
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/$OUT/p1080 -o t -- python3 $ROOT/bench.py --workload 1080p_batch --no-cpu-baseline --batched-only --steps 10 --warmup 3 > $ROOT/gpurun_out/$OUT/b1080_trace.json 2> $ROOT/gpurun_out/$OUT/b1080_trace.err
