#!/bin/bash
# tools/ab_sample.sh <outdir> -- the statistics' sample pass: default (every 4th row group of twice the tiles) against the forced tile steps
OUT=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out/$OUT
cd $ROOT
run() {
  python3 bench.py --workload $2 --no-cpu-baseline --batched-only --steps 10 --warmup 3 > gpurun_out/$OUT/$2_$1.json 2> gpurun_out/$OUT/$2_$1.err || { tail -3 gpurun_out/$OUT/$2_$1.err; exit 1; }
  python3 - gpurun_out/$OUT/$2_$1.json $2 $1 <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], sys.argv[3], 'MPix/s', d['value'], 'ms', d['ms_per_step'], 'embed_only', d['path']['embed_only']['ms_per_step'], 'medians', d['stages']['medians']['ms'])
PY
}
for rep in 1 2; do
  run default 1080p_batch
  TFFT_STATS_TILE_STEP=8 run step8 1080p_batch
  run default 4k_batch
  TFFT_STATS_TILE_STEP=16 run step16 4k_batch
done
