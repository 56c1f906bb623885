#!/bin/bash
# tools/nostats.sh <outdir> -- what the statistics' traffic costs the embed span: the batch workloads with and without capacities
OUT=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out/$OUT
cd $ROOT
for wl in 1080p_batch 4k_batch; do
  for f in stats nostats; do
    flag=""; [ $f = nostats ] && flag="--no-stats"
    python3 bench.py --workload $wl --no-cpu-baseline --batched-only --steps 10 --warmup 3 $flag > gpurun_out/$OUT/${wl}_$f.json 2> gpurun_out/$OUT/${wl}_$f.err || { tail -3 gpurun_out/$OUT/${wl}_$f.err; exit 1; }
    python3 - gpurun_out/$OUT/${wl}_$f.json $wl $f <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], sys.argv[3], 'MPix/s', d['value'], 'ms', d['ms_per_step'], 'embed_only', d['path']['embed_only']['ms_per_step'],
      {k: round(v['ms'], 3) for k, v in d['stages'].items() if k in ('cols_fwd_b', 'medians', 'cols_inv_a', 'rows_inv')})
PY
  done
done
