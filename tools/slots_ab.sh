#!/bin/bash
# tools/slots_ab.sh <outdir> <workload> <slots...> -- images per launch: does a chunk whose intermediates fit the 256 MB Infinity Cache pay?
OUT=$1; WL=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out/$OUT
cd $ROOT
for sl in "$@"; do
  python3 bench.py --workload $WL --slots $sl --no-cpu-baseline --no-others --stage-reps 3 --steps 10 --warmup 3 > gpurun_out/$OUT/${WL}_s$sl.json 2> gpurun_out/$OUT/${WL}_s$sl.err || { tail -3 gpurun_out/$OUT/${WL}_s$sl.err; exit 1; }
  python3 - gpurun_out/$OUT/${WL}_s$sl.json $WL $sl <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], 'slots', sys.argv[3], 'MPix/s', d['value'], 'ms', d['ms_per_step'], 'embed_only', d['path']['embed_only']['ms_per_step'])
PY
done
