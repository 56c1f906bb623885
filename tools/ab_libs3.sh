#!/bin/bash
# tools/ab_libs3.sh <outdir> <name>... -- step / embed-only time and the row-side stage times, default library and variant builds, three workloads
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out/$OUT
cd $ROOT
for rep in 1 2; do
for lib in default "$@"; do
  if [ $lib = default ]; then unset TFFT_LIB; else export TFFT_LIB=$ROOT/steganosaurus_amd/variants/libturtlefft_hip_$lib.so; fi
  for wl in 1080p_batch 4k_batch 8192_single; do
    python3 bench.py --workload $wl --no-cpu-baseline --no-others --steps 10 --warmup 3 > gpurun_out/$OUT/${wl}_$lib.json 2> gpurun_out/$OUT/${wl}_$lib.err || { tail -3 gpurun_out/$OUT/${wl}_$lib.err; exit 1; }
    python3 - gpurun_out/$OUT/${wl}_$lib.json $wl $lib <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], sys.argv[3], 'MPix/s', d['value'], 'ms', d['ms_per_step'], 'embed_only', d.get('path',{}).get('embed_only',{}).get('ms_per_step'), {k: round(v['ms'],3) for k,v in d['stages'].items() if k.startswith('rows')})
PY
  done
done
done
