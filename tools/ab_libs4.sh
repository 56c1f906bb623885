#!/bin/bash
# tools/ab_libs4.sh <outdir> <workload> <name>... -- embed-only / step time of one workload, default library and variant builds, two rounds
OUT=$1; WL=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out/$OUT
cd $ROOT
for rep in 1 2; do
for lib in default "$@"; do
  if [ $lib = default ]; then unset TFFT_LIB; else export TFFT_LIB=$ROOT/steganosaurus_amd/variants/libturtlefft_hip_$lib.so; fi
  python3 bench.py --workload $WL --no-cpu-baseline --no-others --steps 10 --warmup 3 > gpurun_out/$OUT/${WL}_$lib.json 2> gpurun_out/$OUT/${WL}_$lib.err || { tail -3 gpurun_out/$OUT/${WL}_$lib.err; exit 1; }
  python3 - gpurun_out/$OUT/${WL}_$lib.json $WL $lib <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], sys.argv[3], 'MPix/s', d['value'], 'ms', d['ms_per_step'], 'embed_only', d.get('path',{}).get('embed_only',{}).get('ms_per_step'), 'final fwd', d['stages'].get('cols_fwd_b',{}).get('ms'))
PY
done
done
