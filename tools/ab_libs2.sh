#!/bin/bash
# tools/ab_libs2.sh <outdir> <name>... -- embed-only / step time of the batch workloads, default library and variant builds
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out/$OUT
cd $ROOT
for lib in default "$@"; do
  if [ $lib = default ]; then unset TFFT_LIB; else export TFFT_LIB=$ROOT/steganosaurus_amd/variants/libturtlefft_hip_$lib.so; fi
  for wl in 1080p_batch 4k_batch; do
    python3 bench.py --workload $wl --no-cpu-baseline --batched-only --steps 10 --warmup 3 > gpurun_out/$OUT/${wl}_$lib.json 2> gpurun_out/$OUT/${wl}_$lib.err || { tail -3 gpurun_out/$OUT/${wl}_$lib.err; exit 1; }
    python3 - gpurun_out/$OUT/${wl}_$lib.json $wl $lib <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], sys.argv[3], 'MPix/s', d['value'], 'ms', d['ms_per_step'], 'embed_only', d['path']['embed_only']['ms_per_step'])
PY
  done
done
