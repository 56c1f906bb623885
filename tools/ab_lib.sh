#!/bin/bash
# tools/ab_lib.sh <outdir> <variant name> -- the default library against steganosaurus_amd/variants/libturtlefft_hip_<name>.so on the batch
# workloads (value, ms per step, embed-only, a few stages), and the stego bytes of the in-kernel statistics path against the others
OUT=$1; NAME=$2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out/$OUT
cd $ROOT
for lib in default $NAME; do
  if [ $lib = default ]; then unset TFFT_LIB; else export TFFT_LIB=$ROOT/steganosaurus_amd/variants/libturtlefft_hip_$lib.so; fi
  for wl in 1080p_batch 4k_batch 1080p_single; do
    python3 bench.py --workload $wl --no-cpu-baseline --no-others --steps 10 --warmup 3 > gpurun_out/$OUT/${wl}_$lib.json 2> gpurun_out/$OUT/${wl}_$lib.err || { tail -3 gpurun_out/$OUT/${wl}_$lib.err; exit 1; }
    python3 - gpurun_out/$OUT/${wl}_$lib.json $wl $lib <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], sys.argv[3], 'MPix/s', d['value'], 'ms', d['ms_per_step'], 'embed_only', d.get('path', {}).get('embed_only', {}).get('ms_per_step'),
      {k: round(v['ms'], 3) for k, v in d.get('stages', {}).items()})
PY
  done
  python3 tools/cmp_tile.py 1920 1080 231184 | grep -E "tile vs|per image"
  python3 tools/cmp_tile.py 2048 2048 231184 | grep -E "tile vs|per image"
done
