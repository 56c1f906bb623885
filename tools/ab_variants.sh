#!/bin/bash
# tools/ab_variants.sh <outdir> <workload> <name>... -- embed-only time + the timeline's dominant kernels for the default library and each variant build
OUT=$1; WL=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out/$OUT
cd /tmp && export TMPDIR=/tmp
for lib in default "$@"; do
  if [ $lib = default ]; then unset TFFT_LIB; else export TFFT_LIB=$ROOT/steganosaurus_amd/variants/libturtlefft_hip_$lib.so; fi
  rocprofv3 --kernel-trace -d $ROOT/gpurun_out/$OUT/$lib -o t -- python3 $ROOT/bench.py --workload $WL --no-cpu-baseline --batched-only --steps 6 --warmup 2 > $ROOT/gpurun_out/$OUT/$lib.json 2> $ROOT/gpurun_out/$OUT/$lib.err || { tail -3 $ROOT/gpurun_out/$OUT/$lib.err; }
  echo "== $lib"
  python3 $ROOT/tools/step_timeline.py $ROOT/gpurun_out/$OUT/$lib/t_results.db k_gather_bits 4 | awk '$2 > 60'
done
