#!/bin/bash
# tools/ab.sh <outdir> <workloads, comma separated> <variant names...> -- A/B of device-library builds on ONE box.
# "main" = the shipped library; other names = steganosaurus_amd/variants/libturtlefft_hip_<name>.so (csrc/Makefile `variant`).
OUT=$1; WLS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out/$OUT
cd $ROOT
for wl in ${WLS//,/ }; do
  for v in "$@"; do
    if [ "$v" = main ]; then unset TFFT_LIB; else export TFFT_LIB=$ROOT/steganosaurus_amd/variants/libturtlefft_hip_$v.so; fi
    timeout -k 10 300 python3 bench.py --workload $wl --no-cpu-baseline --batched-only --steps 10 --warmup 3 > gpurun_out/$OUT/${wl}_$v.json 2> gpurun_out/$OUT/${wl}_$v.err || { echo "FAILED $wl $v"; tail -3 gpurun_out/$OUT/${wl}_$v.err; exit 1; }
    python3 - gpurun_out/$OUT/${wl}_$v.json $wl $v <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
st={k:round(v['ms'],3) for k,v in d['stages'].items()}
print(sys.argv[2], sys.argv[3], 'MPix/s', d['value'], 'ms', d['ms_per_step'], 'embed_only', d['path']['embed_only']['ms_per_step'], 'frac_e', d['path']['embed_only']['frac_of_peak'], st, 'ber', round(d['check']['roundtrip_ber'],4))
PY
  done
done
