#!/bin/bash
# tools/abenv.sh <outdir> <workload> <ENVVAR> <values...> -- A/B of one environment knob of the shipped library on ONE box
OUT=$1; WL=$2; VAR=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out/$OUT
cd $ROOT
for v in "$@"; do
  export $VAR=$v
  timeout -k 10 300 python3 bench.py --workload $WL --no-cpu-baseline --batched-only --steps 10 --warmup 3 > gpurun_out/$OUT/${WL}_${VAR}_$v.json 2> gpurun_out/$OUT/${WL}_${VAR}_$v.err || { echo "FAILED $WL $v"; tail -3 gpurun_out/$OUT/${WL}_${VAR}_$v.err; exit 1; }
  python3 - gpurun_out/$OUT/${WL}_${VAR}_$v.json $WL $VAR=$v <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
st={k:round(v['ms'],3) for k,v in d['stages'].items()}
print(sys.argv[2], sys.argv[3], 'MPix/s', d['value'], 'ms', d['ms_per_step'], 'embed_only', d['path']['embed_only']['ms_per_step'], st, 'ber', round(d['check']['roundtrip_ber'],4))
PY
done
