cd $GRAFT_REPO_ROOT
for wl in 1080p_single 4k_single 512_single; do
python3 bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-others > gpurun_out/single_$wl.json 2> gpurun_out/single_$wl.err || { tail -3 gpurun_out/single_$wl.err; exit 1; }
python3 -c "
import json,sys
d=json.loads(open('gpurun_out/single_$wl.json').read().strip().splitlines()[-1]); print('$wl', d['value'], d['ms_per_step'], d['path']['embed_only']['ms_per_step'], d['check'].get('payloads_recovered'))"
done
