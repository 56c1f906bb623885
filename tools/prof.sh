#!/bin/bash
# tools/prof.sh <tag> [bench args...] -- rocprofv3 kernel trace + three PMC passes of bench.py (run on the GPU box).
# Counters go in separate runs with --kernel-trace only (never with sys/hip/hsa traces).
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python3 $ROOT/bench.py --no-cpu-baseline --batched-only "$@" > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT -o pmc_sq -- python3 $ROOT/bench.py --no-cpu-baseline --batched-only "$@" > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o pmc_fetch -- python3 $ROOT/bench.py --no-cpu-baseline --batched-only "$@" > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o pmc_write -- python3 $ROOT/bench.py --no-cpu-baseline --batched-only "$@" > /dev/null 2> $OUT/pmc_write.err
ls -la $OUT
