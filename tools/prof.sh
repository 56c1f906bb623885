#!/bin/bash
# tools/prof.sh <tag> [bench args...] -- rocprofv3 kernel trace + PMC passes of bench.py (run on the GPU box).
# Counters go in separate runs with --kernel-trace only (never with sys/hip/hsa traces); FETCH_SIZE and WRITE_SIZE do not fit
# one pass (MI355X_MICROARCH.md, rocprofv3 PMC slots).
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --batched-only"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- $B "$@" > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT -o pmc_sq -- $B "$@" > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_BUSY_CYCLES --output-format csv -d $OUT -o pmc_sq2 -- $B "$@" > /dev/null 2> $OUT/pmc_sq2.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o pmc_fetch -- $B "$@" > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o pmc_write -- $B "$@" > /dev/null 2> $OUT/pmc_write.err
python3 $ROOT/tools/prof_summary.py $OUT > $OUT/summary.txt 2> $OUT/summary.err
ls $OUT | wc -l
