#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + PMC passes of bench.py.
# usage: tools/profile_bench.sh <tag> [bench args...]
# PMC passes are separate runs (gfx950: SQ 8 slots, TCC 4; FETCH_SIZE and WRITE_SIZE do not fit one pass)
set -u
TAG=${1:-r01}; shift || true
ARGS=${@:---steps 16 --warmup 2 --streams 1 --cpu-sample 0}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $REPO/bench.py $ARGS > $OUT/kt_bench.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/pmc1 -o p -- python $REPO/bench.py $ARGS > $OUT/pmc1_bench.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_INSTS_LDS SQ_INST_CYCLES_SALU SQ_IFETCH SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT --kernel-trace -d $OUT/pmc2 -o p -- python $REPO/bench.py $ARGS > $OUT/pmc2_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc3 -o p -- python $REPO/bench.py $ARGS > $OUT/pmc3_bench.log 2>&1
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc4 -o p -- python $REPO/bench.py $ARGS > $OUT/pmc4_bench.log 2>&1
grep "^{" $OUT/kt_bench.log | tail -1 > $OUT/bench_line.json
