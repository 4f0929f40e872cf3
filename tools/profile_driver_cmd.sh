#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 of the DRIVER's exact bench command
# (`python bench.py --gpus 1 --steps 20 --warmup 5`): kernel-trace stats, then PMC passes in their
# own runs (gfx950: 8 SQ slots per pass; FETCH_SIZE and WRITE_SIZE do not fit one TCC pass).
# usage: tools/profile_driver_cmd.sh <tag> [bench args...]      -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r03}; shift || true
ARGS=${@:---gpus 1 --steps 20 --warmup 5}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT; echo "$ARGS" > $OUT/args.txt
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
    local name=$1; shift
    # (counter passes: the measured leg only -- no CPU sample, no strict-build / host-pointer legs, so
    #  that the sums belong to (steps + warmup) x batch problems of the benchmarked kernels)
    rocprofv3 --pmc "$@" --kernel-trace -d $OUT/$name -o p -- python $REPO/bench.py $ARGS --cpu-sample 0 --no-strict --no-pcie > $OUT/${name}_bench.log 2>&1
}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $REPO/bench.py $ARGS > $OUT/kt_bench.log 2>&1
run pmc_f64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES
run pmc_int SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU
run pmc_sq SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU
run pmc_fetch FETCH_SIZE GRBM_GUI_ACTIVE
run pmc_write WRITE_SIZE
grep "^{" $OUT/kt_bench.log | tail -1 > $OUT/bench_line.json
cd $REPO && python tools/read_prof.py $OUT --json $OUT/summary.json > $OUT/summary.txt 2>&1
