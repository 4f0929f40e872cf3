#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 of one of bench.py's shapes -- by default the DRIVER's exact
# command (`python bench.py --gpus 1 --steps 20 --warmup 5`): kernel-trace stats, then PMC passes in their own
# runs (gfx950: 8 SQ slots per pass; FETCH_SIZE and WRITE_SIZE do not fit one TCC pass).
# usage: tools/profile_driver_cmd.sh <tag> [bench args...]      -> gpurun_out/prof_<tag>/
# The profiled runs time the headline region only (--no-legs --no-strict --no-pcie --cpu-sample 0), so that
# the counter sums and kernel statistics belong to (steps + warmup) x batch problems of the benchmarked
# kernels; one more, unprofiled, run of the full command gives the bench line stored beside them.
set -u
TAG=${1:-r03}; shift || true
ARGS=${@:---gpus 1 --steps 20 --warmup 5}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT; echo "$ARGS" > $OUT/args.txt
CLEAN="--cpu-sample 0 --no-strict --no-pcie --no-legs"
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
    local name=$1; shift
    rocprofv3 --pmc "$@" --kernel-trace -d $OUT/$name -o p -- python $REPO/bench.py $ARGS $CLEAN > $OUT/${name}_bench.log 2>&1
}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $REPO/bench.py $ARGS $CLEAN > $OUT/kt_bench.log 2>&1
run pmc_f64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES
run pmc_int SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32
run pmc_sq SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU
run pmc_mem SQ_INSTS_FLAT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run pmc_fetch FETCH_SIZE GRBM_GUI_ACTIVE
run pmc_write WRITE_SIZE
cd $REPO
grep "^{" $OUT/kt_bench.log | tail -1 > $OUT/bench_line.json
if [ -z "${PIK_PROFILE_SKIP_FULL:-}" ]; then  # (tools/gpu/final_lines.sh takes the full lines of the final library anyway)
    python bench.py $ARGS --cpu-sample 0 > $OUT/full_bench.log 2>&1
    grep "^{" $OUT/full_bench.log | tail -1 > $OUT/bench_line_full.json
fi
python tools/read_prof.py $OUT --json $OUT/summary.json > $OUT/summary.txt 2>&1
# (what travels back is capped at 64 MiB: the raw databases and traces stay on the box)
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
