#!/bin/bash
# One wavefront's generation, counted and timed (review item "the floor proven from cycles"): the long runners of
# config 2 alone on the chip at a fixed number of lanes per elite (tools/gpu/long_runners.py) under rocprofv3 --
# kernel trace for the duration, three PMC passes for the instructions by class, the waits and the wave cycles of
# the measured kernel.  usage: tools/gpu/critical_path.sh <tag> <fast|exact> <lanes>...   -> gpurun_out/<tag>/
set -u
TAG=$1; FL=$2; shift 2
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  D=$OUT/${FL}_lpe$L; mkdir -p $D
  python $REPO/tools/gpu/long_runners.py $FL $L > $D/plain.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/kt -o kt -- python $REPO/tools/gpu/long_runners.py $FL $L > $D/kt.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace -d $D/pmc_f64 -o p -- python $REPO/tools/gpu/long_runners.py $FL $L > $D/pmc_f64.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 --kernel-trace -d $D/pmc_int -o p -- python $REPO/tools/gpu/long_runners.py $FL $L > $D/pmc_int.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_FLAT --kernel-trace -d $D/pmc_sq -o p -- python $REPO/tools/gpu/long_runners.py $FL $L > $D/pmc_sq.log 2>&1
done
cd $REPO
python tools/critical_path.py $OUT $FL "$@" > $OUT/critical_path_$FL.txt 2>&1
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
cat $OUT/critical_path_$FL.txt
