#!/bin/bash
# round-2 GPU job 1: baseline measurements before the kernel work
set -u
mkdir -p gpurun_out/j1
O=gpurun_out/j1
(cd /tmp && rocprofv3 -L 2>/dev/null | grep -i -E "F64|THREAD_CYCLES|SQ_INSTS_VALU" | head -40) > $O/counters.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q -s > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
PIK_LIB=$PWD/pick_ik_amd/libpick_ik_amd_nogen.so python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 > $O/bench_driver_nogen.json 2>&1
python bench.py --cpu-sample 0 > $O/bench_default.json 2>&1
PIK_LIB=$PWD/pick_ik_amd/libpick_ik_amd_nogen.so python bench.py --cpu-sample 0 > $O/bench_default_nogen.json 2>&1
# one mega-batch = the 20 driver steps as one launch chain; schedules for the tail
for sched in "" "0:1,32:4" "0:1,16:4" "0:1,8:4"; do
  PIK_LPE_SCHED="$sched" python bench.py --batch 81920 --steps 3 --warmup 1 --streams 1 --cpu-sample 0 > "$O/mega_s1_sched_${sched//[:,]/_}.json" 2>&1
done
python bench.py --batch 20480 --steps 8 --warmup 4 --streams 4 --cpu-sample 0 > $O/chunk4.json 2>&1
python bench.py --steps 16 --warmup 2 --streams 1 --cpu-sample 0 > $O/serial_s1.json 2>&1
tools/profile_driver_cmd.sh r02a
