#!/bin/bash
set -u
O=gpurun_out/j4; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --timeout 150 -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
for cfg in "20 1" "10 2" "5 4" "4 4" "1 16"; do set -- $cfg; timeout 120 python bench.py --gpus 1 --steps 20 --warmup 5 --pool $1 --streams $2 --cpu-sample 0 --no-strict --no-pcie > $O/drv_pool$1_s$2.json 2>&1; done
timeout 120 python bench.py --steps 16 --warmup 2 --pool 1 --streams 1 --cpu-sample 0 --no-strict --no-pcie > $O/serial_s1.json 2>&1
for cfg in "16 2" "16 4" "32 2" "32 4" "8 8" "64 2"; do set -- $cfg; timeout 200 python bench.py --pool $1 --streams $2 --cpu-sample 0 --no-strict --no-pcie > $O/def_pool$1_s$2.json 2>&1; done
for marks in "2,4,8,16,32,64" "2,4,8,12,16,20,24,32,40,48,64,80" "1,2,3,4,6,8,12,16,24,32,48,64,80"; do PIK_PASSES=$marks timeout 120 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --no-strict --no-pcie > "$O/drv_marks_${marks//,/_}.json" 2>&1; done
