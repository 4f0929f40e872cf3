#!/bin/bash
set -u
O=gpurun_out/j3; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_multibatch.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q --timeout 120 -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
for lpe in 1 4 8 16; do PIK_LPE=$lpe timeout 120 python bench.py --steps 16 --warmup 2 --streams 1 --cpu-sample 0 > $O/serial_s1_lpe$lpe.json 2>&1; done
timeout 120 python bench.py --steps 16 --warmup 2 --streams 1 --cpu-sample 0 > $O/serial_s1.json 2>&1
for sched in "0:1,8:16" "0:1,16:16" "0:1,4:4,16:16" "0:4,8:16" "0:4,4:16" "0:4,2:8,8:16"; do
  PIK_LPE_SCHED="$sched" timeout 120 python bench.py --steps 16 --warmup 2 --streams 1 --cpu-sample 0 > "$O/serial_s1_sched_${sched//[:,]/_}.json" 2>&1
done
