#!/bin/bash
set -u
O=gpurun_out/j5; mkdir -p $O
PIK_FUZZ_CASES=300 PIK_FUZZ_TREES=100 timeout 1200 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_multi_tip.py -m gpu -q --timeout 900 > $O/soak.log 2>&1; echo "rc $?" >> $O/soak.log
tools/profile_driver_cmd.sh r02b
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python bench.py --pool 64 --streams 2 --cpu-sample 0 > $O/bench_pool64_s2.json 2>&1
timeout 300 python bench.py --config 5 --steps 2 --warmup 1 --cpu-sample 0 --no-strict --no-pcie > $O/bench_config5.json 2>&1
PIK_BENCH_FORCE_DIST=1 timeout 300 python bench.py --config 5 --steps 2 --warmup 1 --cpu-sample 0 --no-strict --no-pcie > $O/bench_config5_dist1.json 2>&1
PIK_BENCH_FORCE_DIST=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --no-strict --no-pcie > $O/bench_driver_dist1.json 2>&1
