#!/bin/bash
set -u
O=gpurun_out/j2; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 > $O/bench_driver.json 2> $O/bench_driver.err
python bench.py --cpu-sample 0 > $O/bench_default.json 2>&1
python bench.py --steps 16 --warmup 2 --streams 1 --cpu-sample 0 > $O/serial_s1.json 2>&1
for lpe in 4 8 16; do PIK_LPE=$lpe python bench.py --steps 16 --warmup 2 --streams 1 --cpu-sample 0 > $O/serial_s1_lpe$lpe.json 2>&1; done
