#!/bin/bash
set -u
O=gpurun_out/j7; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -s --timeout 200 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
PIK_FUZZ_CASES=120 timeout 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -q --timeout 500 > $O/fuzz120.log 2>&1; echo "rc $?" >> $O/fuzz120.log
