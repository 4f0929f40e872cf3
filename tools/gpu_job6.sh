#!/bin/bash
set -u
O=gpurun_out/j6; mkdir -p $O
timeout 600 python -m pytest tests/test_plugin_shim.py tests/test_host_cpp.py -m gpu -q --timeout 300 > $O/pytest_shim.log 2>&1; echo "rc $?" >> $O/pytest_shim.log
timeout 60 tests/native/shim_check gpu > $O/shim_check.txt 2>&1; echo "rc $?" >> $O/shim_check.txt
tools/profile_driver_cmd.sh r02c
