# strict (exact) flavour: parity tests against the oracle, then the timeline of the driver's command
set -u
mkdir -p gpurun_out/r04b
timeout 1500 python -m pytest tests/test_gpu_strict_parity.py tests/test_gpu_fuzz.py tests/test_gpu_floating.py tests/test_gpu_multi_tip.py -x -q -m gpu > gpurun_out/r04b/pytest.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/r04b/pytest.log
sed -i 's#gpurun_out/r04a#gpurun_out/r04b#' tools/gpu/strict_timeline.sh
bash tools/gpu/strict_timeline.sh
