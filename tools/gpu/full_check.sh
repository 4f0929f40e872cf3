# whole GPU suite, then the exact kernels' timeline (tag = $1)
set -u
TAG=${1:-r04x}
mkdir -p gpurun_out/$TAG
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/$TAG/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/$TAG/pytest.log
bash tools/gpu/strict_timeline.sh $TAG
