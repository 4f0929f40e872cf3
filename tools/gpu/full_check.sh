# whole GPU suite, then the strict timeline (tag = $1)
set -u
TAG=${1:-r04x}
mkdir -p gpurun_out/$TAG
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/$TAG/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/$TAG/pytest.log
sed -i "s#gpurun_out/r04[a-z]*#gpurun_out/$TAG#" tools/gpu/strict_timeline.sh
bash tools/gpu/strict_timeline.sh | tail -32
grep -o '"parity_exact": {"value": [0-9.]*' gpurun_out/$TAG/bench.log
