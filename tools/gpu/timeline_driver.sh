# timeline of the passes of the driver's pool (fast flavour), one pool (--warmup 0): tag = $1
set -u
TAG=${1:-r04tl}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- python $REPO/bench.py --gpus 1 --steps 20 --warmup 0 --no-legs --no-strict --no-pcie --cpu-sample 0 > $OUT/bench.log 2>&1
cd $REPO
f=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $f --min-us 50 > $OUT/timeline.txt 2>&1
tail -30 $OUT/timeline.txt
find $OUT/kt -name "*.csv" -size +20M -delete; find $OUT -name "*.db" -delete
