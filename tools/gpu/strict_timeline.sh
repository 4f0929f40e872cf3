set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r04a; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- python $REPO/bench.py --gpus 1 --steps 20 --warmup 5 --no-legs --no-pcie --cpu-sample 0 > $OUT/bench.log 2>&1
cd $REPO
f=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
head -1 $f > $OUT/strict_trace.csv; grep pik_strict $f >> $OUT/strict_trace.csv
python tools/timeline.py $OUT/strict_trace.csv --min-us 50 > $OUT/strict_timeline.txt 2>&1
find $OUT/kt -name "*.csv" -size +20M -delete; find $OUT -name "*.db" -delete
tail -3 $OUT/bench.log | cut -c1-600
tail -40 $OUT/strict_timeline.txt
