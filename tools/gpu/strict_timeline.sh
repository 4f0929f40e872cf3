# timeline of the exact kernels' passes on the driver's command (both exact builds); tag = $1
set -u
TAG=${1:-r04x}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- python $REPO/bench.py --gpus 1 --steps 20 --warmup 5 --no-legs --no-pcie --cpu-sample 0 > $OUT/bench.log 2>&1
cd $REPO
f=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
for ns in pik_exact pik_strict; do
  head -1 $f > $OUT/${ns}_trace.csv; grep $ns $f >> $OUT/${ns}_trace.csv
  python tools/timeline.py $OUT/${ns}_trace.csv --min-us 50 > $OUT/${ns}_timeline.txt 2>&1
  echo "== $ns"; tail -19 $OUT/${ns}_timeline.txt
done
find $OUT/kt -name "*.csv" -size +20M -delete; find $OUT -name "*.db" -delete
grep "^{" $OUT/bench.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); pe=d['parity_exact']; print('value',d['value']); print('exact', pe['value'], pe['identical_to_oracle_on_sample'], pe.get('sustained',{}).get('value')); print('plain', pe['plain_ieee']['value'], pe['plain_ieee']['identical_to_oracle_on_sample'], pe['plain_ieee'].get('sustained',{}).get('value'))"
