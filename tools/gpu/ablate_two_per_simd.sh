# What memetic_kernel<D,1,false,2> (two wavefronts per SIMD, 256 registers, 360 B of scratch per lane) buys and
# what its scratch costs.  (1) interleaved A/B on one box: the default run and the driver's command with the
# variant (default) and without it (option two_per_simd = 0: the one-per-SIMD build, 512 registers, no scratch,
# takes its passes); (2) PMC instruction counts per kernel of the default run: scratch / global memory
# instructions (FLAT + VMEM) beside the vector instructions.   tag = $1  -> gpurun_out/<tag>/
set -u
TAG=${1:-r04occ}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
CLEAN="--cpu-sample 0 --no-strict --no-pcie --no-legs"
one() { python bench.py "$@" $CLEAN 2>/dev/null | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'])"; }
{
echo "# bench.py --gpus 1 (512 steps, pools of 64, 4 streams): converged solves/s, ms per step"
for rep in 1 2 3; do
  echo "default        $(one --gpus 1)"
  echo "two_per_simd=0 $(PIK_OCC2=0 one --gpus 1)"
done
echo "# bench.py --gpus 1 --steps 20 --warmup 5 (the driver's command)"
for rep in 1 2 3; do
  echo "default        $(one --gpus 1 --steps 20 --warmup 5)"
  echo "two_per_simd=0 $(PIK_OCC2=0 one --gpus 1 --steps 20 --warmup 5)"
done
echo "# bench.py --config 3 --steps 3 --warmup 1"
for rep in 1 2; do
  echo "default        $(one --config 3 --steps 3 --warmup 1)"
  echo "two_per_simd=0 $(PIK_OCC2=0 one --config 3 --steps 3 --warmup 1)"
done
} | tee $OUT/ab.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
  --kernel-trace --output-format csv -d $OUT/pmc -o p -- python $REPO/bench.py --gpus 1 $CLEAN > $OUT/pmc.log 2>&1
cd $REPO
python - $OUT <<'PY' | tee $OUT/instructions.txt
import csv, glob, sys, re, collections
out = sys.argv[1]
fs = glob.glob(f"{out}/pmc/**/*counter_collection.csv", recursive=True)
rows = csv.DictReader(open(fs[0]))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    k = r["Kernel_Name"]
    m = re.search(r"(pik\w*)::memetic_kernel<([^>]*)>", k)
    if not m: continue
    agg[m.group(1) + "::memetic_kernel<" + m.group(2).replace(" ", "") + ">"][r["Counter_Name"]] += float(r["Counter_Value"])
print("# default run, PMC sums per kernel variant (wave-level instruction counts)")
print(f"{'kernel':52s} {'waves':>9s} {'VALU':>11s} {'SALU':>11s} {'LDS':>10s} {'FLAT':>10s} {'VMEM_RD':>10s} {'VMEM_WR':>10s}  (FLAT+VMEM)/VALU")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_INSTS_VALU"]):
    mem = v["SQ_INSTS_FLAT"] + v["SQ_INSTS_VMEM_RD"] + v["SQ_INSTS_VMEM_WR"]
    print(f"{k:52s} {v['SQ_WAVES']:9.4g} {v['SQ_INSTS_VALU']:11.4g} {v['SQ_INSTS_SALU']:11.4g} {v['SQ_INSTS_LDS']:10.4g} {v['SQ_INSTS_FLAT']:10.4g} "
          f"{v['SQ_INSTS_VMEM_RD']:10.4g} {v['SQ_INSTS_VMEM_WR']:10.4g}  {mem / max(v['SQ_INSTS_VALU'], 1):.5f}")
PY
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +10M -delete
