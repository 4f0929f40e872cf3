# PMC counters of the exact kernels on the driver's pool, per kernel variant; tag = $1, flavour = $2
set -u
TAG=${1:-r04x}; FL=${2:-exact}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $REPO/tools/gpu/exact_only.py $FL > $OUT/plain_run.log 2>&1; cat $OUT/plain_run.log
run() { local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o p -- python $REPO/tools/gpu/exact_only.py $FL > $OUT/${name}.log 2>&1; }
run a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_FLAT SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run b SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC
run c SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_BRANCH SQ_INST_CYCLES_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
cd $REPO
python - <<'PY' $OUT
import csv, glob, sys, re, collections
out = sys.argv[1]
for name in "abc":
    fs = glob.glob(f"{out}/{name}/**/*counter_collection.csv", recursive=True)
    if not fs: print(name, "no csv"); continue
    rows = list(csv.DictReader(open(fs[0])))
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in rows:
        k = r["Kernel_Name"]
        if "memetic_kernel" not in k: continue
        m = re.search(r"(pik\w*)::memetic_kernel<([^>]*)>", k); key = m.group(1) + "<" + m.group(2) + ">"
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in agg.items():
        print(name, k, {a: f"{b:.4g}" for a, b in v.items()})
PY
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +10M -delete
