#!/bin/bash
# Where a lone wavefront's cycles go beyond its instruction issue: the long runners of config 2 at a fixed number of
# lanes per elite (tools/gpu/long_runners.py) under two PMC passes -- instruction cache and fetch, waits and the
# memory-instruction levels.  usage: tools/gpu/stall_counters.sh <tag> <fast|exact> <lanes>...  -> gpurun_out/<tag>/
set -u
TAG=$1; FL=$2; shift 2
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  D=$OUT/${FL}_lpe$L; mkdir -p $D
  rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $D/a -o p -- python $REPO/tools/gpu/long_runners.py $FL $L > $D/a.log 2>&1
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $D/b -o p -- python $REPO/tools/gpu/long_runners.py $FL $L > $D/b.log 2>&1
  python - $D $FL $L <<'PY'
import csv, glob, sys, collections
d, fl, L = sys.argv[1:4]
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(d + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "memetic_kernel" not in k: continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
print(f"== {fl} flavour, {L} lanes per elite (sums over the memetic kernels' dispatches)")
for k in sorted(tot): print(f"   {k:32s} {tot[k]:16.0f}   ({n[k]} rows)")
wc = tot.get("SQ_WAVE_CYCLES", 0)
if wc:
    for k in ("SQ_IFETCH_LEVEL", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_INST_LEVEL_SMEM", "SQ_INST_LEVEL_VMEM", "SQ_INST_LEVEL_LDS"):
        if k in tot: print(f"   {k} / SQ_WAVE_CYCLES = {tot[k] / wc * (2 if k in tot and n[k] != n['SQ_WAVE_CYCLES'] else 1):.3f}")
    if "SQC_ICACHE_REQ" in tot: print(f"   icache hit rate {tot['SQC_ICACHE_HITS'] / max(tot['SQC_ICACHE_REQ'], 1):.4f}, misses per branch {tot['SQC_ICACHE_MISSES'] / max(tot['SQ_INSTS_BRANCH'], 1):.4f}")
PY
done 2>&1 | tee $OUT/stall_counters_$FL.txt
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
