# Fuzz soak of the round-6 forms on generated cases OTHER than the suite's: the class fuzz at lengths 1..10 (80 cases per
# exact build and shift), the floating-joint fuzz, the random trees of several tips, the mimic fuzz.  -> gpurun_out/<tag>/soak.txt
set -u
TAG=${1:-r06soak}; shift || true
SHIFTS=${@:-810000 820000 830000 840000}
OUT=gpurun_out/$TAG; mkdir -p $OUT
{
echo "# tools/gpu/soak_r06.sh: tests/test_gpu_fuzz.py (class cases 80), test_gpu_floating.py, test_gpu_multi_tip.py (random trees 32), test_gpu_mimic.py on other generated cases (PIK_FUZZ_SEED_SHIFT), final library of round 6, tolerance 0 against the oracle"
for s in $SHIFTS; do
  echo "== PIK_FUZZ_SEED_SHIFT=$s"
  PIK_FUZZ_SEED_SHIFT=$s PIK_FUZZ_CLASS_CASES=80 PIK_FUZZ_TREES=32 timeout 1500 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_floating.py tests/test_gpu_multi_tip.py tests/test_gpu_mimic.py -q -m gpu 2>&1 | tail -2
done
} > $OUT/soak.txt 2>&1
cat $OUT/soak.txt
