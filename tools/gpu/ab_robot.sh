# interleaved A/B of library builds on one robot through bench.py's timed region: ab_robot.sh <robot> <reps> lib...
# (per library and repetition: solves/s, ms per step, success rate, identical to the oracle on the sample)
robot=$1; reps=$2; shift 2
for rep in $(seq "$reps"); do for lib in "$@"; do
  echo "$(basename $lib) $(PIK_LIB=$(realpath $lib) python bench.py --robot $robot --arithmetic exact --no-legs --no-pcie --no-strict --cpu-sample 0 --steps 8 --warmup 2 2>/dev/null | grep '^{' | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"]), round(d["ms_per_step"],2), d["config"].get("success_rate"), (d.get("parity") or {}).get("identical_to_oracle_on_sample"))')"
done; done | sort
