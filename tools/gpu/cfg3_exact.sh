# config 3 (UR5, population 256, 65 536 targets, joint goals) through the exact kernels: cfg3_exact.sh <reps> lib... (env passes through)
reps=$1; shift
for rep in $(seq "$reps"); do for lib in "$@"; do
  echo "$(basename $lib) $(PIK_LIB=$(realpath $lib) python bench.py --config 3 --arithmetic exact --no-legs --no-pcie --no-strict --cpu-sample 0 --steps 2 --warmup 1 2>/dev/null | grep '^{' | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"]), round(d["ms_per_step"],2), d["config"].get("success_rate"))')"
done; done | sort
