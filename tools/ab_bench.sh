#!/bin/bash
# Interleaved A/B of library builds on ONE GPU box (run through gpurun): single bench runs differ by
# +-3 % from run to run and from box to box, so a change under ~5 % only shows in interleaved
# repetitions on the same box.
# usage: tools/ab_bench.sh <reps> "<bench.py args>" <libA.so> <libB.so> [...]
#   e.g. gpurun -- 'tools/ab_bench.sh 5 "" _ab/base.so _ab/new.so'      (default run)
#        gpurun -- 'tools/ab_bench.sh 4 "--gpus 1 --steps 20 --warmup 5" _ab/base.so _ab/new.so'
# Build the variants with `PIK_ONLY_D=7 python pick_ik_amd/build.py --fast-only` and copy
# pick_ik_amd/libpick_ik_amd.so aside under a directory that travels with the snapshot (not gpurun_out/).
reps=$1; shift; args=$1; shift
for rep in $(seq "$reps"); do for lib in "$@"; do
  v=$(PIK_LIB=$(realpath "$lib") timeout 300 python bench.py $args --no-strict --no-pcie --cpu-sample 0 2>/dev/null |
      python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']))")
  echo "$(basename "$lib") $v"
done; done | sort | awk '{s[$1]+=$2; n[$1]++; if(!($1 in mn)||$2<mn[$1])mn[$1]=$2; if($2>mx[$1])mx[$1]=$2}
  END {for (k in s) printf "%-24s mean %.0f  min %d  max %d  n %d\n", k, s[k]/n[k], mn[k], mx[k], n[k]}'
