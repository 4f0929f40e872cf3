# What `arithmetic = exact` costs on robots that are NOT of the Panda / UR5 kind (VERDICT r05 "missing 4"): a chain of
# class 2 with nine variables, a tree with two tip frames, a chain on a floating base -- each through bench.py's own
# timed region (4096 targets per step, population 128) under rocprofv3 (kernel trace, then PMC passes), then
# tools/price_robots.py turns counters + bench line into gpurun_out/<tag>_price_<robot>.json.   tag = $1, robots = $2...
set -u
TAG=${1:-r06}; shift || true
ROBOTS=${@:-panda_on_torso torso_dual_arm floating_panda}
for r in $ROBOTS; do
  for fl in exact fast; do
    sfx=""; [ "$fl" = exact ] && sfx="_exact"
    PIK_PROFILE_SKIP_FULL=1 timeout 900 bash tools/profile_driver_cmd.sh ${TAG}_price_${r}$sfx --gpus 1 --steps 8 --warmup 2 --robot $r --arithmetic $fl
    python tools/price_robots.py gpurun_out/prof_${TAG}_price_${r}$sfx > gpurun_out/${TAG}_price_${r}$sfx.json
    cat gpurun_out/${TAG}_price_${r}$sfx.json
  done
done
