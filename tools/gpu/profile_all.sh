# all benchmarked shapes under rocprofv3 (tools/profile_driver_cmd.sh each); round tag = $1, flavours = $2 (default both)
# -> gpurun_out/prof_<tag>_<shape>[_exact]/ ; then: python tools/make_roofline_inputs.py <tag> <shape>[_exact]=<tag>_<shape>[_exact] ...
set -u
R=${1:-r06}
FL=${2:-"exact fast"}
for fl in $FL; do
  sfx=""; [ "$fl" = exact ] && sfx="_exact"
  A="--arithmetic $fl"
  bash tools/profile_driver_cmd.sh ${R}_driver_cmd$sfx --gpus 1 --steps 20 --warmup 5 $A
  bash tools/profile_driver_cmd.sh ${R}_default_run$sfx --gpus 1 $A
  bash tools/profile_driver_cmd.sh ${R}_single_batch$sfx --gpus 1 --steps 20 --warmup 5 --pool 1 --streams 1 $A
  bash tools/profile_driver_cmd.sh ${R}_config3$sfx --gpus 1 --config 3 --steps 3 --warmup 1 $A
  bash tools/profile_driver_cmd.sh ${R}_config4$sfx --gpus 1 --config 4 --steps 3 --warmup 1 $A
  bash tools/profile_driver_cmd.sh ${R}_config5$sfx --gpus 1 --config 5 --steps 2 --warmup 1 $A
  for s in driver_cmd default_run single_batch config3 config4 config5; do echo "== $s$sfx"; tail -5 gpurun_out/prof_${R}_$s$sfx/summary.txt; done
done
