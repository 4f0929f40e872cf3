# all benchmarked shapes under rocprofv3 (tools/profile_driver_cmd.sh each); round tag = $1
set -u
R=${1:-r04}
bash tools/profile_driver_cmd.sh ${R}_driver --gpus 1 --steps 20 --warmup 5
bash tools/profile_driver_cmd.sh ${R}_default --gpus 1
bash tools/profile_driver_cmd.sh ${R}_single --gpus 1 --steps 20 --warmup 5 --pool 1 --streams 1
bash tools/profile_driver_cmd.sh ${R}_config3 --gpus 1 --config 3 --steps 3 --warmup 1
bash tools/profile_driver_cmd.sh ${R}_config4 --gpus 1 --config 4 --steps 3 --warmup 1
bash tools/profile_driver_cmd.sh ${R}_config5 --gpus 1 --config 5 --steps 2 --warmup 1
for s in driver default single config3 config4 config5; do echo "== $s"; tail -5 gpurun_out/prof_${R}_$s/summary.txt; done
