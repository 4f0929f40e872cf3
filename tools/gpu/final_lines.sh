# The bench lines of the final library, with the committed roofline inputs (run AFTER tools/make_roofline_inputs.py):
# the driver's command, the default run, one batch at a time, configs 3 / 4 / 5.   tag = $1 -> gpurun_out/<tag>/
set -u
TAG=${1:-r06final}
OUT=gpurun_out/$TAG; mkdir -p $OUT
line() { local name=$1; shift; python bench.py "$@" 2> $OUT/$name.err | grep "^{" | tail -1 > $OUT/$name.json; python - $OUT/$name.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]; f = d.get("fast", {})
print(sys.argv[1], d["config"]["arithmetic"], round(d["value"]), round(d["ms_per_step"], 4), "frac", r["frac"] and round(r["frac"], 4),
      "stale", r.get("inputs_stale"), "identical", d.get("parity", {}).get("identical_to_oracle_on_sample"),
      "| fast", round(f.get("value", 0)), "frac", (f.get("roofline") or {}).get("frac"))
PY
}
line bench_driver --gpus 1 --steps 20 --warmup 5
line bench_default --gpus 1
line bench_one_batch_at_a_time --gpus 1 --steps 20 --warmup 5 --pool 1 --streams 1 --no-legs
line bench_config3 --gpus 1 --config 3 --steps 3 --warmup 1
line bench_config4 --gpus 1 --config 4 --steps 3 --warmup 1
line bench_config5 --gpus 1 --config 5 --steps 2 --warmup 1
