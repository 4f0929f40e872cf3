"""The multi-device paths on REAL devices when a box has at least two GPUs (skipped on the one-GPU test boxes, so
that the driver's first 8-GPU run is not the first time this code meets a second device):
  * bench.py --gpus 2: self-launch through torch.distributed.run, two ranks over RCCL, the final all-gather of
    solutions and status words across two devices (bench.py asserts every rank's slice of the gathered copy);
  * pikamd_solve_batch_sharded over two device ordinals against one call on one device, bit for bit."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _n_devices():
    try:
        import torch
        return torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:
        return 0


needs_two = pytest.mark.skipif(_n_devices() < 2, reason="needs two visible GPUs")


@needs_two
@pytest.mark.parametrize("arithmetic", ["exact", "fast"])
def test_bench_two_ranks_over_rccl(arithmetic):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--arithmetic", arithmetic, "--no-legs", "--no-strict", "--no-pcie", "--cpu-sample", "0"],
                       env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert lines, r.stdout[-2000:]
    d = json.loads(lines[-1])
    assert d["n_gpus"] == 2 and d["n_ranks"] == 2 and d["steps"] == 2 and d["scaling"] == "weak"
    assert d["config"]["parallelism"] == "shard2" and d["config"]["arithmetic"] == arithmetic
    assert 0.97 < d["config"]["success_rate"] <= 1.0 and d["value"] > 0


@needs_two
def test_sharded_call_over_two_devices_equals_one_call():
    import pick_ik_amd as pk
    from pick_ik_amd import solver as S
    ch = pk.robots.panda()
    handles = [pk.Solver(ch, device=0), pk.Solver(ch, device=1)]
    try:
        B = 30001
        rng = np.random.default_rng(7)
        goal = handles[0].fk(rng.uniform(ch.qmin, ch.qmax, size=(B, 7)))
        seed = np.tile(pk.robots.PANDA_HOME, (B, 1))
        p = pk.default_params(memetic_population_size=32, memetic_max_generations=12)
        for exact in (False, True):
            for h in handles:
                h.set_option("arithmetic", "exact" if exact else "fast")
            ref = handles[0].solve_batch(p, goal, seed, rng_seed=5, problem_offset=77)
            one = handles[1].solve_batch(p, goal, seed, rng_seed=5, problem_offset=77)  # the second device alone
            got = S.solve_batch_sharded(handles, p, goal, seed, rng_seed=5, problem_offset=77)
            for a, b, c in zip(got, ref, one):
                np.testing.assert_array_equal(a, b)
                np.testing.assert_array_equal(c, b)
    finally:
        for h in handles:
            h.close()
