"""bench.py's distributed path on the GPU box: one rank over RCCL (`nccl` backend, PIK_BENCH_FORCE_DIST=1) --
process group, warm-up gather, timed region with the final all-gather of solutions and status words, the
check that the gathered copy holds this rank's results at this rank's position (bench.py asserts it), the JSON
line.  The driver's first `--gpus 8` run must not be the first time this code meets RCCL."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(*args):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(PIK_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", *args, "--no-legs", "--no-strict", "--no-pcie",
                        "--cpu-sample", "0"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert lines, r.stdout[-2000:]
    return json.loads(lines[-1])


def test_config2_one_rank_over_rccl():
    d = _run("--steps", "2", "--warmup", "1")
    assert d["n_gpus"] == 1 and d["n_ranks"] == 1 and d["steps"] == 2
    assert d["scaling"] == "weak" and d["value"] > 0
    assert 0.97 < d["config"]["success_rate"] <= 1.0
    assert d["config"]["parallelism"] == "shard1"


def test_config5_one_rank_over_rccl():
    d = _run("--config", "5", "--batch", "8192", "--steps", "2", "--warmup", "1")
    assert d["n_gpus"] == 1 and d["n_ranks"] == 1 and d["scaling"] == "strong"
    assert d["config"]["batch_per_gpu"] == 8192 and d["value"] > 0
    assert 0.97 < d["config"]["success_rate"] <= 1.0
