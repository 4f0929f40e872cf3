"""bench.py owns its launch: `python bench.py --gpus N` without a launcher starts N ranks itself
(torch.distributed.run on 127.0.0.1), refuses when fewer than N GPUs are visible, and refuses a
launcher whose WORLD_SIZE disagrees with --gpus.  On CPU the launch path is exercised with
--launch-check (gloo): ranks started, process group formed, one all-reduce, world size reported."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "PIK_BENCH_SELF_LAUNCHED")}
    env.update(kw)
    return env


def _last_json(text):
    lines = [ln for ln in text.strip().splitlines() if ln.startswith("{")]
    assert lines, text
    return json.loads(lines[-1])


def test_gpus_2_starts_two_ranks_by_itself():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check"], env=_env(), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert d == {"launch_check": True, "n_gpus": 2, "ranks": 2, "all_reduce_of_ones": 2, "backend": "gloo",
                 "self_launched": True}


def test_external_launcher_still_works():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2",
                        "--launch-check"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["ranks"] == 2 and d["self_launched"] is False


def test_refuses_more_ranks_than_gpus():
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 64:
        return
    r = subprocess.run([sys.executable, BENCH, "--gpus", "64", "--steps", "1", "--warmup", "0"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "refusing" in (r.stderr + r.stdout)


def test_refuses_a_launcher_with_a_different_rank_count():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check"], env=_env(WORLD_SIZE="1", RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "must agree" in (r.stderr + r.stdout)
