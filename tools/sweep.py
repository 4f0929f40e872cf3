#!/usr/bin/env python3
"""Runs bench.py over a list of argument sets and prints one compact line each (GPU box)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True,
                         text=True, env=e, timeout=600)
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    if not lines:
        print("FAILED", args, out.stderr[-500:])
        return None
    return json.loads(lines[-1])


if __name__ == "__main__":
    for spec in sys.argv[1:]:
        env = {}
        args = []
        for tok in spec.split():
            if "=" in tok and not tok.startswith("--"):
                k, v = tok.split("=", 1)
                env[k] = v
            else:
                args.append(tok)
        d = run(args + ["--cpu-sample", "0"], env)
        if d:
            r = d["roofline"]
            c = d["config"]
            print(f"{spec:70s} | {d['value']:10.0f} solves/s  {d['ms_per_step']:8.3f} ms/step  launch "
                  f"{r['avg_launch_ms']:8.3f} ms  succ {c['success_rate']:.4f}  gens {c['mean_generations']:.2f}  "
                  f"frac {(r['frac'] or 0):.4f}  enq {c.get('host_enqueue_ms_per_step', 0):.3f} ms")
