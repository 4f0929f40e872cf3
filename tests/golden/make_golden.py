"""Generates tests/golden/golden_v4.npz from the CPU oracle (which is itself pinned against the
reference's known-answer tests in tests/test_oracle_golden.py).

The reference cannot be compiled or imported in this environment (needs ROS 2 / MoveIt / Eigen /
rsl), so these vectors are outputs of the restatement, not of pick_ik itself.
Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from pick_ik_amd import robots  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_v4.npz")

CONFIGS = {
    # name: (robot, seed pose, params)   -- scaled-down versions of BASELINE.json configs 2..4
    "panda_p16": ("panda", robots.PANDA_HOME, dict()),
    "panda_p128": ("panda", robots.PANDA_HOME, dict(memetic_population_size=128)),
    "ur5_p256_goals": ("ur5", robots.UR5_HOME,
                       dict(memetic_population_size=256, center_joints_weight=0.01,
                            minimal_displacement_weight=0.001, cost_threshold=0.01)),
    "panda_approx": ("panda", robots.PANDA_HOME,
                     dict(memetic_population_size=128, return_approximate_solution=1)),
}


def targets(o, chain, rng, n, unreachable=False):
    q = rng.uniform(chain.qmin, chain.qmax, size=(n, chain.dof))
    g = o.fk(q)
    if unreachable:
        d = g[:, :3] / np.linalg.norm(g[:, :3], axis=1, keepdims=True)
        g[:, :3] = d * rng.uniform(1.0, 1.5, size=(n, 1))
    return q, g


def main():
    out = {}
    rng = np.random.default_rng(20260928)
    for name in ("panda", "ur5", "rr"):
        ch = robots.by_name(name)
        o = O.Oracle(ch)
        q = rng.uniform(ch.qmin, ch.qmax, size=(64, ch.dof))
        out[f"fk_{name}_q"] = q
        out[f"fk_{name}_pose"] = o.fk(q)
        # cost / solution test of random candidates against random goals, with all goals enabled
        _, goal = targets(o, ch, rng, 64)
        seed = rng.uniform(ch.qmin, ch.qmax, size=(64, ch.dof))
        p = O.default_params(center_joints_weight=0.3, avoid_joint_limits_weight=0.2,
                             minimal_displacement_weight=0.1)
        cost = np.array([o.cost(p, goal[i], seed[i], q[i])[0][0] for i in range(64)])
        out[f"cost_{name}_goal"] = goal
        out[f"cost_{name}_seed"] = seed
        out[f"cost_{name}_cost"] = cost
        # one step() from those candidates
        p0 = O.default_params()
        c0 = np.array([o.cost(p0, goal[i], seed[i], q[i])[0][0] for i in range(64)])
        local, best, lc, bc, grad, imp = o.gd_step(p0, goal, seed, q, q, c0, c0)
        out[f"step_{name}_c0"] = c0
        out[f"step_{name}_local"] = local
        out[f"step_{name}_lc"] = lc
        out[f"step_{name}_grad"] = grad
        # ik_gradient from a nearby seed
        near = np.clip(q + rng.normal(0, 0.1, size=q.shape), ch.qmin, ch.qmax)
        pg = O.default_params(mode=1)
        sol, st, c, stats = o.solve_batch(pg, o.fk(q), near)
        out[f"gd_{name}_seed"] = near
        out[f"gd_{name}_sol"] = sol
        out[f"gd_{name}_status"] = st
        out[f"gd_{name}_iters"] = stats["generations"]
    for cname, (robot, home, kw) in CONFIGS.items():
        ch = robots.by_name(robot)
        o = O.Oracle(ch)
        n = 512  # enough problems for statistical gates (paired verdict test, quantiles) to mean something
        _, goal = targets(o, ch, rng, n, unreachable=(cname == "panda_approx"))
        seed = np.tile(home, (n, 1))
        sol, st, c, stats = o.solve_batch(O.default_params(**kw), goal, seed, rng_seed=0xC0FFEE,
                                          num_threads=O.max_threads())
        out[f"mem_{cname}_goal"] = goal
        out[f"mem_{cname}_sol"] = sol
        out[f"mem_{cname}_status"] = st
        out[f"mem_{cname}_cost"] = c
        out[f"mem_{cname}_gens"] = stats["generations"]
        out[f"mem_{cname}_evals"] = stats["cost_evals"]
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
