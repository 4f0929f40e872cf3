"""Shared helpers for the parity tests (data generation and the named configurations)."""
import os

import numpy as np
import pytest

from pick_ik_amd import robots

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_v4.npz")

# scaled-down BASELINE.json configs (same as tests/golden/make_golden.py)
CONFIGS = {
    "panda_p16": ("panda", robots.PANDA_HOME, dict()),
    "panda_p128": ("panda", robots.PANDA_HOME, dict(memetic_population_size=128)),
    "ur5_p256_goals": ("ur5", robots.UR5_HOME,
                       dict(memetic_population_size=256, center_joints_weight=0.01,
                            minimal_displacement_weight=0.001, cost_threshold=0.01)),
    "panda_approx": ("panda", robots.PANDA_HOME,
                     dict(memetic_population_size=128, return_approximate_solution=1)),
}
# ... + BASELINE configs[4]'s population 512, for the tolerance-0 comparisons with the oracle (the golden file holds
# vectors for CONFIGS only)
EXACT_CONFIGS = dict(CONFIGS, panda_p512=("panda", robots.PANDA_HOME, dict(memetic_population_size=512)))


#: the two arithmetic flavours of the product library, as pk.Solver(..., exact=<value>): None = the library's default
#: (arithmetic = exact), False = the opt-in fast flavour
ARITHMETIC = [pytest.param(None, id="default_exact"), pytest.param(False, id="fast")]


def golden():
    return np.load(GOLDEN)


def paired_verdict_gate(ok_a, ok_b, what=""):
    """The same problems solved by two implementations of a chaotic search: verdicts flip both ways.
    Under "equally likely to succeed" the discordant pairs split like fair coin tosses (McNemar), so
    the gate is |n(a only) - n(b only)| <= 3 sqrt(n discordant) + 1 -- three sigma, however many
    problems and however many flips; returns (only a, only b)."""
    ok_a, ok_b = np.asarray(ok_a, bool), np.asarray(ok_b, bool)
    a_only, b_only = int((ok_a & ~ok_b).sum()), int((~ok_a & ok_b).sum())
    assert abs(a_only - b_only) <= 3.0 * np.sqrt(a_only + b_only) + 1.0, (what, a_only, b_only, len(ok_a))
    return a_only, b_only


def random_targets(fk, chain, rng, n, unreachable=False):
    """q* ~ U(limits), target = FK(q*); unreachable: position pushed out to radius U(1.0, 1.5) m
    (SURVEY.md section 8(d) config 4)."""
    q = rng.uniform(chain.qmin, chain.qmax, size=(n, chain.dof))
    g = fk(q)
    if unreachable:
        d = g[:, :3] / np.linalg.norm(g[:, :3], axis=1, keepdims=True)
        g[:, :3] = d * rng.uniform(1.0, 1.5, size=(n, 1))
    return q, g


def quat_angle(qa, qb):
    """angular distance between unit quaternions (w x y z) [n][4]"""
    w = np.abs((qa * qb).sum(axis=1))
    d = qa[:, :1] * qb - qb[:, :1] * qa  # not used for the angle; keep simple & robust:
    del d
    # |vec| via the quaternion product qa * conj(qb)
    aw, ax, ay, az = qa.T
    bw, bx, by, bz = qb[:, 0], -qb[:, 1], -qb[:, 2], -qb[:, 3]
    vx = aw * bx + ax * bw + ay * bz - az * by
    vy = aw * by + ay * bw + az * bx - ax * bz
    vz = aw * bz + az * bw + ax * by - ay * bx
    return 2.0 * np.arctan2(np.sqrt(vx * vx + vy * vy + vz * vz), w)
