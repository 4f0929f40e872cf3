"""A host cost function that takes part in the search.

kinematics::KinematicsBase::IKCostFn is one more Goal of weight 1 per tip pose inside cost_fn and under
cost_threshold^2 in solution_fn (src/pick_ik_plugin.cpp:130-135, src/goal.cpp:146-161, 175-182, 188-203).
pikamd_solve_batch_host runs such queries on the host with the exact kernels' arithmetic; the oracle gained the same
goal as a C function pointer.  Checked here, at tolerance ZERO:
  * host solver == oracle, same callback, both exact builds (math modes "fma" / "portable"), memetic / species /
    local mode / joint goals / two tips;
  * host solver with a callback that returns 0 == the exact KERNELS on the GPU (the same algorithm, the same
    arithmetic, two executions);
  * the callback steers: a preference for one joint is followed where the plain query ignores it, and a cost that
    only a guided search gets under the threshold is solved (where ranking finished candidates cannot)."""
import numpy as np
import pytest

import pick_ik_amd as pk
from pick_ik_amd import robots

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def O(oracle_mod):
    import __graft_entry__ as g
    g.build()
    return oracle_mod


def eq(a, b, what=""):
    np.testing.assert_array_equal(a, b, err_msg=what)


def prefer(j, target, w):
    return lambda q, pose: w * (q[j] - target) ** 2


CASES = [
    ("panda", dict(memetic_population_size=24, memetic_max_generations=12, cost_threshold=0.05), prefer(2, 0.3, 0.2)),
    ("panda", dict(memetic_population_size=16, memetic_max_generations=8, memetic_num_threads=2, cost_threshold=0.05,
                   memetic_stop_on_first_solution=0), prefer(4, -0.4, 0.1)),
    ("panda", dict(mode=1, gd_max_iters=40, cost_threshold=0.05), prefer(0, 0.1, 0.3)),
    ("ur5", dict(memetic_population_size=20, memetic_max_generations=10, center_joints_weight=0.05,
                 minimal_displacement_weight=0.01, cost_threshold=0.2, return_approximate_solution=1), prefer(1, -1.0, 0.05)),
    ("torso_dual_arm", dict(memetic_population_size=16, memetic_max_generations=6, cost_threshold=0.1),
     lambda q, pose: 0.02 * (q[0] - 0.2) ** 2 * (1 + pose)),
]


@pytest.mark.parametrize("strict", [False, True], ids=["exact_fma", "plain_ieee"])
@pytest.mark.parametrize("name,kw,fn", CASES)
def test_host_solver_equals_oracle_with_the_same_cost_function(O, name, kw, fn, strict):
    ch = robots.by_name(name)
    rng = np.random.default_rng(7)
    n = 10
    q = rng.uniform(ch.qmin, ch.qmax, size=(n, ch.dof))
    seed = np.clip(q + rng.normal(0, 0.3, size=q.shape), ch.qmin, ch.qmax)
    guess = np.clip(seed + rng.normal(0, 0.05, size=q.shape), ch.qmin, ch.qmax)
    o = O.Oracle(ch)
    s = pk.Solver(ch, device=0, strict=strict)
    try:
        with O.math_mode("portable" if strict else "fma"):
            goal = o.fk(q)
            a = s.solve_batch_host(pk.default_params(**kw), goal, seed, fn, rng_seed=11, problem_offset=40,
                                   initial_guess=guess)
            b = o.solve_batch(O.default_params(**kw), goal, seed, rng_seed=11, problem_offset=40, initial_guess=guess,
                              cost_fn=fn)
        for x, y, w in zip(a, b, ("solution", "status", "cost", "stats")):
            eq(x, y, f"{name} {kw} strict={strict}: {w}")
    finally:
        s.close()


@pytest.mark.parametrize("name,kw", [(c[0], c[1]) for c in CASES])
def test_host_solver_with_a_zero_cost_equals_the_exact_kernels(O, name, kw):
    ch = robots.by_name(name)
    rng = np.random.default_rng(8)
    n = 12
    q = rng.uniform(ch.qmin, ch.qmax, size=(n, ch.dof))
    seed = np.clip(q + rng.normal(0, 0.3, size=q.shape), ch.qmin, ch.qmax)
    for how in (dict(exact=True), dict(strict=True)):
        s = pk.Solver(ch, device=0, **how)
        try:
            goal = s.fk(q)
            gpu = s.solve_batch(pk.default_params(**kw), goal, seed, rng_seed=3, problem_offset=9)
            host = s.solve_batch_host(pk.default_params(**kw), goal, seed, lambda q_, pose: 0.0, rng_seed=3, problem_offset=9)
            for x, y, w in zip(gpu, host, ("solution", "status", "cost", "stats")):
                eq(x, y, f"{name} {kw} {how}: {w}")
        finally:
            s.close()


def _mimic_panda():
    from tests.test_mimic_cpu import with_mimic
    ch, _ = with_mimic(np.random.default_rng(0), robots.panda(), 3, 1, -0.6, 0.2)
    return ch


@pytest.mark.parametrize("make,kw", [
    (_mimic_panda, dict(memetic_population_size=20, memetic_max_generations=8, cost_threshold=0.05)),
    (robots.floating_panda, dict(memetic_population_size=16, memetic_max_generations=5, cost_threshold=0.05,
                                 return_approximate_solution=1)),
    (_mimic_panda, dict(mode=1, gd_max_iters=30, cost_threshold=0.05)),
], ids=["mimic-memetic", "floating-memetic", "mimic-local"])
def test_host_solver_on_chains_of_the_literal_kernels(O, make, kw):
    """a joint that follows a variable / a floating joint: the host solver (with a callback) against the oracle, and
    (with a zero callback) against the literal kernels that serve such chains -- all three the same bits"""
    ch = make()
    rng = np.random.default_rng(12)
    n = 6
    lo = np.where(ch.bounded == 1, ch.qmin, -1.0)
    hi = np.where(ch.bounded == 1, ch.qmax, 1.0)
    q = rng.uniform(lo, hi, size=(n, ch.dof))
    seed = np.clip(q + rng.normal(0, 0.2, size=q.shape), lo, hi)
    fn = prefer(1, 0.2, 0.1)
    o = O.Oracle(ch)
    s = pk.Solver(ch, device=0)
    try:
        with O.math_mode("fma"):
            goal = o.fk(q)
            eq(s.fk(q), goal, "fk")
            a = s.solve_batch_host(pk.default_params(**kw), goal, seed, fn, rng_seed=5, problem_offset=3)
            b = o.solve_batch(O.default_params(**kw), goal, seed, rng_seed=5, problem_offset=3, cost_fn=fn)
        for x, y, w in zip(a, b, ("solution", "status", "cost", "stats")):
            eq(x, y, f"{ch.name} {kw} host vs oracle: {w}")
        gpu = s.solve_batch(pk.default_params(**kw), goal, seed, rng_seed=5, problem_offset=3)
        host = s.solve_batch_host(pk.default_params(**kw), goal, seed, lambda q_, pose: 0.0, rng_seed=5, problem_offset=3)
        assert s.kernel_name(pk.default_params(**kw)).startswith("pik_exact::")
        for x, y, w in zip(gpu, host, ("solution", "status", "cost", "stats")):
            eq(x, y, f"{ch.name} {kw} kernels vs host: {w}")
    finally:
        s.close()


def test_the_cost_function_steers_the_search(O):
    ch = robots.panda()
    rng = np.random.default_rng(9)
    n = 16
    q = rng.uniform(ch.qmin, ch.qmax, size=(n, ch.dof))
    seed = np.tile(robots.PANDA_HOME, (n, 1))
    s = pk.Solver(ch, device=0)
    try:
        goal = s.fk(q)
        p = pk.default_params(memetic_population_size=32, memetic_max_generations=40, cost_threshold=0.02)
        # the Panda is redundant: ask for joint 2 near 0.5 rad (cost 0.5 (q2 - 0.5)^2 < 0.02^2 => within 0.028 rad)
        want = 0.5
        sol, st, _, _ = s.solve_batch_host(p, goal, seed, lambda q_, pose: 0.5 * (q_[2] - want) ** 2, rng_seed=1)
        ok = st == pk.SUCCESS
        assert ok.mean() >= 0.5, ok.mean()
        assert (np.abs(sol[ok, 2] - want) <= 0.0283).all()
        pose = s.fk(sol[ok])
        assert (np.linalg.norm(pose[:, :3] - goal[ok, :3], axis=1) <= 1e-3 * (1 + 1e-9)).all()
        # the plain query lands wherever the redundancy leaves it: hardly ever inside that window
        plain, pst, _, _ = s.solve_batch(p, goal, seed, rng_seed=1)
        inside = np.abs(plain[pst == pk.SUCCESS, 2] - want) <= 0.0283
        assert inside.mean() < 0.3
    finally:
        s.close()


def test_a_call_without_a_cost_function_is_refused(O):
    import ctypes as C
    ch = robots.panda()
    s = pk.Solver(ch, device=0)
    try:
        g = s.fk(robots.PANDA_HOME[None])
        with pytest.raises(pk.solver.PickIkAmdError, match="host cost function"):
            s.solve_batch_host(pk.default_params(), g, robots.PANDA_HOME[None], pk.solver.COST_FN(0))
    finally:
        s.close()
