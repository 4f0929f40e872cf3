"""GPU BIT-EXACT parity: the strict-arithmetic build of the kernels (same sources compiled with
-DPIK_STRICT -ffp-contract=off: no FMA contraction, generic joint rotations) against the CPU oracle
in its portable-math mode (same sincos/atan2 algorithm as the device, everything else IEEE
+,-,*,/,sqrt).

The gradient descent of this algorithm is chaotic (a 1e-15 perturbation of a seed moves the
oracle's own answer by 1e-4 rad, see tests/test_gpu_parity.py), so agreement "to a tolerance"
cannot be asked of whole solves; agreement BIT FOR BIT can, and it checks every piece of control
flow the kernels implement: Philox streams and slot layout, the speculative mating-pool rounds,
the running top-E selection, extinction factors, wipeouts, early exits, the literal evaluation
counters.  Tolerance in this file: exactly zero.
"""
import numpy as np
import pytest

import pick_ik_amd as pk
from pick_ik_amd import robots
from tests.common import EXACT_CONFIGS, golden, random_targets

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("exact_flavour")]  # (both exact builds, see conftest)


@pytest.fixture(scope="module")
def O(oracle_mod):
    return oracle_mod


@pytest.fixture(scope="module")
def solvers():
    import __graft_entry__ as g
    g.build()
    cache = {}

    def get(name):
        key = (name, pk.Solver)  # (pk.Solver is the exact-kernels class under the "fma" flavour)
        if key not in cache:
            cache[key] = pk.Solver(robots.by_name(name), device=0, strict=True)
        return cache[key]

    yield get
    for s in cache.values():
        s.close()


def eq(a, b, what=""):
    np.testing.assert_array_equal(a, b, err_msg=what)


@pytest.mark.parametrize("name", ["panda", "ur5", "rr"])
def test_primitives_bit_exact(solvers, O, name):
    s = solvers(name)
    ch = s.chain
    o = O.Oracle(ch)
    G = golden()
    rng = np.random.default_rng(1)
    q = np.concatenate([G[f"fk_{name}_q"], rng.uniform(ch.qmin, ch.qmax, size=(2000, ch.dof)),
                        rng.uniform(-40, 40, size=(200, ch.dof))])
    with O.math_mode("portable"):
        eq(s.fk(q), o.fk(q), "fk")
        kw = dict(center_joints_weight=0.3, avoid_joint_limits_weight=0.2,
                  minimal_displacement_weight=0.1)
        goal, seed, qq = G[f"cost_{name}_goal"], G[f"cost_{name}_seed"], G[f"fk_{name}_q"]
        gc, gs = s.cost(pk.default_params(**kw), goal, seed, qq)
        oc = np.array([o.cost(O.default_params(**kw), goal[i], seed[i], qq[i])[0][0]
                       for i in range(len(qq))])
        eq(gc, oc, "cost")
        # candidates straddling the solution thresholds
        qs = rng.uniform(ch.qmin, ch.qmax, size=(512, ch.dof))
        g2 = o.fk(qs)
        cand = qs + rng.normal(0, 1, size=qs.shape) * np.logspace(-6, -2, 512)[:, None]
        gc, gs = s.cost(pk.default_params(), g2, qs, cand)
        res = [o.cost(O.default_params(), g2[i], qs[i], cand[i]) for i in range(512)]
        eq(gc, np.array([r[0][0] for r in res]), "cost near goal")
        eq(gs, np.array([r[1][0] for r in res]), "solution_fn")
        # step()
        c0 = np.array([o.cost(O.default_params(), goal[i], seed[i], qq[i])[0][0]
                       for i in range(len(qq))])
        a = s.gd_step(pk.default_params(), goal, seed, qq, qq, c0, c0)
        b = o.gd_step(O.default_params(), goal, seed, qq, qq, c0, c0)
        for x, y, w in zip(a, b, ("local", "best", "local_cost", "best_cost", "gradient", "improved")):
            eq(x, y, w)


@pytest.mark.parametrize("name", ["panda", "ur5", "rr"])
def test_ik_gradient_bit_exact(solvers, O, name):
    s = solvers(name)
    ch = s.chain
    o = O.Oracle(ch)
    rng = np.random.default_rng(9)
    qs, goal = random_targets(o.fk, ch, rng, 300)
    seeds = rng.uniform(ch.qmin, ch.qmax, size=(300, ch.dof))  # far seeds: long, chaotic descents
    seeds[:100] = np.clip(qs[:100] + rng.normal(0, 0.05, size=(100, ch.dof)), ch.qmin, ch.qmax)
    for kw in (dict(mode=1), dict(mode=1, return_approximate_solution=1),
               dict(mode=1, stop_optimization_on_valid_solution=0, gd_max_iters=40)):
        with O.math_mode("portable"):
            b = o.solve_batch(O.default_params(**kw), goal, seeds, num_threads=O.max_threads())
            # adaptive (300 problems: sixteen lanes per problem, the team kernel of local mode), one lane, and both
            # team kernels forced
            for lanes in (None, 1, 4, 16):
                s.set_option("lanes_per_elite", lanes)
                try:
                    a = s.solve_batch(pk.default_params(**kw), goal, seeds)
                finally:
                    s.set_option("lanes_per_elite", None)
                for x, y, w in zip(a, b, ("solution", "status", "cost", "stats")):
                    eq(x, y, f"{name} {kw} lanes per problem {lanes}: {w}")
        assert (a[1] == 1).sum() > 20


def run_both(O, s, kw, goal, seed, rng_seed, offset=0, lanes=(None, 1)):
    """the strict build under the adaptive schedule (small calls: the widest variant the elite count
    allows, the 2D literal probes of a step dealt out to the lanes of an elite) AND with one lane per
    elite, each against the oracle"""
    with O.math_mode("portable"):
        b = O.Oracle(s.chain).solve_batch(O.default_params(**kw), goal, seed, rng_seed=rng_seed,
                                          problem_offset=offset, num_threads=O.max_threads())
        for lpe in lanes:
            s.set_option("lanes_per_elite", lpe)
            try:
                a = s.solve_batch(pk.default_params(**kw), goal, seed, rng_seed=rng_seed, problem_offset=offset)
            finally:
                s.set_option("lanes_per_elite", None)
            for x, y, w in zip(a, b, ("solution", "status", "cost", "stats")):
                eq(x, y, f"{kw} lanes_per_elite={lpe} {w}")
    return a


@pytest.mark.parametrize("lpe", [1, 2, 4, 8, 16])
def test_memetic_lanes_per_elite_bit_exact(solvers, O, lpe):
    """Every kernel variant of the strict build against the oracle directly: 1..16 lanes per elite, with
    and without compaction passes, reachable and unreachable targets (the latter run all generations:
    the tail the wide variants exist for)."""
    s = solvers("panda")
    o = O.Oracle(s.chain)
    rng = np.random.default_rng(1000 + lpe)
    _, goal = random_targets(o.fk, s.chain, rng, 72)
    goal[60:] = random_targets(o.fk, s.chain, rng, 12, unreachable=True)[1]
    seed = np.tile(robots.PANDA_HOME, (72, 1))
    seed[::5] = rng.uniform(s.chain.qmin, s.chain.qmax, size=seed[::5].shape)
    for marks in ("none", "1,2,4,7,11", None):
        s.set_option("passes", marks)
        try:
            for kw in (dict(memetic_population_size=32, memetic_max_generations=16),
                       dict(memetic_population_size=20, memetic_elite_size=2, memetic_max_generations=12,
                            minimal_displacement_weight=0.01, center_joints_weight=0.02, cost_threshold=0.05)):
                if lpe * (1 if kw.get("memetic_elite_size", 4) == 4 else 1) > 16:
                    continue
                a = run_both(O, s, kw, goal, seed, rng_seed=4242 + lpe, offset=99, lanes=(lpe,))
        finally:
            s.set_option("passes", None)
    assert (a[1] == pk.SUCCESS).sum() > 10 and (a[1] != pk.SUCCESS).sum() > 0


@pytest.mark.parametrize("cname", list(EXACT_CONFIGS))
def test_memetic_configs_bit_exact(solvers, O, cname):
    robot, home, kw = EXACT_CONFIGS[cname]
    s = solvers(robot)
    o = O.Oracle(s.chain)
    rng = np.random.default_rng(sum(map(ord, cname)))
    n = 96
    _, goal = random_targets(o.fk, s.chain, rng, n, unreachable=(cname == "panda_approx"))
    seed = np.tile(home, (n, 1))
    sol, st, c, stats = run_both(O, s, kw, goal, seed, rng_seed=0xC0FFEE)
    if cname == "panda_approx":
        assert (st == pk.APPROXIMATE).any()
    else:
        assert (st == pk.SUCCESS).mean() > (0.7 if robot == "ur5" else 0.9)
    assert stats["wipeouts"].sum() > 0 and stats["pool_erasures"].sum() > 0


@pytest.mark.parametrize("B,P,E", [(256, 16, 4), (100, 128, 4), (37, 24, 1), (64, 20, 2),
                                   (48, 17, 3), (40, 33, 5), (33, 40, 8), (9, 80, 16),
                                   (5, 200, 32), (3, 130, 64)])
def test_memetic_shapes_bit_exact(solvers, O, B, P, E):
    s = solvers("panda")
    rng = np.random.default_rng(B * 1000 + P)
    _, goal = random_targets(O.Oracle(s.chain).fk, s.chain, rng, B)
    seed = np.tile(robots.PANDA_HOME, (B, 1))
    seed[::3] = rng.uniform(s.chain.qmin, s.chain.qmax, size=seed[::3].shape)
    run_both(O, s, dict(memetic_population_size=P, memetic_elite_size=E,
                        memetic_max_generations=30), goal, seed, rng_seed=B, offset=12345)


def test_memetic_variants_bit_exact(solvers, O):
    """parameter corners: keep optimizing after a valid solution, tiny budgets, all goals on,
    position-only / orientation-only costs, huge problem offsets (64-bit stream keys)."""
    s = solvers("panda")
    o = O.Oracle(s.chain)
    rng = np.random.default_rng(31)
    _, goal = random_targets(o.fk, s.chain, rng, 48)
    seed = np.tile(robots.PANDA_HOME, (48, 1))
    for kw in (dict(stop_optimization_on_valid_solution=0, memetic_max_generations=4),
               dict(memetic_gd_max_iters=1, memetic_max_generations=20),
               dict(memetic_gd_max_iters=0, memetic_max_generations=20),
               dict(center_joints_weight=0.02, avoid_joint_limits_weight=0.05,
                    minimal_displacement_weight=0.01, cost_threshold=0.05),
               dict(rotation_scale=0.0), dict(position_scale=0.0),
               dict(memetic_wipeout_fitness_tol=1e-2), dict(gd_step_size=1e-3)):
        run_both(O, s, kw, goal, seed, rng_seed=77)
    run_both(O, s, {}, goal, seed, rng_seed=(1 << 63) + 12345, offset=(1 << 40) + 7)
    sr = solvers("rr")
    _, goal = random_targets(O.Oracle(sr.chain).fk, sr.chain, rng, 40)
    run_both(O, sr, dict(memetic_population_size=12, memetic_elite_size=3), goal,
             np.zeros((40, 2)), rng_seed=5)


def _continuous_chain():
    """UR5 with continuous wrist joints and a Panda with two continuous joints: MoveIt reports
    position_bounded_ = false for continuous joints (reference src/robot.cpp:61-65)."""
    import dataclasses
    ur = robots.ur5()
    ur = dataclasses.replace(ur, bounded=np.array([1, 1, 1, 0, 0, 0], np.uint8))
    pa = robots.panda()
    pa = dataclasses.replace(pa, bounded=np.array([0, 1, 1, 1, 1, 1, 0], np.uint8))
    return ur, pa


@pytest.mark.parametrize("which", [0, 1])
def test_memetic_unbounded_variables_bit_exact(O, which, monkeypatch):
    """Continuous (unbounded) variables: random restarts are centred on the current value
    (src/robot.cpp:26-28) and, when the mating pool runs empty, on the stale rank-i individual of the
    previous generation (src/ik_memetic.cpp:181-184) -- the kernel keeps the population in HBM for
    these chains.  Small pools so that the empty-pool branch is exercised constantly."""
    import __graft_entry__ as g
    g.build()
    ch = _continuous_chain()[which]
    s = pk.Solver(ch, strict=True)
    o = O.Oracle(ch)
    rng = np.random.default_rng(17 + which)
    lo = np.where(ch.bounded == 1, ch.qmin, -3.0)
    hi = np.where(ch.bounded == 1, ch.qmax, 3.0)
    q = rng.uniform(lo, hi, size=(120, ch.dof))
    goal = o.fk(q)
    seed = rng.uniform(lo, hi, size=(120, ch.dof))
    for kw in (dict(memetic_population_size=24, memetic_elite_size=2, memetic_max_generations=25),
               dict(memetic_population_size=40, memetic_elite_size=4, memetic_max_generations=20),
               dict(memetic_population_size=16, memetic_elite_size=1, memetic_max_generations=15),
               dict(mode=1)):
        for marks in ("none", "1,2,3,5,8,13"):
            monkeypatch.setenv("PIK_PASSES", marks)
            a = run_both(O, s, kw, goal, seed, rng_seed=21, offset=5)
        if kw.get("mode") != 1:
            assert a[3]["pool_erasures"].sum() > 0
    s.close()


@pytest.mark.parametrize("S", [2, 3, 4])
def test_memetic_species_bit_exact(solvers, O, S):
    """memetic_num_threads > 1: species in lock-step, stop-on-first coupling, minimum-fitness
    selection over the species that returned a value (src/ik_memetic.cpp:312-371)."""
    s = solvers("panda")
    o = O.Oracle(s.chain)
    rng = np.random.default_rng(100 + S)
    _, goal = random_targets(o.fk, s.chain, rng, 90)
    goal[60:] = random_targets(o.fk, s.chain, rng, 30, unreachable=True)[1]
    goal[60:, :3] *= 1.5
    seed = np.tile(robots.PANDA_HOME, (90, 1))
    seed[::4] = rng.uniform(s.chain.qmin, s.chain.qmax, size=seed[::4].shape)
    for first in (1, 0):
        for approx in (0, 1):
            kw = dict(memetic_num_threads=S, memetic_stop_on_first_solution=first,
                      return_approximate_solution=approx, memetic_max_generations=9,
                      memetic_population_size=20)
            # with and without compaction passes: several species park and resume as one problem
            # (the default marks 2, 4, 8 cut this 9-generation budget as well)
            for marks in ("none", "1,2,3,5,8", None):
                s.set_option("passes", marks)
                try:
                    # (and with more lanes per elite: all species of a problem share the wavefront)
                    a = run_both(O, s, kw, goal, seed, rng_seed=S * 10 + first, offset=3, lanes=(None, 1, 2, 4))
                finally:
                    s.set_option("passes", None)
            st = a[1]
            assert (st == pk.SUCCESS).any()
            assert ((st == pk.APPROXIMATE).any() if approx else (st == pk.NO_IK_SOLUTION).any())
    # a seed that already solves the goal: returned untouched, nothing evaluated
    home = robots.PANDA_HOME
    a = run_both(O, s, dict(memetic_num_threads=S), o.fk(home), home[None], rng_seed=1)
    assert a[1][0] == pk.SUCCESS and a[3]["cost_evals"][0] == 0
    # elite counts that are not powers of two, 8 species x 8-lane groups fill the wavefront
    run_both(O, s, dict(memetic_num_threads=S, memetic_elite_size=3, memetic_population_size=18,
                        memetic_max_generations=6), goal[:40], seed[:40], rng_seed=9)


@pytest.mark.parametrize("kw", [
    dict(memetic_population_size=24, minimal_displacement_weight=0.05, cost_threshold=0.05),
    dict(memetic_population_size=16, memetic_max_generations=4),
    dict(mode=1, gd_max_iters=40, minimal_displacement_weight=0.02, cost_threshold=0.05),
], ids=["memetic_displacement", "memetic_short", "local"])
def test_initial_guess_separate_from_seed_bit_exact(solvers, O, kw):
    """The plugin's two joint vectors (src/pick_ik_plugin.cpp:199-245): ik_seed_state stays the
    minimal-displacement reference and the vector returned on failure, init_state (re-randomised on
    restarts) is where the search starts.  Strict build vs oracle, bit for bit, including problems
    whose guess already is a solution and problems that fail."""
    s = solvers("panda")
    ch = s.chain
    o = O.Oracle(ch)
    rng = np.random.default_rng(21)
    n = 160
    q = rng.uniform(ch.qmin, ch.qmax, size=(n, ch.dof))
    seed = np.clip(q + rng.normal(0, 0.2, size=q.shape), ch.qmin, ch.qmax)
    guess = rng.uniform(ch.qmin, ch.qmax, size=(n, ch.dof))
    guess[:8] = q[:8]  # the guess is already a solution: returned untouched
    with O.math_mode("portable"):
        goal = o.fk(q)
        goal[8:16, :3] = [2.5, -2.0, 3.0]  # unreachable: failure returns the SEED, cost of the guess
        a = s.solve_batch(pk.default_params(**kw), goal, seed, rng_seed=3, problem_offset=11,
                          initial_guess=guess)
        b = o.solve_batch(O.default_params(**kw), goal, seed, rng_seed=3, problem_offset=11,
                          num_threads=O.max_threads(), initial_guess=guess)
    for x, y, w in zip(a, b, ("solution", "status", "cost", "stats")):
        eq(x, y, w)
    eq(a[0][:8], guess[:8], "early accept returns the guess")
    fail = a[1] == pk.NO_IK_SOLUTION
    assert fail[8:16].all()
    eq(a[0][fail], seed[fail], "failure returns ik_seed_state")
    assert (a[1] == pk.SUCCESS).sum() >= 9
