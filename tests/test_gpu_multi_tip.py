"""GPU parity for several tip frames (the plugin's `tip_frames`; reference
src/pick_ik_plugin.cpp:57-69, src/goal.cpp:27-49, 80-89, src/robot.cpp:130-160): one pose cost and
one frame test per tip over one vector of active variables.

  * the multi-tip ORACLE is pinned against the single-chain oracle (itself pinned to the
    reference's known-answer tests): for independent arms its FK is each arm's FK and its cost is
    the sum of the arms' costs (tests/test_multi_tip_cpu.py does this without a GPU);
  * strict build: FK, cost/verdict, step(), ik_gradient and ik_memetic BIT-EXACT against the oracle
    (portable-math mode), tolerance zero, on a dual UR5 cell (12 variables, independent arms) and
    on a torso + two arms tree (9 variables, one joint moves both tips);
  * fast build: answers do not depend on the compaction marks, every SUCCESS passes the oracle's
    solution test for ALL tips, success rate comparable to the oracle's.
"""
import numpy as np
import pytest

import pick_ik_amd as pk
from pick_ik_amd import robots

pytestmark = pytest.mark.gpu
# other generated cases than the suite's: PIK_FUZZ_SEED_SHIFT=100000 pytest ... (soaks, profiles/r04_fuzz_soaks.txt)
SEED_SHIFT = int(__import__("os").environ.get("PIK_FUZZ_SEED_SHIFT", "0"))


def dual_ur5():
    return robots.side_by_side("dual_ur5", [robots.ur5(), robots.ur5()], [(0, 0.45, 0), (0, -0.45, 0)])


CHAINS = {"dual_ur5": dual_ur5, "torso_dual_arm": robots.torso_dual_arm}


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()


def eq(a, b, what=""):
    np.testing.assert_array_equal(a, b, err_msg=what)


def problems(O, ch, n, seed):
    rng = np.random.default_rng(seed)
    o = O.Oracle(ch)
    q = rng.uniform(ch.qmin, ch.qmax, size=(n, ch.dof))
    with O.math_mode("portable"):
        goal = o.fk(q)
    sd = rng.uniform(ch.qmin, ch.qmax, size=(n, ch.dof))
    near = rng.uniform(size=n) < 0.4
    sd[near] = np.clip(q[near] + rng.normal(0, 0.05, size=(int(near.sum()), ch.dof)), ch.qmin, ch.qmax)
    return o, q, goal, sd


@pytest.mark.parametrize("name", list(CHAINS))
def test_multi_tip_primitives_bit_exact(built, oracle_mod, name, exact_flavour):
    O = oracle_mod
    ch = CHAINS[name]()
    o, q, goal, sd = problems(O, ch, 300, 1)
    s = pk.Solver(ch, device=0, strict=True)
    try:
        assert s.n_tips == 2 and goal.shape == (300, 2, 7)
        kw = dict(center_joints_weight=0.3, avoid_joint_limits_weight=0.2, minimal_displacement_weight=0.1)
        with O.math_mode("portable"):
            eq(s.fk(q), o.fk(q), "fk")
            rng = np.random.default_rng(2)
            cand = q + rng.normal(0, 1, size=q.shape) * np.logspace(-6, -1, len(q))[:, None]
            for params in (dict(), kw):
                gc, gs = s.cost(pk.default_params(**params), goal, sd, cand)
                res = [o.cost(O.default_params(**params), goal[i], sd[i], cand[i]) for i in range(len(q))]
                eq(gc, np.array([r[0][0] for r in res]), "cost")
                eq(gs, np.array([r[1][0] for r in res]), "solution_fn")
                if not params:
                    assert 0 < gs.sum() < len(gs)  # candidates straddle the thresholds
            c0 = np.array([o.cost(O.default_params(), goal[i], sd[i], cand[i])[0][0] for i in range(len(q))])
            a = s.gd_step(pk.default_params(), goal, sd, cand, cand, c0, c0)
            b = o.gd_step(O.default_params(), goal, sd, cand, cand, c0, c0)
            for x, y, w in zip(a, b, ("local", "best", "local_cost", "best_cost", "gradient", "improved")):
                eq(x, y, w)
    finally:
        s.close()


@pytest.mark.parametrize("name", list(CHAINS))
def test_multi_tip_solvers_bit_exact(built, oracle_mod, name, monkeypatch, exact_flavour):
    O = oracle_mod
    ch = CHAINS[name]()
    o, q, goal, sd = problems(O, ch, 96, 3)
    s = pk.Solver(ch, device=0, strict=True)
    try:
        for kw in (dict(mode=1), dict(mode=1, return_approximate_solution=1, gd_max_iters=30),
                   dict(memetic_population_size=24, memetic_max_generations=12),
                   dict(memetic_population_size=40, memetic_elite_size=3, memetic_max_generations=8,
                        center_joints_weight=0.01, minimal_displacement_weight=0.001, cost_threshold=0.05),
                   dict(memetic_population_size=20, memetic_num_threads=2, memetic_max_generations=6)):
            for marks in ("none", "1,2,3,5,8"):
                if kw.get("mode") == 1 and marks != "none":
                    continue
                monkeypatch.setenv("PIK_PASSES", marks)
                with O.math_mode("portable"):
                    a = s.solve_batch(pk.default_params(**kw), goal, sd, rng_seed=11, problem_offset=7)
                    b = o.solve_batch(O.default_params(**kw), goal, sd, rng_seed=11, problem_offset=7,
                                      num_threads=O.max_threads())
                for x, y, w in zip(a, b, ("solution", "status", "cost", "stats")):
                    eq(x, y, f"{name} {kw} marks {marks}: {w}")
            assert (a[1] == pk.SUCCESS).any()
    finally:
        s.close()


@pytest.mark.parametrize("name", list(CHAINS))
def test_multi_tip_memoised_descent_lanes_and_goals(built, oracle_mod, name, monkeypatch, exact_flavour):
    """the memoised descent for several tip frames (pik_exact.hpp gradient_descent_exact_multi: the probes of a variable
    fork off the walk of every path that contains it, the other tips' pose costs are the accept evaluation's) against
    the oracle, tolerance zero: one and two lanes per elite, with and without compaction passes, all three joint goals
    on, step() with joint goals, ik_gradient"""
    O = oracle_mod
    ch = CHAINS[name]()
    o, q, goal, sd = problems(O, ch, 64, 5)
    s = pk.Solver(ch, device=0, strict=True)
    goals = dict(center_joints_weight=0.05, avoid_joint_limits_weight=0.02, minimal_displacement_weight=0.01,
                 cost_threshold=0.3)
    try:
        with O.math_mode("portable"):
            rng = np.random.default_rng(3)
            cand = q + rng.normal(0, 1, size=q.shape) * np.logspace(-5, -1, len(q))[:, None]
            c0 = np.array([o.cost(O.default_params(**goals), goal[i], sd[i], cand[i])[0][0] for i in range(len(q))])
            a = s.gd_step(pk.default_params(**goals), goal, sd, cand, cand, c0, c0)
            b = o.gd_step(O.default_params(**goals), goal, sd, cand, cand, c0, c0)
            for x, y, w in zip(a, b, ("local", "best", "local_cost", "best_cost", "gradient", "improved")):
                eq(x, y, f"{name} step() with joint goals: {w}")
        for kw in (dict(memetic_population_size=16, memetic_max_generations=6, **goals),
                   dict(memetic_population_size=12, memetic_elite_size=2, memetic_max_generations=5),
                   dict(mode=1, gd_max_iters=25, **goals)):
            with O.math_mode("portable"):
                b = o.solve_batch(O.default_params(**kw), goal, sd, rng_seed=5, problem_offset=1, num_threads=O.max_threads())
            for lpe, marks in (("1", "none"), ("2", "none"), ("1", "1,2,3"), ("2", "1,3"), ("0", "none")):
                if kw.get("mode") == 1 and (marks != "none" or lpe == "2"):
                    continue
                monkeypatch.setenv("PIK_LPE", lpe)
                monkeypatch.setenv("PIK_PASSES", marks)
                with O.math_mode("portable"):
                    a = s.solve_batch(pk.default_params(**kw), goal, sd, rng_seed=5, problem_offset=1)
                for x, y, w in zip(a, b, ("solution", "status", "cost", "stats")):
                    eq(x, y, f"{name} {kw} lanes {lpe} marks {marks}: {w}")
    finally:
        s.close()


@pytest.mark.parametrize("name", list(CHAINS))
def test_multi_tip_fast_build(built, oracle_mod, name, monkeypatch):
    O = oracle_mod
    ch = CHAINS[name]()
    o, q, goal, sd = problems(O, ch, 256, 5)
    home = np.clip(np.zeros(ch.dof), ch.qmin, ch.qmax)
    seed = np.tile(home, (256, 1))
    s = pk.Solver(ch, device=0, exact=False)
    try:
        # primitives to rounding
        f, of = s.fk(q), o.fk(q)
        np.testing.assert_allclose(f[..., :3], of[..., :3], rtol=0, atol=1e-12)
        sgn = np.sign((f[..., 3:] * of[..., 3:]).sum(axis=-1, keepdims=True))
        np.testing.assert_allclose(f[..., 3:] * sgn, of[..., 3:], rtol=0, atol=1e-12)
        gc, _ = s.cost(pk.default_params(), goal, sd, sd)
        oc = np.array([o.cost(O.default_params(), goal[i], sd[i], sd[i])[0][0] for i in range(256)])
        np.testing.assert_allclose(gc, oc, rtol=1e-11, atol=1e-18)
        # one step(): the frame-based probes of every tip against literal central differences
        a = s.gd_step(pk.default_params(), goal, sd, sd, sd, oc, oc)
        b = o.gd_step(O.default_params(), goal, sd, sd, sd, oc, oc)
        scale = np.abs(b[4]).max(axis=1, keepdims=True) + 1e-300
        assert (np.abs(a[4] - b[4]) / scale).max() < 1e-6
        np.testing.assert_allclose(a[0], b[0], rtol=0, atol=1e-9)
        # whole solves
        p = pk.default_params(memetic_population_size=64)
        outs = []
        # compaction marks x lanes per elite (1; 2: the line-search pair; 8 / 16: the cooperative descent for
        # several tips; None = adaptive)
        for marks, lpe in (("none", "1"), ("1,2,4,7", "1"), ("2,4,8,16,32,64", "1"), ("none", "2"),
                           ("1,2,4,7", "2"), ("none", "8"), ("1,2,4,7", "16"), ("none", "16"), ("2,3", "8"),
                           ("2,4,8,16,32,64", None), (None, None)):
            for var, val in (("PIK_PASSES", marks), ("PIK_LPE", lpe)):
                if val is None:
                    monkeypatch.delenv(var, raising=False)
                else:
                    monkeypatch.setenv(var, val)
            outs.append(s.solve_batch(p, goal, seed, rng_seed=21))
        monkeypatch.delenv("PIK_LPE", raising=False)
        for other in outs[1:]:
            for x, y, w in zip(outs[0], other, ("solution", "status", "cost", "stats")):
                eq(x, y, f"{name}: {w} depends on the compaction marks / lanes per elite")
        # ... and with species (two populations per problem sharing the wavefront)
        ps = pk.default_params(memetic_population_size=32, memetic_num_threads=2, memetic_max_generations=20)
        souts = []
        for marks, lpe in (("none", "1"), ("1,2,4,7", "2"), ("none", "8"), ("2,3", "8"), (None, None)):
            s.set_option("passes", marks)
            s.set_option("lanes_per_elite", lpe)
            souts.append(s.solve_batch(ps, goal, seed, rng_seed=22))
        s.set_option("passes", None)
        s.set_option("lanes_per_elite", None)
        for other in souts[1:]:
            for x, y, w in zip(souts[0], other, ("solution", "status", "cost", "stats")):
                eq(x, y, f"{name}, two species: {w} depends on the compaction marks / lanes per elite")
        assert (souts[0][1] == pk.SUCCESS).mean() > 0.5
        sol, st, cost, _ = outs[0]
        ob = o.solve_batch(O.default_params(memetic_population_size=64), goal, seed, rng_seed=21,
                           num_threads=O.max_threads())
        ok = st == pk.SUCCESS
        assert ok.mean() >= 0.97 * (ob[1] == 1).mean() and ok.mean() > 0.5, (ok.mean(), (ob[1] == 1).mean())
        op = O.default_params(memetic_population_size=64)
        tips = o.fk(sol[ok])
        assert np.abs(tips[..., :3] - goal[ok][..., :3]).max() <= 1.0e-3 + 1e-12
        for i in np.flatnonzero(ok)[:64]:
            c, is_sol = o.cost(op, goal[i], seed[i], sol[i])
            assert is_sol[0] == 1 and abs(c[0] - cost[i]) <= 1e-9 * max(1.0, c[0])
        eq(sol[st == pk.NO_IK_SOLUTION], seed[st == pk.NO_IK_SOLUTION])
    finally:
        s.close()


def test_multi_tip_bad_descriptions(built):
    ch = robots.torso_dual_arm()
    import dataclasses
    # a variable that moves no tip
    lone = dataclasses.replace(ch, qmin=np.append(ch.qmin, -1.0), qmax=np.append(ch.qmax, 1.0),
                               vmax=np.append(ch.vmax, 1.0), bounded=np.append(ch.bounded, 1).astype(np.uint8))
    with pytest.raises(pk.PickIkAmdError, match="no tip"):
        pk.Solver(lone)
    # variable indices must increase along a path
    t0 = ch.tips[0]
    bad = dataclasses.replace(ch, tips=(dataclasses.replace(t0, variable=t0.variable[::-1].copy()), ch.tips[1]))
    with pytest.raises(pk.PickIkAmdError, match="increasing"):
        pk.Solver(bad)


def random_tree(rng):
    """2..8 tips over 3..16 variables: a shared prefix of 0..2 joints, the other variables dealt to
    the tips in interleaved order (so a tip's variables are not contiguous); arbitrary origins and
    axes, a few prismatic / continuous joints."""
    n_tips = int(rng.integers(2, 9))
    shared = int(rng.integers(0, 3))
    dof = int(rng.integers(max(shared + n_tips, 3), 17))
    owner = np.concatenate([np.arange(n_tips), rng.integers(0, n_tips, size=dof - shared - n_tips)])
    rng.shuffle(owner)
    jt_all = (rng.uniform(size=dof) < 0.15).astype(np.int32)
    org_all = np.concatenate([rng.uniform(-0.3, 0.3, size=(dof, 3)), rng.uniform(-np.pi, np.pi, size=(dof, 3))], axis=1)
    ax_all = rng.normal(size=(dof, 3))
    paths = []
    for k in range(n_tips):
        var = list(range(shared)) + [shared + i for i in range(dof - shared) if owner[i] == k]
        org = org_all[var].copy()
        if shared and k:  # a branch: its first own joint hangs off the shared part differently
            org[shared:shared + 1] += 0.05 * k
        tip = np.concatenate([rng.uniform(-0.2, 0.2, size=3), rng.uniform(-np.pi, np.pi, size=3)])
        paths.append((var, org, ax_all[var], jt_all[var], tip))
    bounded = np.where(jt_all == 1, 1, rng.uniform(size=dof) < 0.85).astype(np.uint8)
    span = np.where(jt_all == 1, rng.uniform(0.05, 0.3, size=dof), rng.uniform(0.5, 3.0, size=dof))
    mid = rng.uniform(-0.3, 0.3, size=dof)
    return robots.multi_chain("tree", paths, mid - span, mid + span, rng.uniform(0.5, 3.0, size=dof), bounded)


@pytest.mark.parametrize("i", range(int(__import__("os").environ.get("PIK_FUZZ_TREES", "16"))))
def test_multi_tip_random_trees_bit_exact(built, oracle_mod, i, monkeypatch, exact_flavour):
    O = oracle_mod
    rng = np.random.default_rng(0x7EE + i + SEED_SHIFT)
    ch = random_tree(rng)
    o = O.Oracle(ch)
    lo = np.where(ch.bounded == 1, ch.qmin, -3.0)
    hi = np.where(ch.bounded == 1, ch.qmax, 3.0)
    B = int(rng.integers(8, 80))
    q = rng.uniform(lo, hi, size=(B, ch.dof))
    sd = rng.uniform(lo, hi, size=(B, ch.dof))
    near = rng.uniform(size=B) < 0.5
    sd[near] = np.clip(q[near] + rng.normal(0, 0.03, size=(int(near.sum()), ch.dof)), lo, hi)
    kw = dict(memetic_population_size=int(rng.integers(6, 40)), memetic_elite_size=int(rng.choice([1, 2, 4])),
              memetic_max_generations=int(rng.integers(2, 12)), position_threshold=1e-2, orientation_threshold=1e-2,
              return_approximate_solution=int(rng.uniform() < 0.3))
    if rng.uniform() < 0.4:
        kw.update(center_joints_weight=0.02, avoid_joint_limits_weight=0.05, cost_threshold=0.2)
    if i % 4 == 3:
        kw = dict(mode=1, gd_max_iters=40)
    s = pk.Solver(ch, device=0, strict=True)
    f = pk.Solver(ch, device=0, exact=False)
    try:
        monkeypatch.setenv("PIK_PASSES", "1,2,4" if i % 2 else "none")
        with O.math_mode("portable"):
            goal = o.fk(q)
            eq(s.fk(q), goal, f"tree {i}: fk")
            a = s.solve_batch(pk.default_params(**kw), goal, sd, rng_seed=i, problem_offset=3)
            b = o.solve_batch(O.default_params(**kw), goal, sd, rng_seed=i, problem_offset=3,
                              num_threads=O.max_threads())
        for x, y, w in zip(a, b, ("solution", "status", "cost", "stats")):
            eq(x, y, f"tree {i} ({ch.n_tips} tips, {ch.dof} variables) {kw}: {w}")
        # fast build: every SUCCESS is a solution for all tips by the oracle's test ...
        sol, st, cost, _ = f.solve_batch(pk.default_params(**kw), goal, sd, rng_seed=i, problem_offset=3)
        for j in np.flatnonzero(st == pk.SUCCESS)[:20]:
            assert o.cost(O.default_params(**kw), goal[j], sd[j], sol[j])[1][0] == 1
        # ... and the answer does not depend on the lanes per elite (8 / 16: the cooperative descent for
        # several tips -- served when every tip's chain is a plain Denavit-Hartenberg one and the elites fit)
        ref = None
        for lanes, marks in (("1", "none"), ("2", "1,2,4"), ("8", "none"), ("16", "1,3"), ("8", "2,3"), (None, None)):
            f.set_option("lanes_per_elite", lanes)
            f.set_option("passes", marks)
            out = f.solve_batch(pk.default_params(**kw), goal, sd, rng_seed=i, problem_offset=3)
            if ref is None:
                ref = out
            for x, y, w in zip(ref, out, ("solution", "status", "cost", "stats")):
                eq(x, y, f"tree {i} ({ch.n_tips} tips, {ch.dof} variables) {kw} lanes {lanes} marks {marks}: {w}")
    finally:
        s.close()
        f.close()
