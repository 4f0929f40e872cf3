"""GPU parity fuzz: randomly generated serial chains (1..16 variables, arbitrary axes and origins,
prismatic, continuous and planar joints mixed in) x randomly drawn solver parameters.

  strict build : whole solves BIT-EXACT against the oracle (portable-math mode), tolerance zero;
  fast build   : the answer does not depend on the execution shape (lanes per elite, compaction
                 marks) -- bit-identical across shapes -- and every SUCCESS is a real solution by
                 the oracle's own solution test (reference src/goal.cpp:163-186).

The draws are seeded: a failure names its case index and reproduces.
"""
import numpy as np
import pytest

import pick_ik_amd as pk
from pick_ik_amd import robots

pytestmark = pytest.mark.gpu
# other generated cases than the suite's: PIK_FUZZ_SEED_SHIFT=100000 pytest ... (soaks, profiles/r04_fuzz_soaks.txt)
SEED_SHIFT = int(__import__("os").environ.get("PIK_FUZZ_SEED_SHIFT", "0"))

import os

N_CASES = int(os.environ.get("PIK_FUZZ_CASES", "40"))  # more cases: PIK_FUZZ_CASES=200 pytest ...


def random_chain(rng, dof):
    origins = np.zeros((dof, 6))
    origins[:, :3] = rng.uniform(-0.35, 0.35, size=(dof, 3))
    origins[:, 3:] = rng.uniform(-np.pi, np.pi, size=(dof, 3))
    style = rng.integers(0, 3)
    if style == 0:  # all joints about z (parallel axes: the degenerate case of the DH construction)
        axes = np.tile([0.0, 0.0, 1.0], (dof, 1))
    elif style == 1:  # principal axes, both signs
        axes = np.eye(3)[rng.integers(0, 3, size=dof)] * rng.choice([-1.0, 1.0], size=(dof, 1))
    else:  # arbitrary, not normalised (the library normalises, like urdfdom/MoveIt)
        axes = rng.normal(size=(dof, 3)) * rng.uniform(0.5, 2.0, size=(dof, 1))
    jt = (rng.uniform(size=dof) < 0.2).astype(np.int32)
    bounded = np.where(jt == 1, 1, rng.uniform(size=dof) < 0.85).astype(np.uint8)
    span = np.where(jt == 1, rng.uniform(0.05, 0.4, size=dof), rng.uniform(0.5, 3.1, size=dof))
    mid = np.where(jt == 1, rng.uniform(-0.1, 0.1, size=dof), rng.uniform(-0.5, 0.5, size=dof))
    qmin, qmax = mid - span, mid + span
    tip = np.concatenate([rng.uniform(-0.2, 0.2, size=3), rng.uniform(-np.pi, np.pi, size=3)])
    vmax = rng.uniform(0.5, 3.0, size=dof)
    # a planar joint (x, y, theta: three consecutive variables) in a third of the chains that have
    # room for one -- drawn from a side stream so that the other chains stay what they were
    side = np.random.default_rng(int(abs(origins[0, 0]) * 1e12) % (1 << 32))
    if dof >= 3 and side.uniform() < 0.34:
        k = int(side.integers(0, dof - 2))
        jt[k:k + 3] = [robots.PLANAR_X, robots.PLANAR_Y, robots.PLANAR_THETA]
        bounded[k:k + 3] = side.uniform(size=3) < 0.7
        s3 = np.array([side.uniform(0.05, 0.4), side.uniform(0.05, 0.4), side.uniform(0.5, 3.1)])
        m3 = np.array([side.uniform(-0.1, 0.1), side.uniform(-0.1, 0.1), side.uniform(-0.5, 0.5)])
        qmin[k:k + 3], qmax[k:k + 3] = m3 - s3, m3 + s3
    return robots._chain(f"fuzz{dof}", origins, axes, tip, qmin, qmax, vmax, bounded=bounded,
                         joint_type=jt)


def random_params(rng):
    E = int(rng.choice([1, 2, 3, 4, 4, 4, 5, 8, 16]))
    P = int(E + 1 + rng.integers(0, 60))
    kw = dict(memetic_population_size=P, memetic_elite_size=E,
              memetic_max_generations=int(rng.integers(1, 30)),
              memetic_gd_max_iters=int(rng.choice([0, 1, 5, 25, 25])),
              gd_step_size=float(rng.choice([1e-4, 1e-3, 1e-5])),
              gd_min_cost_delta=float(rng.choice([1e-12, 1e-9, 1e-6])),
              memetic_wipeout_fitness_tol=float(rng.choice([1e-5, 1e-3, 1e-8])),
              position_threshold=float(rng.choice([1e-3, 1e-2, 1e-4])),
              orientation_threshold=float(rng.choice([1e-3, 1e-2, 1e-4])),
              position_scale=float(rng.choice([1.0, 1.0, 0.5, 2.0])),
              rotation_scale=float(rng.choice([0.5, 0.5, 1.0, 0.0])),
              stop_optimization_on_valid_solution=int(rng.uniform() < 0.8),
              return_approximate_solution=int(rng.uniform() < 0.3))
    if rng.uniform() < 0.4:
        kw.update(center_joints_weight=float(rng.choice([0.0, 0.01, 0.1])),
                  avoid_joint_limits_weight=float(rng.choice([0.0, 0.02, 0.2])),
                  minimal_displacement_weight=float(rng.choice([0.0, 0.001, 0.05])),
                  cost_threshold=float(rng.choice([1e-3, 0.05, 1.0])))
    r = rng.uniform()
    if r < 0.15:
        kw["mode"] = 1
        kw["gd_max_iters"] = int(rng.choice([5, 40, 100]))
    elif r < 0.35:
        S = int(rng.choice([2, 3, 4]))
        while (1 << (S - 1).bit_length()) * (1 << max(E - 1, 0).bit_length()) > 64:
            S -= 1
        if S > 1:
            kw["memetic_num_threads"] = S
            kw["memetic_stop_on_first_solution"] = int(rng.uniform() < 0.5)
    return kw


def make_case(i):
    rng = np.random.default_rng(0xF00D + i + SEED_SHIFT)
    # (cases 0..23 and every later multiple keep the chain lengths 1..12 they always had; 24..27 of every
    #  block of 28 are the long chains 13..16)
    ch = random_chain(rng, 1 + (i % 28) % 12 if i % 28 < 24 else 13 + (i % 28) - 24)
    kw = random_params(rng)
    B = int(rng.integers(1, 150))
    lo = np.where(ch.bounded == 1, ch.qmin, -3.0)
    hi = np.where(ch.bounded == 1, ch.qmax, 3.0)
    q = rng.uniform(lo, hi, size=(B, ch.dof))
    seed = rng.uniform(lo, hi, size=(B, ch.dof))
    near = rng.uniform(size=B) < 0.3
    seed[near] = np.clip(q[near] + rng.normal(0, 0.05, size=(int(near.sum()), ch.dof)), lo, hi)
    return ch, kw, q, seed, int(rng.integers(0, 1 << 62)), int(rng.integers(0, 1 << 40))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()


@pytest.mark.parametrize("i", range(N_CASES))
def test_fuzz_strict_bit_exact(built, oracle_mod, i, monkeypatch, exact_flavour):
    O = oracle_mod
    ch, kw, q, seed, rs, off = make_case(i)
    o = O.Oracle(ch)
    s = pk.Solver(ch, device=0, strict=True)
    try:
        with O.math_mode("portable"):
            goal = o.fk(q)
            np.testing.assert_array_equal(s.fk(q), goal, err_msg=f"case {i} fk")
            if i % 3 == 0:
                monkeypatch.setenv("PIK_PASSES", "1,2,3,5,8")
            a = s.solve_batch(pk.default_params(**kw), goal, seed, rng_seed=rs, problem_offset=off)
            b = o.solve_batch(O.default_params(**kw), goal, seed, rng_seed=rs, problem_offset=off,
                              num_threads=O.max_threads())
        for x, y, w in zip(a, b, ("solution", "status", "cost", "stats")):
            np.testing.assert_array_equal(x, y, err_msg=f"case {i} dof {ch.dof} {kw} {w}")
    finally:
        s.close()


def axis_aligned_chain(rng, dof, all_z, dh=False):
    """every variable a revolute joint about exactly +z (all_z) or about +x / +y / +z, no identity origin, a tip
    transform: the chain classes the exact flavour has specialised forms for (pik_math.hpp UZ / UA).  dh: every
    origin turns about its own x axis only (rpy = (alpha, 0, 0), the link twist of the Denavit-Hartenberg
    convention; some of them exactly 0 or +-pi/2, some translations with exact zeros, as real descriptions have) and
    the tip about its own z axis only -- with all_z that is class 1, whose chain product leaves the exact 0 / 1 entries
    of origins and tip out"""
    origins = np.zeros((dof, 6))
    origins[:, :3] = rng.uniform(-0.35, 0.35, size=(dof, 3))
    origins[:, 3:] = rng.uniform(-np.pi, np.pi, size=(dof, 3))
    if dh:
        origins[:, 4:] = 0.0
        special = rng.uniform(size=dof)
        origins[:, 3] = np.where(special < 0.2, 0.0, np.where(special < 0.4, np.pi / 2,
                                 np.where(special < 0.6, -np.pi / 2, origins[:, 3])))
        origins[:, :3] *= rng.uniform(size=(dof, 3)) < 0.6
    axes = np.tile([0.0, 0.0, 1.0], (dof, 1)) if all_z else np.eye(3)[rng.integers(0, 3, size=dof)]
    bounded = (rng.uniform(size=dof) < 0.85).astype(np.uint8)
    span, mid = rng.uniform(0.5, 3.1, size=dof), rng.uniform(-0.5, 0.5, size=dof)
    tip = np.concatenate([rng.uniform(-0.2, 0.2, size=3), rng.uniform(-np.pi, np.pi, size=3)])
    if dh:  # ... and the tip about its own z axis only (a tool flange; class 1 wants that too), sometimes not at all
        tip[3:5] = 0.0
        if abs(tip[5]) < 0.6:
            tip[5] = 0.0
    return robots._chain(f"aligned{dof}", origins, axes, tip, mid - span, mid + span, rng.uniform(0.5, 3.0, size=dof),
                         bounded=bounded, joint_type=np.zeros(dof, np.int32))


N_CLASS_CASES = int(os.environ.get("PIK_FUZZ_CLASS_CASES", "40"))  # 0-9 / 10-19: class 2; 20-29, 30-39: the same with x-twist origins (20-29: class 1)


@pytest.mark.parametrize("i", range(N_CLASS_CASES))
def test_fuzz_axis_aligned_chain_classes_bit_exact(built, oracle_mod, i, monkeypatch, exact_flavour):
    """the UZ form of the exact flavour (lengths 1..8 have it; 9 and 10, and the chains whose axes are a mix of +x /
    +y / +z, run the general form) against the oracle, tolerance zero: forward kinematics, whole solves under random
    parameters, every lanes-per-elite variant, with and without compaction passes"""
    O = oracle_mod
    rng = np.random.default_rng(0xA71 + i + SEED_SHIFT)
    dof = 1 + i % 10
    ch = axis_aligned_chain(rng, dof, all_z=(i // 10) % 2 == 0, dh=i >= 20)
    kw = random_params(rng)
    B = int(rng.integers(8, 120))
    lo = np.where(ch.bounded == 1, ch.qmin, -3.0)
    hi = np.where(ch.bounded == 1, ch.qmax, 3.0)
    q = rng.uniform(lo, hi, size=(B, dof))
    seed = rng.uniform(lo, hi, size=(B, dof))
    rs, off = int(rng.integers(0, 1 << 62)), int(rng.integers(0, 1 << 40))
    o = O.Oracle(ch)
    s = pk.Solver(ch, device=0, strict=True)
    try:
        with O.math_mode("portable"):
            goal = o.fk(q)
            np.testing.assert_array_equal(s.fk(q), goal, err_msg=f"case {i} fk")
            b = o.solve_batch(O.default_params(**kw), goal, seed, rng_seed=rs, problem_offset=off,
                              num_threads=O.max_threads())
            shapes = [("0", None), ("1", "1,2,3,5,8")] + [(str(l), p) for l in (2, 4, 8, 16) for p in (None, "1,2,4,7")]
            for lpe, passes in shapes[:4] if kw.get("mode") == 1 else shapes:
                monkeypatch.setenv("PIK_LPE", lpe)
                if passes:
                    monkeypatch.setenv("PIK_PASSES", passes)
                else:
                    monkeypatch.delenv("PIK_PASSES", raising=False)
                a = s.solve_batch(pk.default_params(**kw), goal, seed, rng_seed=rs, problem_offset=off)
                for x, y, w in zip(a, b, ("solution", "status", "cost", "stats")):
                    np.testing.assert_array_equal(x, y, err_msg=f"case {i} dof {dof} lanes {lpe} passes {passes} {kw} {w}")
    finally:
        s.close()


@pytest.mark.parametrize("i", range(N_CASES))
def test_fuzz_fast_shape_invariance(built, oracle_mod, i, monkeypatch):
    O = oracle_mod
    ch, kw, q, seed, rs, off = make_case(i)
    o = O.Oracle(ch)
    goal = o.fk(q)
    s = pk.Solver(ch, device=0, exact=False)
    try:
        p = pk.default_params(**kw)
        species = kw.get("memetic_num_threads", 1) > 1
        outs = []
        # lanes per elite x compaction marks; 8 / 16 lanes are the cooperative gradient descent (a
        # request the elite count does not allow falls back to the adaptive schedule, also a shape);
        # (None, None) = the library's defaults: adaptive variant choice on the device
        # (species: the lanes of all of a problem's species have to fit a wavefront, or the request falls back)
        shapes = [("1", "none"), ("4", "none"), ("1", "1,2,4,7"), ("4", "2,3"), ("2", "1,3"), ("8", "none"),
                  ("16", "2,3"), ("8", "1,2,4,7"), (None, None)]
        for lpe, marks in shapes:
            for var, val in (("PIK_LPE", lpe), ("PIK_PASSES", marks)):
                if val is None:
                    monkeypatch.delenv(var, raising=False)
                else:
                    monkeypatch.setenv(var, val)
            outs.append(s.solve_batch(p, goal, seed, rng_seed=rs, problem_offset=off))
        if not species and len(goal) >= 2:
            # and as a pool of two batches (one queue across batches)
            h = len(goal) // 2
            pooled = s.solve_batches(p, [(goal[:h], seed[:h], None, off), (goal[h:], seed[h:], None, off + h)],
                                     rng_seed=rs)
            outs.append(tuple(np.concatenate([pooled[0][k], pooled[1][k]]) for k in range(4)))
        names = [f"lanes {l} marks {m}" for l, m in shapes] + ["pool of two"]
        for other, name in zip(outs[1:], names[1:]):
            for x, y, w in zip(outs[0], other, ("solution", "status", "cost", "stats")):
                np.testing.assert_array_equal(x, y, err_msg=f"case {i} [{names[0]}] vs [{name}] {kw} {w}")
        sol, st, cost, _ = outs[0]
        ok = st == pk.SUCCESS
        op = O.default_params(**kw)
        for b in np.flatnonzero(ok)[:40]:
            c, is_sol = o.cost(op, goal[b], seed[b], sol[b])
            assert is_sol[0] == 1, f"case {i} problem {b}: SUCCESS but oracle rejects (cost {c[0]})"
            assert abs(c[0] - cost[b]) <= 1e-9 * max(1.0, abs(c[0]))
        np.testing.assert_array_equal(sol[st == pk.NO_IK_SOLUTION], seed[st == pk.NO_IK_SOLUTION])
    finally:
        s.close()


def common_case(i):
    """a chain and parameters that HAVE the common configuration: bounded revolute variables on non-degenerate
    axes, default cost terms, four elites, one species -- chain lengths 1..16 in turn"""
    rng = np.random.default_rng(0xC0FFEE + i + SEED_SHIFT)
    dof = 1 + i % 16
    origins = np.zeros((dof, 6))
    origins[:, :3] = rng.uniform(-0.35, 0.35, size=(dof, 3))
    origins[:, 3:] = rng.uniform(-np.pi, np.pi, size=(dof, 3))
    axes = (np.eye(3)[rng.integers(0, 3, size=dof)] * rng.choice([-1.0, 1.0], size=(dof, 1)) if i % 2
            else rng.normal(size=(dof, 3)))
    span, mid = rng.uniform(0.5, 3.1, size=dof), rng.uniform(-0.5, 0.5, size=dof)
    tip = np.concatenate([rng.uniform(-0.2, 0.2, size=3), rng.uniform(-np.pi, np.pi, size=3)])
    ch = robots._chain(f"common{dof}", origins, axes, tip, mid - span, mid + span, rng.uniform(0.5, 3.0, size=dof))
    P = int(5 + rng.integers(0, 60))
    kw = dict(memetic_population_size=P, memetic_max_generations=int(rng.integers(1, 30)),
              memetic_gd_max_iters=int(rng.choice([0, 1, 5, 25, 25])),
              gd_step_size=float(rng.choice([1e-4, 1e-3, 1e-5])),
              gd_min_cost_delta=float(rng.choice([1e-12, 1e-9, 1e-6])),
              memetic_wipeout_fitness_tol=float(rng.choice([1e-5, 1e-3, 1e-8])),
              position_threshold=float(rng.choice([1e-3, 1e-2, 1e-4])),
              orientation_threshold=float(rng.choice([1e-3, 1e-2, 1e-4])),
              position_scale=float(rng.choice([1.0, 1.0, 0.5, 2.0])),
              rotation_scale=float(rng.choice([0.5, 0.5, 1.0, 0.25])),
              stop_optimization_on_valid_solution=int(rng.uniform() < 0.8),
              return_approximate_solution=int(rng.uniform() < 0.3))
    if rng.uniform() < 0.25:
        kw["mode"] = 1
        kw["gd_max_iters"] = int(rng.choice([5, 40, 100]))
    if i % 3 == 2:  # joint goals: the second flavour of the common-configuration kernels
        side = np.random.default_rng(0xBEE + i + SEED_SHIFT)
        kw.update(center_joints_weight=float(side.choice([0.0, 0.01, 0.1])),
                  avoid_joint_limits_weight=float(side.choice([0.0, 0.02, 0.2])),
                  minimal_displacement_weight=float(side.choice([0.001, 0.05])),
                  cost_threshold=float(side.choice([1e-3, 0.05, 1.0])))
    B = int(rng.integers(1, 150))
    q = rng.uniform(ch.qmin, ch.qmax, size=(B, dof))
    seed = rng.uniform(ch.qmin, ch.qmax, size=(B, dof))
    near = rng.uniform(size=B) < 0.3
    seed[near] = np.clip(q[near] + rng.normal(0, 0.05, size=(int(near.sum()), dof)), ch.qmin, ch.qmax)
    return ch, kw, q, seed, int(rng.integers(0, 1 << 62)), int(rng.integers(0, 1 << 40))


@pytest.mark.parametrize("i", range(int(os.environ.get("PIK_FUZZ_COMMON_CASES", "48"))))
def test_fuzz_common_configuration_kernels(built, oracle_mod, i):
    """The kernels compiled for the common configuration, every chain length 1..16: the call is served by them
    (pikamd_kernel_name says which flavour), the answers are bit for bit the general kernels' in every execution
    shape, and every SUCCESS is a solution by the oracle's own test."""
    O = oracle_mod
    ch, kw, q, seed, rs, off = common_case(i)
    o = O.Oracle(ch)
    goal = o.fk(q)
    s = pk.Solver(ch, device=0, exact=False)
    try:
        p = pk.default_params(**kw)
        s.set_option("specialised", "1")
        want = "pik_common_goals::" if i % 3 == 2 else "pik_common::"
        if want not in s.kernel_name(p):  # (an ill-conditioned pair of axes: rare)
            pytest.skip(f"case {i}: {s.kernel_name(p)} serves this chain")
        s.set_option("specialised", "0")
        assert s.kernel_name(p).startswith("pik::")
        outs, names = [], []
        for spec in ("0", "1"):
            s.set_option("specialised", spec)
            for lanes, marks in ((1, "none"), (None, None), (2, "1,3"), (4, "2,3"), (8, "1,2,4,7"), (16, "none")):
                s.set_option("lanes_per_elite", lanes)
                s.set_option("passes", marks)
                outs.append(s.solve_batch(p, goal, seed, rng_seed=rs, problem_offset=off))
                names.append(f"specialised {spec} lanes {lanes} marks {marks}")
        for other, name in zip(outs[1:], names[1:]):
            for x, y, w in zip(outs[0], other, ("solution", "status", "cost", "stats")):
                np.testing.assert_array_equal(x, y, err_msg=f"case {i} dof {ch.dof} [{names[0]}] vs [{name}] {kw} {w}")
        sol, st, cost, _ = outs[-1]
        op = O.default_params(**kw)
        for b in np.flatnonzero(st == pk.SUCCESS)[:40]:
            c, is_sol = o.cost(op, goal[b], seed[b], sol[b])
            assert is_sol[0] == 1, f"case {i} problem {b}: SUCCESS but oracle rejects (cost {c[0]})"
            assert abs(c[0] - cost[b]) <= 1e-9 * max(1.0, abs(c[0]))
    finally:
        s.close()


@pytest.mark.parametrize("eps", [1e-4, 1e-7, 4.9e-12])
def test_ill_conditioned_axes_every_shape(built, oracle_mod, eps):
    """Consecutive joint axes that are nearly but not exactly parallel (a URDF that writes 1.57079632679 for
    pi/2): those pairs take a general constant step instead of the Denavit-Hartenberg one
    (ChainK::dh_general_mask) in the one-lane, two-lane and four-lane kernels; the cooperative descent does not
    have it, a request for 8 / 16 lanes is served by the adaptive schedule over the others.  Same answers in
    every execution shape, local and global mode, every SUCCESS a solution by the oracle's test."""
    import dataclasses
    O = oracle_mod
    ur5 = robots.ur5()
    for variant in range(3):
        origin = ur5.origin_xyz_rpy.copy()
        if variant == 0:    # elbow tilted about x against the (parallel) lift axis
            origin[2, 3] += eps
        elif variant == 1:  # two consecutive ill-conditioned pairs
            origin[2, 3] += eps
            origin[3, 5] -= 0.7 * eps
        else:               # every joint perturbed a little
            origin[:, 3:] += eps * np.array([[0.3, -0.2, 0.9]]) * np.arange(1, 7)[:, None]
        ch = dataclasses.replace(ur5, origin_xyz_rpy=origin)
        o = O.Oracle(ch)
        rng = np.random.default_rng(70 + variant)
        n = 200
        goal = o.fk(rng.uniform(ch.qmin, ch.qmax, size=(n, ch.dof)))
        goal[:20, 2] += 2.0  # out of reach: all generations
        seed = np.tile(robots.UR5_HOME, (n, 1))
        s = pk.Solver(ch, device=0, exact=False)
        try:
            for kw in (dict(memetic_population_size=32, memetic_max_generations=25), dict(mode=1)):
                p = pk.default_params(**kw)
                assert s.kernel_name(p).startswith("pik::"), s.kernel_name(p)  # (not the common configuration)
                outs, names = [], []
                for lanes, marks in ((1, "none"), (2, "1,3"), (4, "2,3"), (8, "none"), (16, "1,2,4,7"), (8, "2,5"), (None, None)):
                    s.set_option("lanes_per_elite", lanes)
                    s.set_option("passes", marks)
                    outs.append(s.solve_batch(p, goal, seed, rng_seed=5, problem_offset=9))
                    names.append(f"lanes {lanes} marks {marks}")
                for other, name in zip(outs[1:], names[1:]):
                    for x, y, w in zip(outs[0], other, ("solution", "status", "cost", "stats")):
                        np.testing.assert_array_equal(x, y, err_msg=f"eps {eps} variant {variant} {kw} [{names[0]}] vs [{name}] {w}")
                sol, st, cost, _ = outs[0]
                assert (st == pk.SUCCESS).sum() > 30, (eps, variant, kw, (st == pk.SUCCESS).sum())
                op = O.default_params(**kw)
                for b in np.flatnonzero(st == pk.SUCCESS)[:40]:
                    c, is_sol = o.cost(op, goal[b], seed[b], sol[b])
                    assert is_sol[0] == 1 and abs(c[0] - cost[b]) <= 1e-9 * max(1.0, abs(c[0]))
        finally:
            s.close()
