"""Mimic joints on the GPU: chains with a joint that follows a variable are solved by the exact kernels (both
libraries), BIT-IDENTICAL to the oracle -- forward kinematics, cost / solution test, step(), ik_gradient,
ik_memetic; the forward kinematics also against the chain in which the joint is an ordinary variable; a robot
description with a <mimic> element through the native reader."""
import numpy as np
import pytest

import pick_ik_amd as pk
from pick_ik_amd import robots
from tests.test_gpu_fuzz import random_chain, random_params
from tests.test_mimic_cpu import CASES, expand, with_mimic

pytestmark = pytest.mark.gpu
# other generated cases than the suite's: PIK_FUZZ_SEED_SHIFT=100000 pytest ... (soaks, profiles/r04_fuzz_soaks.txt)
SEED_SHIFT = int(__import__("os").environ.get("PIK_FUZZ_SEED_SHIFT", "0"))

MODE = {True: "portable", False: "fma"}  # library (strict?) -> oracle math mode


@pytest.fixture(scope="module")
def O(oracle_mod):
    import __graft_entry__ as g
    g.build()
    return oracle_mod


def eq(a, b, what=""):
    np.testing.assert_array_equal(a, b, err_msg=what)


@pytest.mark.parametrize("strict", [True, False])
@pytest.mark.parametrize("name,k,master,mult,off", CASES)
def test_mimic_chain_bit_exact(O, name, k, master, mult, off, strict):
    full = robots.by_name(name)
    rng = np.random.default_rng(5 + k)
    ch, keep = with_mimic(rng, full, k, master, mult, off)
    n = 40
    q = rng.uniform(ch.qmin, ch.qmax, size=(n, ch.dof))
    seed = np.clip(q + rng.normal(0, 0.3, size=q.shape), ch.qmin, ch.qmax)
    o = O.Oracle(ch)
    s = pk.Solver(ch, device=0, strict=strict)
    f = pk.Solver(full, device=0, strict=strict, exact=not strict)
    try:
        assert "pik_exact" in s.kernel_name(pk.default_params()) or strict
        with O.math_mode(MODE[strict]):
            goal = o.fk(q)
            eq(s.fk(q), goal, "fk vs oracle")
            eq(s.fk(q), f.fk(expand(q, keep, k, master, mult, off, full.dof)), "fk vs the joint as a variable")
            kw = dict(center_joints_weight=0.3, minimal_displacement_weight=0.1)
            cand = q + rng.normal(0, 1e-3, size=q.shape)
            gc, gs = s.cost(pk.default_params(**kw), goal, seed, cand)
            res = [o.cost(O.default_params(**kw), goal[i], seed[i], cand[i]) for i in range(n)]
            eq(gc, np.array([r[0][0] for r in res]), "cost")
            eq(gs, np.array([r[1][0] for r in res]), "solution_fn")
            c0 = np.array([o.cost(O.default_params(), goal[i], seed[i], cand[i])[0][0] for i in range(n)])
            for x, y, w in zip(s.gd_step(pk.default_params(), goal, seed, cand, cand, c0, c0),
                               o.gd_step(O.default_params(), goal, seed, cand, cand, c0, c0),
                               ("local", "best", "local_cost", "best_cost", "gradient", "improved")):
                eq(x, y, "step " + w)
            for kw in (dict(memetic_population_size=24, memetic_max_generations=12),
                       dict(mode=1, gd_max_iters=40),
                       dict(memetic_population_size=16, memetic_elite_size=2, memetic_max_generations=6,
                            minimal_displacement_weight=0.01, return_approximate_solution=1)):
                b = o.solve_batch(O.default_params(**kw), goal, seed, rng_seed=5, problem_offset=7, num_threads=O.max_threads())
                for lanes, marks in ((None, None), (1, "none"), (4, "1,3")):
                    s.set_option("lanes_per_elite", lanes)
                    s.set_option("passes", marks)
                    a = s.solve_batch(pk.default_params(**kw), goal, seed, rng_seed=5, problem_offset=7)
                    for x, y, w in zip(a, b, ("solution", "status", "cost", "stats")):
                        eq(x, y, f"{name} strict={strict} {kw} lanes {lanes} marks {marks}: {w}")
    finally:
        s.close()
        f.close()


@pytest.mark.parametrize("i", range(8))
def test_fuzz_mimic_bit_exact(O, i):
    rng = np.random.default_rng(0x313 + i + SEED_SHIFT)
    full = random_chain(rng, 3 + i)
    while any(t not in (robots.REVOLUTE, robots.PRISMATIC) for t in full.joint_type):
        full = random_chain(rng, 3 + i)
    k = int(rng.integers(0, full.dof))
    master = int(rng.choice([j for j in range(full.dof) if j != k]))
    ch, keep = with_mimic(rng, full, k, master, float(rng.uniform(-1.5, 1.5)), float(rng.uniform(-0.3, 0.3)))
    kw = random_params(rng)
    kw.pop("memetic_num_threads", None)
    B = int(rng.integers(1, 50))
    lo = np.where(ch.bounded == 1, ch.qmin, -3.0)
    hi = np.where(ch.bounded == 1, ch.qmax, 3.0)
    q = rng.uniform(lo, hi, size=(B, ch.dof))
    seed = rng.uniform(lo, hi, size=(B, ch.dof))
    o = O.Oracle(ch)
    for strict in (True, False):
        s = pk.Solver(ch, device=0, strict=strict)
        try:
            with O.math_mode(MODE[strict]):
                goal = o.fk(q)
                eq(s.fk(q), goal, f"case {i} fk")
                a = s.solve_batch(pk.default_params(**kw), goal, seed, rng_seed=i, problem_offset=5)
                b = o.solve_batch(O.default_params(**kw), goal, seed, rng_seed=i, problem_offset=5, num_threads=O.max_threads())
            for x, y, w in zip(a, b, ("solution", "status", "cost", "stats")):
                eq(x, y, f"case {i} strict={strict} dof {ch.dof} {kw} {w}")
        finally:
            s.close()


URDF = """<robot name="coupled"><link name="base"/><link name="l1"/><link name="l2"/><link name="l3"/><link name="l4"/><link name="tool"/>
  <joint name="j1" type="revolute"><parent link="base"/><child link="l1"/><origin xyz="0 0 0.3"/><axis xyz="0 0 1"/>
    <limit lower="-2.5" upper="2.5" velocity="1" effort="1"/></joint>
  <joint name="j2" type="revolute"><parent link="l1"/><child link="l2"/><origin xyz="0.25 0 0" rpy="0 0.2 0"/><axis xyz="0 1 0"/>
    <limit lower="-2" upper="2" velocity="1" effort="1"/></joint>
  <joint name="j2b" type="revolute"><parent link="l2"/><child link="l3"/><origin xyz="0.2 0 0"/><axis xyz="0 1 0"/>
    <mimic joint="j2" multiplier="-1" offset="0.1"/><limit lower="-2" upper="2" velocity="1" effort="1"/></joint>
  <joint name="j3" type="revolute"><parent link="l3"/><child link="l4"/><origin xyz="0.2 0 0.05"/><axis xyz="1 0 0"/>
    <limit lower="-2" upper="2" velocity="1" effort="1"/></joint>
  <joint name="tool_fixed" type="fixed"><parent link="l4"/><child link="tool"/><origin xyz="0.1 0 0"/></joint></robot>"""


def test_robot_description_with_a_mimic_joint(O):
    s = pk.Solver.from_urdf(URDF, "base", "tool")
    try:
        assert s.dof == 3 and s.variable_names == ["j1", "j2", "j3"] and len(s.chain.mimic) == 1
        o = O.Oracle(s.chain)
        rng = np.random.default_rng(1)
        q = rng.uniform(s.chain.qmin, s.chain.qmax, size=(64, 3))
        with O.math_mode("fma"):
            goal = o.fk(q)
            eq(s.fk(q), goal)
            p = dict(memetic_population_size=32, memetic_max_generations=30, rotation_scale=0.0)
            a = s.solve_batch(pk.default_params(**p), goal, np.zeros((64, 3)), rng_seed=2)
            b = o.solve_batch(O.default_params(**p), goal, np.zeros((64, 3)), rng_seed=2, num_threads=O.max_threads())
        for x, y in zip(a, b):
            eq(x, y)
        assert (a[1] == pk.SUCCESS).mean() > 0.8
    finally:
        s.close()
