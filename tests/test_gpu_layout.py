"""Option joint_layout = soa: the joint-vector arrays of the solve entry points (seed, initial guess, solution)
as [dof][B] per batch -- the structure-of-arrays layout BASELINE.json's north_star names -- against the default
[B][dof] of a MoveIt caller: the same answers, bit for bit (the library transposes on the device around the
kernels; the kernels gather a problem's vector from one cache line either way)."""
import numpy as np
import pytest

import pick_ik_amd as pk
from pick_ik_amd import robots
from pick_ik_amd import solver as S
from tests.common import ARITHMETIC

pytestmark = pytest.mark.gpu


def soa(a):
    """[B][dof] -> the same numbers laid out [dof][B] (handed over as a flat buffer)"""
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).T)


def from_soa(flat, B, dof):
    return np.asarray(flat).reshape(dof, B).T


@pytest.mark.parametrize("exact", ARITHMETIC)
@pytest.mark.parametrize("name", ["panda", "torso_dual_arm"])
def test_soa_layout_equals_aos(name, exact):
    import __graft_entry__ as g
    g.build()
    ch = robots.by_name(name)
    s = pk.Solver(ch, device=0, exact=exact)
    rng = np.random.default_rng(5)
    B = 700
    goal = s.fk(rng.uniform(ch.qmin, ch.qmax, size=(B, ch.dof)))
    goal.reshape(B, -1)[:30, 2] += 3.0  # out of reach: fail -> the SEED comes back (through the transposition)
    seed = rng.uniform(ch.qmin, ch.qmax, size=(B, ch.dof))
    guess = rng.uniform(ch.qmin, ch.qmax, size=(B, ch.dof))
    try:
        for kw in (dict(memetic_population_size=24, memetic_max_generations=15), dict(mode=1)):
            p = pk.default_params(**kw)
            for ig in (None, guess):
                ref = s.solve_batch(p, goal, seed, rng_seed=4, problem_offset=2, initial_guess=ig)
                s.set_option("joint_layout", "soa")
                try:
                    got = s.solve_batch(p, goal, soa(seed), rng_seed=4, problem_offset=2,
                                        initial_guess=None if ig is None else soa(ig))
                finally:
                    s.set_option("joint_layout", None)
                np.testing.assert_array_equal(from_soa(got[0], B, ch.dof), ref[0], err_msg=f"{name} {kw} solution")
                for a, b, w in zip(got[1:], ref[1:], ("status", "cost", "stats")):
                    np.testing.assert_array_equal(a, b, err_msg=f"{name} {kw} {w}")
                assert (ref[1] == pk.SUCCESS).sum() > 40 and (ref[1] == pk.NO_IK_SOLUTION).sum() >= 20
        # a pool of ragged batches: every batch its own [dof][B_k] arrays
        p = pk.default_params(memetic_population_size=24, memetic_max_generations=15)
        cuts = [0, 1, 130, 131, 700]
        aos = [(goal[a:b], seed[a:b], guess[a:b], a) for a, b in zip(cuts[:-1], cuts[1:])]
        ref = s.solve_batches(p, aos, rng_seed=9)
        s.set_option("joint_layout", "soa")
        try:
            got = s.solve_batches(p, [(gl, soa(sd), soa(ig), off) for gl, sd, ig, off in aos], rng_seed=9)
            with pytest.raises(pk.PickIkAmdError, match="sharded"):
                S.solve_batch_sharded([s], p, goal, soa(seed))
        finally:
            s.set_option("joint_layout", None)
        for (gl, _, _, _), r, o in zip(aos, ref, got):
            np.testing.assert_array_equal(from_soa(o[0], len(gl), ch.dof), r[0])
            for a, b in zip(o[1:], r[1:]):
                np.testing.assert_array_equal(a, b)
        with pytest.raises(pk.PickIkAmdError, match="joint_layout"):
            s.set_option("joint_layout", "blocked")
        assert s.self_test(p, 32) == 0  # (the self test keeps to its own arrays whatever the option says)
    finally:
        s.close()
