"""CPU: the oracle reproduces the committed golden vectors (guards the fixtures + the oracle
against silent drift), and the host-side pieces that need no GPU."""
import numpy as np
import pytest

from pick_ik_amd import robots
from tests.common import CONFIGS, golden


@pytest.mark.parametrize("name", ["panda", "ur5", "rr"])
def test_oracle_matches_golden_primitives(oracle_mod, name):
    O = oracle_mod
    G = golden()
    ch = robots.by_name(name)
    o = O.Oracle(ch)
    q = G[f"fk_{name}_q"]
    np.testing.assert_allclose(o.fk(q), G[f"fk_{name}_pose"], rtol=0, atol=1e-15)
    p = O.default_params(center_joints_weight=0.3, avoid_joint_limits_weight=0.2,
                         minimal_displacement_weight=0.1)
    goal, seed = G[f"cost_{name}_goal"], G[f"cost_{name}_seed"]
    cost = np.array([o.cost(p, goal[i], seed[i], q[i])[0][0] for i in range(len(q))])
    np.testing.assert_allclose(cost, G[f"cost_{name}_cost"], rtol=1e-15)
    c0 = G[f"step_{name}_c0"]
    local, best, lc, bc, grad, imp = o.gd_step(O.default_params(), goal, seed, q, q, c0, c0)
    np.testing.assert_allclose(local, G[f"step_{name}_local"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(grad, G[f"step_{name}_grad"], rtol=1e-13, atol=1e-18)


@pytest.mark.parametrize("cname", list(CONFIGS))
def test_oracle_matches_golden_memetic(oracle_mod, cname):
    O = oracle_mod
    G = golden()
    robot, home, kw = CONFIGS[cname]
    ch = robots.by_name(robot)
    o = O.Oracle(ch)
    goal = G[f"mem_{cname}_goal"]
    seed = np.tile(home, (len(goal), 1))
    sol, st, c, stats = o.solve_batch(O.default_params(**kw), goal, seed, rng_seed=0xC0FFEE,
                                      num_threads=2)
    np.testing.assert_array_equal(st, G[f"mem_{cname}_status"])
    np.testing.assert_allclose(sol, G[f"mem_{cname}_sol"], rtol=0, atol=1e-12)
    np.testing.assert_array_equal(stats["generations"], G[f"mem_{cname}_gens"])
    np.testing.assert_array_equal(stats["cost_evals"], G[f"mem_{cname}_evals"])


def test_oracle_batch_is_thread_and_offset_invariant(oracle_mod):
    """Random streams are keyed by the global problem index: sharding a batch (the multi-GPU
    decomposition) or changing the thread count must not change any answer."""
    O = oracle_mod
    ch = robots.panda()
    o = O.Oracle(ch)
    rng = np.random.default_rng(5)
    q = rng.uniform(ch.qmin, ch.qmax, size=(24, 7))
    goal = o.fk(q)
    seed = np.tile(robots.PANDA_HOME, (24, 1))
    p = O.default_params()
    a = o.solve_batch(p, goal, seed, rng_seed=9, num_threads=1)
    b = o.solve_batch(p, goal, seed, rng_seed=9, num_threads=4)
    np.testing.assert_array_equal(a[0], b[0])
    lo = o.solve_batch(p, goal[:10], seed[:10], rng_seed=9, problem_offset=0)
    hi = o.solve_batch(p, goal[10:], seed[10:], rng_seed=9, problem_offset=10)
    np.testing.assert_array_equal(a[0], np.concatenate([lo[0], hi[0]]))
    np.testing.assert_array_equal(a[1], np.concatenate([lo[1], hi[1]]))
    c = o.solve_batch(p, goal, seed, rng_seed=10)
    assert np.abs(a[0] - c[0]).max() > 1e-9  # a different seed gives different trajectories


def test_oracle_empty_and_invalid(oracle_mod):
    O = oracle_mod
    o = O.Oracle(robots.panda())
    sol, st, c, stats = o.solve_batch(O.default_params(), np.zeros((0, 7)), np.zeros((0, 7)))
    assert sol.shape == (0, 7) and st.shape == (0,)
    with pytest.raises(ValueError):
        o.solve_batch(O.default_params(memetic_population_size=4, memetic_elite_size=4),
                      np.zeros((1, 7)), np.zeros((1, 7)))
