"""The native multi-device front end of the C ABI (pikamd_solve_batch_sharded):
shard arithmetic on the CPU, and on the GPU the sharded call against ONE call over the whole batch,
bit for bit -- with one handle, and with two handles (two host threads, two shards) on the one device
a test box has."""
import numpy as np
import pytest

import pick_ik_amd as pk
from pick_ik_amd import solver as S
from pick_ik_amd.distributed import shard_bounds as py_shard_bounds
from tests.common import ARITHMETIC


def test_shard_bounds_of_the_library_equal_the_python_decomposition():
    for total in (0, 1, 7, 8, 37, 4096, 1048576 + 3):
        for world in (1, 2, 3, 8):
            edges = [S.shard_bounds(total, r, world) for r in range(world)]
            assert edges == [py_shard_bounds(total, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))


@pytest.mark.gpu
@pytest.mark.parametrize("exact", ARITHMETIC)
@pytest.mark.parametrize("B", [1, 5, 1000, 70001])
def test_sharded_call_equals_one_call(B, exact):
    ch = pk.robots.panda()
    handles = [pk.Solver(ch, device=0, exact=exact) for _ in range(3)]
    rng = np.random.default_rng(B)
    goal = handles[0].fk(rng.uniform(ch.qmin, ch.qmax, size=(B, 7)))
    seed = np.tile(pk.robots.PANDA_HOME, (B, 1))
    guess = rng.uniform(ch.qmin, ch.qmax, size=(B, 7))
    p = pk.default_params(memetic_population_size=32, memetic_max_generations=12)
    for ig in (None, guess):
        ref = handles[0].solve_batch(p, goal, seed, rng_seed=99, problem_offset=1234, initial_guess=ig)
        for n in (1, 2, 3):
            got = S.solve_batch_sharded(handles[:n], p, goal, seed, rng_seed=99, problem_offset=1234, initial_guess=ig)
            for a, b in zip(got, ref):
                np.testing.assert_array_equal(a, b)
    if B > 65536:  # more host jobs per device than the default two (option "shard_chunks")
        handles[0].set_option("shard_chunks", "3")
        got = S.solve_batch_sharded(handles[:1], p, goal, seed, rng_seed=99, problem_offset=1234, initial_guess=guess)
        for a, b in zip(got, ref):
            np.testing.assert_array_equal(a, b)
        handles[0].set_option("shard_chunks", None)
        with pytest.raises(pk.PickIkAmdError, match="shard_chunks"):
            handles[0].set_option("shard_chunks", "9")
    with pytest.raises(pk.PickIkAmdError, match="twice"):
        S.solve_batch_sharded([handles[0], handles[0]], p, goal, seed)
    for h in handles:
        h.close()


@pytest.mark.gpu
def test_self_test_finds_every_variant_in_agreement():
    """pikamd_self_test: every kernel variant against the one-lane kernel, on the device, for the chains and
    parameter sets of the reference configurations -- nothing may be switched off"""
    from tests.common import CONFIGS
    for cname, (robot, home, kw) in CONFIGS.items():
        for exact in (None, False):  # the default (exact) kernels, the fast ones
            s = pk.Solver(pk.robots.by_name(robot), device=0, exact=exact)
            assert s.self_test(pk.default_params(**kw), 96) == 0, (cname, exact)
            s.close()
    for name, kw in (("torso_dual_arm", dict(memetic_population_size=24)), ("floating_panda", dict(memetic_population_size=24)),
                     ("panda", dict(memetic_num_threads=2)), ("panda", dict(mode=1))):
        s = pk.Solver(pk.robots.by_name(name), device=0)
        assert s.self_test(pk.default_params(**kw), 48) == 0, name
        s.close()
