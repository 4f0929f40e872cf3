"""pikamd_self_test and the option self_test = auto on the GPU: every kernel variant the launcher would serve for a
chain and a parameter set -- lanes per elite with and without compaction passes, species, several tips, the
cooperative local-mode kernels, the two-per-SIMD build, the general against the common-configuration kernels, the
exact kernels -- against the one-lane kernel, bit for bit; the returned mask names the variants that disagreed
(none may)."""
import numpy as np
import pytest

import pick_ik_amd as pk
from pick_ik_amd import robots

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()


FAST = dict(exact=False)  # (the opt-in Denavit-Hartenberg flavour; dict() = the library's default, the exact kernels)
CASES = [
    ("panda", FAST, dict(memetic_population_size=32, memetic_max_generations=12)),                      # common flavour
    ("panda", FAST, dict(memetic_population_size=32, memetic_elite_size=5, memetic_max_generations=12)),  # general
    ("panda", FAST, dict(memetic_population_size=24, memetic_max_generations=8, memetic_num_threads=2)),  # species
    ("panda", FAST, dict(mode=1, gd_max_iters=40)),                                                      # local mode
    ("ur5", FAST, dict(memetic_population_size=32, memetic_max_generations=10, center_joints_weight=0.05)),
    ("torso_dual_arm", FAST, dict(memetic_population_size=24, memetic_max_generations=8)),               # two tips
    ("torso_dual_arm", FAST, dict(mode=1, gd_max_iters=30)),
    ("panda", dict(), dict(memetic_population_size=24, memetic_max_generations=8)),                       # exact kernels (default)
    ("panda", dict(), dict(memetic_population_size=24, memetic_max_generations=8, memetic_num_threads=2)),
    ("panda", dict(), dict(mode=1, gd_max_iters=40)),
    ("ur5", dict(), dict(memetic_population_size=32, memetic_max_generations=10, center_joints_weight=0.05)),
    ("torso_dual_arm", dict(), dict(memetic_population_size=24, memetic_max_generations=8)),
    ("panda", dict(strict=True), dict(memetic_population_size=24, memetic_max_generations=8)),
    ("floating_panda", dict(), dict(memetic_population_size=24, memetic_max_generations=6)),
]


@pytest.mark.parametrize("name,how,kw", CASES)
def test_self_test_finds_no_disagreeing_variant(built, name, how, kw):
    s = pk.Solver(robots.by_name(name), device=0, **how)
    try:
        assert s.self_test(pk.default_params(**kw), 48) == 0
    finally:
        s.close()


def test_self_test_runs_by_itself_once_per_parameter_set(built, monkeypatch):
    ch = robots.panda()
    rng = np.random.default_rng(3)
    n = 40
    q = rng.uniform(ch.qmin, ch.qmax, size=(n, ch.dof))
    seed = np.tile(robots.PANDA_HOME, (n, 1))
    params = pk.default_params(memetic_population_size=32, memetic_elite_size=5, memetic_max_generations=15)
    monkeypatch.setenv("PIK_SELF_TEST", "off")
    off = pk.Solver(ch, device=0, exact=False)  # (no automatic self test on this handle)
    goal = off.fk(q)
    want = off.solve_batch(params, goal, seed, rng_seed=9)
    off.close()
    monkeypatch.setenv("PIK_SELF_TEST", "auto")
    for how in (dict(exact=False), dict()):
        s = pk.Solver(ch, device=0, **how)
        try:
            assert s.self_test_cost() == (0, 0.0)
            a = s.solve_batch(params, goal, seed, rng_seed=9)   # self test first (general / exact kernels), then the call
            runs, ms = s.self_test_cost()
            assert runs == 1 and 0.0 < ms < 5000.0, (how, runs, ms)  # (the short form: ~10 ms of kernels + a dozen host round
            # trips; the bound only catches the long form coming back -- a cold first launch on a loaded box is slow)
            print(f"automatic self test ({'general kernels' if how else 'exact kernels (default)'}): {ms:.1f} ms")
            b = s.solve_batch(params, goal, seed, rng_seed=9)   # not again
            assert s.self_test_cost()[0] == 1
            for x, y in zip(a, b):
                np.testing.assert_array_equal(x, y)
            # other thresholds / population / budgets select the same kernels: no new run
            p_same = pk.default_params(memetic_population_size=48, memetic_elite_size=5, memetic_max_generations=9,
                                       position_threshold=2e-3, cost_threshold=5e-3, memetic_wipeout_fitness_tol=1e-4)
            s.solve_batch(p_same, goal, seed, rng_seed=9)
            assert s.self_test_cost()[0] == 1
            if how:  # (the fast flavour: the handle without a self test returned the same)
                for x, y in zip(a, want):
                    np.testing.assert_array_equal(x, y)
            # a job of the caller in flight while another parameter set is self-tested: the results of both stand
            p2 = pk.default_params(memetic_population_size=20, memetic_elite_size=3, memetic_max_generations=6)
            s.solve_batches(params, [(goal, seed, None, 0)], rng_seed=9, job=0)
            c = s.solve_batch(p2, goal, seed, rng_seed=4)
            s.wait(0)
            d = s.solve_batch(p2, goal, seed, rng_seed=4)
            for x, y in zip(c, d):
                np.testing.assert_array_equal(x, y)
            assert s.self_test_cost()[0] == 2  # (three elites deal a wavefront out differently: another kernel set)
        finally:
            s.close()
