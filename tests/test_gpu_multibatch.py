"""Several batches per call (pikamd_solve_batches*): the problems of all batches are solved as one
pool by the persistent wavefronts, and every batch must get, bit for bit, the answers a call of its
own gives -- device entry point (HBM-resident, completion counters) and host entry points
(synchronous, and asynchronous jobs that overlap their PCIe transfers)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import pick_ik_amd as pk
from pick_ik_amd import robots
from tests.common import random_targets

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def O(oracle_mod):
    return oracle_mod


@pytest.fixture(scope="module", params=[None, False], ids=["default_exact", "fast"])
def panda(request):
    import __graft_entry__ as g
    g.build()
    s = pk.Solver(robots.panda(), device=0, exact=request.param)
    yield s
    s.close()


def make_batches(s, O, rng, sizes, with_guess=True):
    o = O.Oracle(s.chain)
    out, off = [], 40
    for B in sizes:
        _, goal = random_targets(o.fk, s.chain, rng, B)
        seed = rng.uniform(s.chain.qmin, s.chain.qmax, size=(B, s.dof))
        guess = rng.uniform(s.chain.qmin, s.chain.qmax, size=(B, s.dof)) if with_guess else None
        out.append((goal, seed, guess, off))
        off += 3 * B + 1
    return out


def assert_same(a, b, what):
    for x, y, w in zip(a, b, ("solution", "status", "cost", "stats")):
        np.testing.assert_array_equal(x, y, err_msg=f"{what}: {w}")


@pytest.mark.parametrize("kw", [
    dict(memetic_population_size=48, memetic_max_generations=40),
    dict(memetic_population_size=20, memetic_elite_size=2, memetic_max_generations=30,
         minimal_displacement_weight=0.05, center_joints_weight=0.02, cost_threshold=0.05),
    dict(mode=1, gd_max_iters=60),
], ids=["memetic", "memetic_goals", "local"])
def test_host_multi_batch_equals_single_calls(panda, O, kw):
    s = panda
    rng = np.random.default_rng(31)
    batches = make_batches(s, O, rng, [257, 1, 1000, 0, 63, 64, 65])
    p = pk.default_params(**kw)
    pooled = s.solve_batches(p, batches, rng_seed=9)
    for k, (goal, seed, guess, off) in enumerate(batches):
        single = s.solve_batch(p, goal, seed, rng_seed=9, problem_offset=off, initial_guess=guess)
        assert_same(pooled[k], single, f"batch {k}")
    assert sum((r[1] == pk.SUCCESS).sum() for r in pooled) > 100


def test_pool_with_compaction_passes_and_lanes(panda, O, monkeypatch):
    """the pool under every launch schedule: compaction marks x lanes per elite (incl. the
    one-problem-per-wavefront variants), against the plain single-batch call"""
    s = panda
    rng = np.random.default_rng(32)
    batches = make_batches(s, O, rng, [300, 7, 129], with_guess=False)
    p = pk.default_params(memetic_population_size=40, memetic_max_generations=36)
    monkeypatch.setenv("PIK_PASSES", "none")
    monkeypatch.setenv("PIK_LPE", "1")
    ref = [s.solve_batch(p, g, sd, rng_seed=4, problem_offset=off) for g, sd, _, off in batches]
    for marks in ("none", "1,2,3,4,6,9,14,21,30", "2,4,8,16,32"):
        for lpe in ("1", "2", "4", "8", "16"):
            monkeypatch.setenv("PIK_PASSES", marks)
            monkeypatch.setenv("PIK_LPE", lpe)
            got = s.solve_batches(p, batches, rng_seed=4)
            for k in range(len(batches)):
                assert_same(got[k], ref[k], f"marks={marks} lpe={lpe} batch {k}")


def test_async_jobs_overlap_and_match(panda, O):
    """pikamd_solve_batches_async / pikamd_wait: several jobs in flight, results after wait() equal
    the synchronous calls; a job id cannot be reused before its wait"""
    s = panda
    rng = np.random.default_rng(33)
    p = pk.default_params(memetic_population_size=32, memetic_max_generations=20)
    jobs = [make_batches(s, O, rng, sizes) for sizes in ([500, 20], [3], [800, 800, 1], [64])]
    outs = [s.solve_batches(p, b, rng_seed=100 + j, job=j) for j, b in enumerate(jobs)]
    with pytest.raises(pk.PickIkAmdError, match="still in flight"):
        s.solve_batches(p, jobs[0], rng_seed=1, job=0)
    for j in reversed(range(len(jobs))):
        s.wait(j)
    s.wait(2)  # waiting twice is harmless
    for j, b in enumerate(jobs):
        ref = s.solve_batches(p, b, rng_seed=100 + j)
        for k in range(len(b)):
            assert_same(outs[j][k], ref[k], f"job {j} batch {k}")


def test_argument_checks(panda):
    s = panda
    p = pk.default_params()
    g = np.zeros((2, 7)); g[:, 3] = 1.0
    with pytest.raises(pk.PickIkAmdError):
        s.solve_batches(p, [(g, np.zeros((2, 7)), None, 0)] * (pk.solver.MAX_BATCHES + 1))
    assert s.solve_batches(p, []) == []
    with pytest.raises(pk.PickIkAmdError, match="job out of range"):
        s.solve_batches(p, [(g, np.zeros((2, 7)), None, 0)], job=pk.solver.MAX_HOST_JOBS - 1)


def test_device_multi_batch_pool():
    """HBM-resident batches + completion counters (own interpreter: torch allocates the buffers)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "multibatch_check.py")], cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "multi-batch check OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_solver_created_from_urdf_text(O):
    """pikamd_create_from_urdf: the library reads the robot description itself (one tip and several
    tips); FK equals the table-built solver's, solves return oracle-valid solutions."""
    from pick_ik_amd.urdf import chain_to_urdf
    from tests.test_urdf_cpu import DUAL
    ch = robots.panda()
    s = pk.Solver.from_urdf(chain_to_urdf(ch), "base", "tip")
    t = pk.Solver(ch)
    rng = np.random.default_rng(3)
    q = rng.uniform(ch.qmin, ch.qmax, size=(500, 7))
    np.testing.assert_allclose(s.fk(q), t.fk(q), rtol=0, atol=1e-12)
    goal = s.fk(q[:200])
    seed = np.tile(robots.PANDA_HOME, (200, 1))
    p, po = pk.default_params(memetic_population_size=32), O.default_params(memetic_population_size=32)
    sol, st, _, _ = s.solve_batch(p, goal, seed, rng_seed=1)
    assert (st == pk.SUCCESS).mean() > 0.9
    o = O.Oracle(ch)
    for b in np.nonzero(st == pk.SUCCESS)[0]:
        assert o.cost(po, goal[b], seed[b], sol[b])[1][0] == 1
    s.close()
    t.close()
    m = pk.Solver.from_urdf(DUAL, "base", ["lhand", "rhand"])
    assert m.n_tips == 2 and m.dof == 5 and m.variable_names[0] == "torso_yaw"
    om = O.Oracle(m.chain)
    qm = rng.uniform(-1, 1, size=(64, 5))
    qm[:, 4] = np.abs(qm[:, 4]) * 0.2
    np.testing.assert_allclose(m.fk(qm), om.fk(qm), rtol=0, atol=1e-12)
    sol, st, _, _ = m.solve_batch(pk.default_params(memetic_population_size=32), om.fk(qm).reshape(64, 14),
                                  np.zeros((64, 5)), rng_seed=2)
    assert (st == pk.SUCCESS).mean() > 0.5
    m.close()


def test_regime_does_not_change_results(panda, O, monkeypatch):
    """With three or more other calls in flight the library picks one lane per elite for every pass
    (efficiency) instead of the widest variant that fits (latency): a scheduling choice, results are
    bit-identical -- forced both ways here, and reached for real by enqueueing five async jobs."""
    s = panda
    rng = np.random.default_rng(35)
    batches = make_batches(s, O, rng, [600, 50], with_guess=False)
    p = pk.default_params(memetic_population_size=40, memetic_max_generations=30)
    outs = {}
    for regime in ("latency", "throughput"):
        monkeypatch.setenv("PIK_REGIME", regime)
        outs[regime] = s.solve_batches(p, batches, rng_seed=6)
    monkeypatch.delenv("PIK_REGIME")
    for k in range(len(batches)):
        assert_same(outs["latency"][k], outs["throughput"][k], f"batch {k}")
    jobs = [s.solve_batches(p, batches, rng_seed=6, job=j) for j in range(5)]
    for j in range(5):
        s.wait(j)
        for k in range(len(batches)):
            assert_same(jobs[j][k], outs["latency"][k], f"job {j} batch {k}")
