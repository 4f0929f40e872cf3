"""The benchmarked (product) build's own arithmetic, checked directly.

  * sqrt_pair -- the one primitive of the product build that is not an IEEE operation (v_rsq_f64 + Goldschmidt) --
    returns the CORRECTLY ROUNDED root and the IEEE quotient 0.5 / root on 10^7 inputs per range
    (tests/native/sqrt_pair_check.hip);
  * therefore every operation of the product's arithmetic is reproducible on a host, and a SEQUENTIAL host loop
    over the same source (pik_host_solve.hpp compiled as the fast flavour: Denavit-Hartenberg forward kinematics,
    frame-based gradient probes, line search by angle addition, the reference's sequential mating pool and
    std::sort) must return the kernels' answers BIT FOR BIT: 512 problems of BASELINE config 2, and generated
    chains / parameters that the general kernels serve.  This ties the product kernels' whole machinery --
    persistent wavefronts, streamed children with re-speculation, ballot/shuffle top-E, compaction passes,
    cooperative descent, the kernels specialised for the common configuration -- to a plain loop directly,
    instead of through the exact build."""
import json
import os
import subprocess

import numpy as np
import pytest

import pick_ik_amd as pk
from pick_ik_amd import robots
from tests.test_gpu_fuzz import random_chain, random_params

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NATIVE = os.path.join(ROOT, "tests", "native")
HIPCC = "/opt/rocm/bin/hipcc"


def _build(src, exe, flags):
    deps = [os.path.join(NATIVE, src)] + [os.path.join(ROOT, "pick_ik_amd", "csrc", f)
                                          for f in ("pik_math.hpp", "pik_host.hpp", "pik_solver.hpp", "pik_host_solve.hpp")]
    out = os.path.join(NATIVE, exe)
    if not os.path.exists(out) or os.path.getmtime(out) < max(map(os.path.getmtime, deps)):
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-ffp-contract=on", *flags,
                        os.path.join(NATIVE, src), "-o", out], check=True)
    return out


def test_sqrt_pair_is_correctly_rounded():
    exe = _build("sqrt_pair_check.hip", "sqrt_pair_check", [])
    for lo, hi in (("1e-12", "1e2"), ("1", "4")):  # sums of squares of lengths / of quaternion components; [1, 4]: matrix -> quaternion
        r = subprocess.run([exe, "10000000", lo, hi], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print(d)
        assert d["root_not_correctly_rounded"] == 0 and d["half_inverse_differs_from_0.5_over_sqrt"] == 0, d


def _host(ch, kw, goal, seed, rng_seed, offset):
    exe = _build("host_product_check.hip", "host_product_check", ["--cuda-host-only", "-mfma"])
    p = pk.default_params(**kw)
    lines = [f"{ch.dof} {len(goal)} {rng_seed} {offset}"]
    multi = hasattr(ch, "tips")
    d = ch.dof
    flat = lambda a: " ".join(repr(float(x)) for x in np.ravel(a))  # noqa: E731
    if multi:  # (the serial-chain arrays are placeholders; the tip paths follow)
        lines += [flat(np.zeros((d, 6))), flat(np.tile([0.0, 0.0, 1.0], (d, 1))), flat(np.zeros(6))]
    else:
        lines += [flat(ch.origin_xyz_rpy), flat(ch.axis), flat(ch.tip_xyz_rpy)]
    lines += [flat(ch.qmin), flat(ch.qmax), flat(ch.vmax)]
    lines.append(" ".join(f"{int(t)} {int(b)}" for t, b in zip(ch.joint_type if not multi else np.zeros(d), ch.bounded)))
    if multi:
        lines.append(str(ch.n_tips))
        for t in ch.tips:
            lines += [str(len(t.variable)), " ".join(str(int(v)) for v in t.variable), flat(t.origin_xyz_rpy), flat(t.axis),
                      " ".join(str(int(v)) for v in t.joint_type), flat(t.tip_xyz_rpy)]
    else:
        lines.append("0")
    lines.append(f"{int(p.memetic_num_threads)} {int(p.memetic_stop_on_first_solution)}")
    lines.append(" ".join(repr(x) for x in (
        int(p.mode), float(p.gd_step_size), int(p.gd_max_iters), float(p.gd_min_cost_delta), float(p.position_threshold),
        float(p.orientation_threshold), float(p.cost_threshold), float(p.position_scale), float(p.rotation_scale),
        float(p.center_joints_weight), float(p.avoid_joint_limits_weight), float(p.minimal_displacement_weight),
        int(p.stop_optimization_on_valid_solution), int(p.memetic_population_size), int(p.memetic_elite_size),
        float(p.memetic_wipeout_fitness_tol), int(p.memetic_max_generations), int(p.memetic_gd_max_iters),
        int(p.return_approximate_solution))))
    for g, s in zip(goal, seed):
        lines.append(" ".join(repr(float(x)) for x in np.concatenate([np.ravel(g), s])))
    r = subprocess.run([exe], input="\n".join(lines), capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [ln.split() for ln in r.stdout.strip().splitlines()]
    st = np.array([int(x[0]) for x in rows], dtype=np.int32)
    cost = np.array([float(x[1]) for x in rows])
    stats = np.array([[int(v) for v in x[2:6]] for x in rows], dtype=np.int64)
    sol = np.array([[float(v) for v in x[6:]] for x in rows]).reshape(len(rows), ch.dof)
    return sol, st, cost, stats


def _compare(ch, kw, B, seed_pose, rng_seed, offset, what, reachable=True):
    rng = np.random.default_rng(rng_seed + 17)
    lo = np.where(ch.bounded == 1, ch.qmin, -3.0)
    hi = np.where(ch.bounded == 1, ch.qmax, 3.0)
    q = rng.uniform(lo, hi, size=(B, ch.dof))
    seed = np.tile(seed_pose, (B, 1)) if seed_pose is not None else rng.uniform(lo, hi, size=(B, ch.dof))
    s = pk.Solver(ch, device=0, exact=False)
    try:
        goal = s.fk(q)
        if not reachable:
            goal[:, :3] *= 3.0
        name = s.kernel_name(pk.default_params(**kw))
        sol, st, cost, stats = s.solve_batch(pk.default_params(**kw), goal, seed, rng_seed=rng_seed, problem_offset=offset)
    finally:
        s.close()
    hsol, hst, hcost, hstats = _host(ch, kw, goal, seed, rng_seed, offset)
    np.testing.assert_array_equal(st, hst, err_msg=f"{what} ({name}): status")
    np.testing.assert_array_equal(sol, hsol, err_msg=f"{what} ({name}): solution")
    np.testing.assert_array_equal(cost, hcost, err_msg=f"{what} ({name}): cost")
    np.testing.assert_array_equal(stats["cost_evals"], hstats[:, 0], err_msg=f"{what}: evaluations")
    np.testing.assert_array_equal(stats["generations"], hstats[:, 1], err_msg=f"{what}: generations")
    np.testing.assert_array_equal(stats["wipeouts"], hstats[:, 2], err_msg=f"{what}: wipeouts")
    np.testing.assert_array_equal(stats["pool_erasures"], hstats[:, 3], err_msg=f"{what}: erasures")
    return name, (st == pk.SUCCESS).mean()


def test_config2_whole_solves_equal_the_sequential_host_execution():
    """BASELINE config 2 (Panda, population 128, the benchmark's seed pose): 512 problems, the kernels specialised
    for the common configuration under the adaptive schedule against the host loop"""
    name, ok = _compare(robots.panda(), dict(memetic_population_size=128), 512, robots.PANDA_HOME, 1234, 4096, "config 2")
    assert name.startswith("pik_common::") and ok > 0.97


@pytest.mark.parametrize("kw,what", [
    (dict(memetic_population_size=32, memetic_elite_size=5, memetic_max_generations=30), "five elites (general kernels)"),
    (dict(memetic_population_size=48, center_joints_weight=0.05, minimal_displacement_weight=0.01, cost_threshold=0.2,
          memetic_max_generations=25), "joint goals"),
    (dict(memetic_population_size=24, memetic_max_generations=15, return_approximate_solution=1), "approximate mode"),
    (dict(mode=1, gd_max_iters=60), "local mode"),
    (dict(memetic_population_size=40, memetic_max_generations=20, gd_step_size=0.01), "a gradient step too large for the angle addition"),
    (dict(memetic_population_size=24, memetic_max_generations=12, memetic_num_threads=2), "two species"),
    (dict(memetic_population_size=20, memetic_max_generations=10, memetic_num_threads=3, memetic_stop_on_first_solution=0,
          return_approximate_solution=1), "three species, no stop on the first solution"),
])
def test_panda_parameter_sets(kw, what):
    _compare(robots.panda(), kw, 96, None, 77, 5, what, reachable=(what != "approximate mode"))


@pytest.mark.parametrize("i", range(int(os.environ.get("PIK_PRODUCT_CHAINS", "24"))))  # (a soak: PIK_PRODUCT_CHAINS=120)
def test_generated_chains(i):
    """random chains (arbitrary axes, prismatic and continuous joints) x random parameters, one species"""
    rng = np.random.default_rng(0x9A0 + i)
    ch = random_chain(rng, 2 + i % 11)
    while any(t not in (robots.REVOLUTE, robots.PRISMATIC) for t in ch.joint_type):
        ch = random_chain(rng, 2 + i % 11)
    kw = random_params(rng)
    kw.pop("memetic_num_threads", None)
    kw.pop("memetic_stop_on_first_solution", None)
    _compare(ch, kw, int(rng.integers(8, 60)), None, 100 + i, 3, f"chain {i} dof {ch.dof} {kw}")


@pytest.mark.parametrize("name", ["torso_dual_arm", "dual_ur5"])
@pytest.mark.parametrize("kw", [dict(memetic_population_size=32, memetic_max_generations=15),
                                dict(memetic_population_size=24, memetic_max_generations=10, center_joints_weight=0.05,
                                     cost_threshold=0.3),
                                dict(mode=1, gd_max_iters=40)])
def test_several_tip_frames(name, kw):
    """two-arm chains: the several-tip kernels (gradient with the accept evaluation, full line-search evaluations,
    the cooperative descent at 8 / 16 lanes under the adaptive schedule) against the host loop"""
    ch = robots.by_name(name)
    _compare(ch, kw, 48, None, 31, 2, f"{name} {kw}")
