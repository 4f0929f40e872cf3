"""CPU: the kernels' arithmetic (pik_math.hpp) and the host model extraction (pik_host.hpp)
compiled for the host with g++ and compared with the oracle -- FK through the Denavit-Hartenberg form of the chain,
cost + solution verdict, frame-based gradient probes vs literal central differences, the in-house
sincos/atan2 and Philox.  The fast flavour must agree to rounding; the PIK_STRICT flavour (compiled
-ffp-contract=off) must agree BIT FOR BIT with the oracle's portable-math mode -- the same check the
GPU strict build passes, available without a GPU."""
import math
import os
import subprocess

import numpy as np
import pytest

from pick_ik_amd import robots

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "host_math_check.cpp")


CLANG = "/opt/rocm/lib/llvm/bin/clang++"  # the host compiler hipcc uses for the library itself


# the exact flavours and the oracle math mode each one is bit-identical to
EXACT_MODE = {"strict": "portable", "exact_fma": "fma"}


def build(strict, compiler="g++"):
    """strict: False (fast flavour), True / "strict" (plain exact flavour) or "exact_fma" (the fused one)"""
    flavour = "strict" if strict is True else (strict or "")
    tag = ("_" + flavour if flavour else "") + ("" if compiler == "g++" else "_clang")
    exe = os.path.join(ROOT, "tests", "native", "host_math_check" + tag)
    deps = [SRC] + [os.path.join(ROOT, "pick_ik_amd", "csrc", f) for f in ("pik_math.hpp", "pik_host.hpp")]
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(map(os.path.getmtime, deps)):
        flags = {"": [], "strict": ["-DPIK_STRICT=1", "-ffp-contract=off"],
                 "exact_fma": ["-DPIK_STRICT=1", "-DPIK_EXACT_FMA=1", "-ffp-contract=off"]}[flavour]
        opt = ["-O2", "-mfma"] if compiler == "g++" else ["-O3"]
        subprocess.run([compiler, "-std=c++17", *opt, *flags, SRC, "-o", exe], check=True)
    return exe


def run(exe, ch, weights, q, goal, seed):
    lines = [f"{ch.dof} {len(q)}"]
    for arr in (ch.origin_xyz_rpy, ch.axis, ch.tip_xyz_rpy, ch.qmin, ch.qmax, ch.vmax):
        lines.append(" ".join(repr(float(x)) for x in np.ravel(arr)))
    lines.append(" ".join(f"{int(t)} {int(b)}" for t, b in zip(ch.joint_type, ch.bounded)))
    lines.append(" ".join(repr(float(w)) for w in weights))
    for i in range(len(q)):
        lines.append(" ".join(repr(float(x)) for x in np.concatenate([q[i], goal[i], seed[i]])))
    r = subprocess.run([exe], input="\n".join(lines), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = {"fk": [], "cost": [], "grad": [], "sincos": [], "atan2": [], "philox": [], "sincosdelta": [], "class": [],
           "fkuz": []}
    for ln in r.stdout.splitlines():
        k, *v = ln.split()
        out[k].append(v)
    return out


def general_chain():
    import dataclasses
    ch = robots.panda()
    axis = ch.axis.copy()
    axis[2] = [0.3, -0.5, 0.8]
    axis[4] = [0.0, -1.0, 0.0]
    axis[5] = [0.0, 0.0, -1.0]
    return dataclasses.replace(ch, axis=axis, joint_type=np.array([0, 0, 0, 1, 0, 0, 0], np.int32))


@pytest.mark.parametrize("strict", [False, "strict", "exact_fma"], ids=["fast", "strict", "exact_fma"])
@pytest.mark.parametrize("name", ["panda", "ur5", "rr", "general"])
def test_device_math_on_host(oracle_mod, name, strict):
    O = oracle_mod
    ch = general_chain() if name == "general" else robots.by_name(name)
    o = O.Oracle(ch)
    rng = np.random.default_rng(5)
    n = 40
    q = rng.uniform(ch.qmin, ch.qmax, size=(n, ch.dof))
    near = q + rng.normal(0, 1, size=q.shape) * np.logspace(-5, -1, n)[:, None]
    goal = o.fk(near)
    seed = rng.uniform(ch.qmin, ch.qmax, size=(n, ch.dof))
    weights = (0.3, 0.2, 0.1)
    out = run(build(strict), ch, weights, q, goal, seed)
    p = O.default_params(center_joints_weight=weights[0], avoid_joint_limits_weight=weights[1],
                         minimal_displacement_weight=weights[2])
    fk = np.array(out["fk"], dtype=float)
    cost = np.array([c[0] for c in out["cost"]], dtype=float)
    sol = np.array([c[1] for c in out["cost"]], dtype=int)
    with O.math_mode(EXACT_MODE[strict] if strict else "libm"):
        ofk = o.fk(q)
        oc = np.array([o.cost(p, goal[i], seed[i], q[i]) for i in range(n)])
    ocost, osol = oc[:, 0, 0], oc[:, 1, 0].astype(int)
    if strict:
        np.testing.assert_array_equal(fk, ofk)
        np.testing.assert_array_equal(cost, ocost)
        np.testing.assert_array_equal(sol, osol)
    else:
        np.testing.assert_allclose(fk[:, :3], ofk[:, :3], rtol=0, atol=1e-12)
        sgn = np.sign((fk[:, 3:] * ofk[:, 3:]).sum(axis=1, keepdims=True))
        np.testing.assert_allclose(fk[:, 3:] * sgn, ofk[:, 3:], rtol=0, atol=1e-12)
        np.testing.assert_allclose(cost, ocost, rtol=1e-11, atol=1e-18)
        assert (sol == osol).mean() >= 0.95
        # frame-based probes == central differences of the full cost (relative to the gradient scale)
        g = np.array(out["grad"], dtype=float).reshape(n, ch.dof, 2)
        scale = np.abs(g[:, :, 1]).max(axis=1, keepdims=True) + 1e-300
        assert (np.abs(g[:, :, 0] - g[:, :, 1]) / scale).max() < 1e-6
    # in-house transcendentals and Philox
    for x, s, c in out["sincos"]:
        x, s, c = float(x), float(s), float(c)
        assert abs(s - math.sin(x)) <= 4e-16 and abs(c - math.cos(x)) <= 4e-16
    for y, x, r in out["atan2"]:
        assert float(r) == pytest.approx(math.atan2(float(y), float(x)), abs=3e-16)
    assert out["philox"][0] == ["d16cfe09", "94fdcceb", "5001e420", "24126ea1"]
    # the line-search angle addition of the product build (steps up to 1e-3 rad)
    assert strict or len(out["sincosdelta"]) == 30
    for th, d, s2, c2 in out["sincosdelta"]:
        th, d, s2, c2 = float(th), float(d), float(s2), float(c2)
        assert abs(s2 - math.sin(th + d)) <= 4e-16 and abs(c2 - math.cos(th + d)) <= 4e-16


@pytest.mark.parametrize("flavour", ["strict", "exact_fma"])
@pytest.mark.parametrize("compiler", ["g++", CLANG], ids=["gcc", "clang"])
def test_fuzz_chains_strict_on_host(oracle_mod, compiler, flavour):
    """The randomly generated chains of tests/test_gpu_fuzz.py (1..16 joints, arbitrary axes,
    prismatic / continuous joints) through the strict arithmetic on the host: FK, cost and verdict
    bit-exact against the oracle -- with g++ AND with the clang hipcc uses for the library's host
    side (the model extraction must not depend on the compiler: a sin()/cos() pair that one
    compiler merges into sincos() and the other does not once moved an origin matrix by 1 ulp)."""
    if not os.path.exists(compiler) and compiler != "g++":
        pytest.skip("no rocm clang++")
    from tests.test_gpu_fuzz import make_case, N_CASES
    O = oracle_mod
    exe = build(flavour, compiler)
    weights = (0.3, 0.2, 0.1)
    p = O.default_params(center_joints_weight=weights[0], avoid_joint_limits_weight=weights[1],
                         minimal_displacement_weight=weights[2])
    for i in range(N_CASES):
        ch, _, q, seed, _, _ = make_case(i)
        q, seed = q[:24], seed[:24]
        o = O.Oracle(ch)
        with O.math_mode(EXACT_MODE[flavour]):
            goal = o.fk(np.clip(q + 0.01, None, None))
            ofk = o.fk(q)
            oc = np.array([o.cost(p, goal[k], seed[k], q[k]) for k in range(len(q))])
        out = run(exe, ch, weights, q, goal, seed)
        np.testing.assert_array_equal(np.array(out["fk"], dtype=float), ofk, err_msg=f"case {i} fk")
        np.testing.assert_array_equal(np.array([c[0] for c in out["cost"]], dtype=float),
                                      oc[:, 0, 0], err_msg=f"case {i} cost")
        np.testing.assert_array_equal(np.array([c[1] for c in out["cost"]], dtype=int),
                                      oc[:, 1, 0].astype(int), err_msg=f"case {i} verdict")


def test_fuzz_chains_fast_fk_on_host(oracle_mod):
    """The fast build's Denavit-Hartenberg form of the chain (pik_host.hpp build_dh: frames on the
    joint axes, x = common normal; parallel, intersecting and coincident axes, prismatic joints) on
    the randomly generated chains: FK within 1e-12 of the oracle, frame-based probes within 1e-6 of
    literal central differences."""
    from tests.test_gpu_fuzz import make_case, N_CASES
    O = oracle_mod
    exe = build(False)
    for i in range(N_CASES):
        ch = make_case(i)[0]
        o = O.Oracle(ch)
        rng = np.random.default_rng(100 + i)
        lo = np.where(ch.bounded == 1, ch.qmin, -3.0)
        hi = np.where(ch.bounded == 1, ch.qmax, 3.0)
        q = rng.uniform(lo, hi, size=(24, ch.dof))
        goal = o.fk(q + 0.01)
        out = run(exe, ch, (0.3, 0.2, 0.1), q, goal, q)
        fk = np.array(out["fk"], dtype=float)
        ofk = o.fk(q)
        np.testing.assert_allclose(fk[:, :3], ofk[:, :3], rtol=0, atol=1e-12, err_msg=f"case {i}")
        sgn = np.sign((fk[:, 3:] * ofk[:, 3:]).sum(axis=1, keepdims=True))
        np.testing.assert_allclose(fk[:, 3:] * sgn, ofk[:, 3:], rtol=0, atol=1e-12, err_msg=f"case {i}")
        g = np.array(out["grad"], dtype=float).reshape(len(q), ch.dof, 2)
        scale = np.abs(g[:, :, 1]).max(axis=1, keepdims=True) + 1e-300
        assert (np.abs(g[:, :, 0] - g[:, :, 1]) / scale).max() < 1e-6, f"case {i}"


@pytest.mark.parametrize("eps", [1e-3, 1e-5, 1e-7, 1e-9, 4.9e-12, 3e-12, 1e-13])
def test_nearly_parallel_axes_fast_fk_on_host(oracle_mod, eps):
    """Consecutive joint axes that are nearly but not exactly parallel (a URDF that writes
    1.57079632679 for pi/2 is 4.9e-12 rad off): the common normal of such a pair lies ~L / angle away,
    so the plain Denavit-Hartenberg construction lost 1e-16 L / angle of FK accuracy (3e-5 m at
    3e-12 rad).  build_dh gives these pairs a general constant step; FK must stay within 1e-12."""
    import dataclasses
    O = oracle_mod
    exe = build(False)
    ur5 = robots.ur5()
    worst = 0.0
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
        rng = np.random.default_rng(7 + variant)
        q = rng.uniform(ch.qmin, ch.qmax, size=(32, ch.dof))
        goal = o.fk(q + 0.01)
        out = run(exe, ch, (0.3, 0.2, 0.1), q, goal, q)
        fk = np.array(out["fk"], dtype=float)
        ofk = o.fk(q)
        worst = max(worst, np.abs(fk[:, :3] - ofk[:, :3]).max())
        np.testing.assert_allclose(fk[:, :3], ofk[:, :3], rtol=0, atol=1e-12, err_msg=f"variant {variant}")
        sgn = np.sign((fk[:, 3:] * ofk[:, 3:]).sum(axis=1, keepdims=True))
        np.testing.assert_allclose(fk[:, 3:] * sgn, ofk[:, 3:], rtol=0, atol=1e-12)
        # the frame-based probes still see the right joint axes
        g = np.array(out["grad"], dtype=float).reshape(len(q), ch.dof, 2)
        scale = np.abs(g[:, :, 1]).max(axis=1, keepdims=True) + 1e-300
        assert (np.abs(g[:, :, 0] - g[:, :, 1]) / scale).max() < 1e-6
    print(f"eps {eps:g}: worst FK position error {worst:.2e} m")


@pytest.mark.parametrize("compiler", ["g++", CLANG], ids=["gcc", "clang"])
def test_chain_classes_and_their_sparse_products_on_host(oracle_mod, compiler):
    """The exact flavour picks its form by the chain's CLASS (pik_host.hpp make_chain_k): 1 = every joint about +z
    behind an origin that turns about its own x axis only (the Panda: the form that carries the benchmark), 2 = every
    axis exactly +x / +y / +z.  The class forms leave the products by the EXACT ones and zeros of the fixed
    transforms out of the chain product (pik_math.hpp x_iso_mul); their forward kinematics (fk_uz) must be, bit for
    bit, the oracle's full product -- reference robots and the generated chains of the GPU class fuzz."""
    if not os.path.exists(compiler) and compiler != "g++":
        pytest.skip("no rocm clang++")
    from tests.test_gpu_fuzz import axis_aligned_chain
    O = oracle_mod
    exe = build("exact_fma", compiler)
    rng = np.random.default_rng(0xC1A55)
    cases = [("panda", robots.panda(), 1), ("ur5", robots.ur5(), 2),
             ("panda_on_torso", robots.panda_on_torso(), 2)]  # (nine variables: the forms reach ten since round 6)
    all_z_of, dh_of = {}, {}
    for i in range(30):
        dof = 1 + i % 10
        all_z, dh = (i // 10) != 1, (i // 10) != 0  # 0-9: z joints, any origins; 10-19: mixed axes, x twists; 20-29: z, x twists
        cases.append((f"generated {i}", axis_aligned_chain(rng, dof, all_z=all_z, dh=dh), None))
        all_z_of[f"generated {i}"], dh_of[f"generated {i}"] = all_z, dh
    seen = set()
    for name, ch, want in cases:
        n = 12
        lo = np.where(ch.bounded == 1, ch.qmin, -3.0)
        hi = np.where(ch.bounded == 1, ch.qmax, 3.0)
        q = rng.uniform(lo, hi, size=(n, ch.dof))
        o = O.Oracle(ch)
        with O.math_mode("fma"):
            ofk = o.fk(q)
        out = run(exe, ch, (0.0, 0.0, 0.0), q, ofk, q)
        cls, okinds, tkind = int(out["class"][0][0]), int(out["class"][0][1], 16), int(out["class"][0][2])
        kinds = [(okinds >> (3 * j)) & 7 for j in range(ch.dof)]
        if want is not None:
            assert cls == want, (name, cls)
        if name == "panda":
            assert kinds == [4, 1, 1, 1, 1, 1, 1] and tkind == 3  # no twist, six x twists; the hand turns about z
        if name == "ur5":
            assert kinds == [4, 2, 4, 2, 4, 4] and tkind == 3
        if cls == 1:
            assert all(k in (1, 4) for k in kinds) and tkind in (3, 4), (name, kinds, tkind)
        if name.startswith("generated") and all_z_of[name] and dh_of[name] and ch.dof <= 10:
            assert cls in (0, 1), (name, cls)  # (0: an origin that happens to be the identity)
        seen.add(cls)
        np.testing.assert_array_equal(np.array(out["fk"], dtype=float), ofk, err_msg=f"{name} fk")
        if cls:
            np.testing.assert_array_equal(np.array(out["fkuz"], dtype=float), ofk, err_msg=f"{name} class {cls} fk_uz")
    assert seen >= {1, 2}
