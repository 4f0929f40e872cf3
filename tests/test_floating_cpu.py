"""Floating joints (moveit::core::FloatingJointModel: seven variables, one transform
Translation(v0 v1 v2) * Quaterniond(w = v6, v3, v4, v5), reference src/forward_kinematics.cpp:64-70) in the
oracle and in the chain description of the C ABI -- the parts that need no GPU."""
import ctypes as C

import numpy as np
import pytest

from pick_ik_amd import robots
from pick_ik_amd import solver as S


def _rot(q):  # Eigen toRotationMatrix of an UNNORMALISED quaternion (w x y z)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def _rpy(r, p, y):
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr], [-sp, cp * sr, cp * cr]])


def test_oracle_floating_base_equals_the_composed_transform(oracle_mod):
    O = oracle_mod
    ch = robots.floating_panda()
    arm = robots.panda()
    rng = np.random.default_rng(3)
    q = rng.uniform(ch.qmin, ch.qmax, size=(50, ch.dof))
    q[:10, 3:7] /= np.linalg.norm(q[:10, 3:7], axis=1, keepdims=True)  # some unit quaternions too
    got = O.Oracle(ch).fk(q)
    arm_pose = O.Oracle(arm).fk(q[:, 7:])
    o6 = ch.origin_xyz_rpy[0]
    for i in range(len(q)):
        Rb = _rpy(*o6[3:]) @ _rot([q[i, 6], q[i, 3], q[i, 4], q[i, 5]])
        tb = _rpy(*o6[3:]) @ q[i, :3] + o6[:3]
        w, x, y, z = arm_pose[i, 3:]
        Ra = _rot([w, x, y, z])
        np.testing.assert_allclose(got[i, :3], Rb @ arm_pose[i, :3] + tb, atol=1e-12)
        if i < 10:  # a proper rotation: compare the orientation as well (quaternion up to sign)
            Rt = Rb @ Ra
            gw, gx, gy, gz = got[i, 3:]
            np.testing.assert_allclose(_rot([gw, gx, gy, gz]), Rt, atol=1e-12)


def test_oracle_solves_on_a_floating_base(oracle_mod):
    """14 variables, the base free to move: every target of the fixed-base arm shifted by a base offset is
    reachable; the memetic search finds most of them"""
    O = oracle_mod
    ch = robots.floating_panda()
    o = O.Oracle(ch)
    rng = np.random.default_rng(5)
    q = rng.uniform(ch.qmin, ch.qmax, size=(24, ch.dof))
    q[:, 3:7] /= np.linalg.norm(q[:, 3:7], axis=1, keepdims=True)
    goal = o.fk(q)
    seed = np.tile(robots.FLOATING_PANDA_HOME, (24, 1))
    sol, st, cost, stats = o.solve_batch(O.default_params(memetic_population_size=32), goal, seed, rng_seed=3,
                                         num_threads=O.max_threads())
    assert (st == O.SUCCESS).mean() > 0.5
    for b in np.flatnonzero(st == O.SUCCESS):
        assert o.cost(O.default_params(), goal[b], seed[b], sol[b])[1][0] == 1


def test_chain_description_validates_floating_joints():
    """pikamd_create checks the description before it looks for a device"""
    ch = robots.floating_panda()
    L = S.lib()

    def create(jt):
        k = [S._f64(ch.origin_xyz_rpy), S._f64(ch.axis), np.ascontiguousarray(jt, dtype=np.int32), S._f64(ch.tip_xyz_rpy)]
        lim = [S._f64(ch.qmin), S._f64(ch.qmax), S._f64(ch.vmax), np.ascontiguousarray(ch.bounded, dtype=np.uint8)]
        c = S._Chain(ch.dof, S._dp(k[0]), S._dp(k[1]), S._ip(k[2]), S._dp(k[3]), S._dp(lim[0]), S._dp(lim[1]),
                     S._dp(lim[2]), lim[3].ctypes.data_as(C.POINTER(C.c_uint8)))
        h = C.c_void_p()
        rc = L.pikamd_create(C.byref(c), 0, C.byref(h))
        msg = L.pikamd_last_error().decode()
        if h:
            L.pikamd_destroy(h)
        return rc, msg

    bad = ch.joint_type.copy()
    bad[3] = robots.REVOLUTE  # a hole in the seven variables
    rc, msg = create(bad)
    assert rc == -1 and "floating joint" in msg, (rc, msg)
    bad = ch.joint_type.copy()
    bad[:7] = bad[:7][::-1]  # wrong order
    rc, msg = create(bad)
    assert rc == -1 and "floating joint" in msg, (rc, msg)
    rc, msg = create(ch.joint_type)  # a valid description: the only thing missing here is a device
    import torch
    if not torch.cuda.is_available():
        assert rc == -2 and "no HIP device" in msg, (rc, msg)
