"""Planar joints (moveit::core::PlanarJointModel: x, y, theta; reference src/forward_kinematics.cpp:72-79
hands them to JointModel::computeTransform): three consecutive variables of the chain.  The oracle's
FK against the formula written out directly, T_parent * O * Translation(x, y, 0) * Rz(theta), and the
two robot-description readers on a mobile manipulator."""
import numpy as np
import pytest

from pick_ik_amd import robots
from pick_ik_amd.urdf import _iso, chain_from_urdf

MOBILE = """<robot name="mobile">
  <link name="odom"/><link name="base"/><link name="l1"/><link name="l2"/><link name="tool"/>
  <joint name="virtual" type="planar"><parent link="odom"/><child link="base"/>
    <origin xyz="0.5 -0.25 0.1" rpy="0 0 0.4"/><limit lower="-2" upper="2" velocity="0.7"/></joint>
  <joint name="lift" type="prismatic"><parent link="base"/><child link="l1"/>
    <origin xyz="0.1 0 0.3"/><axis xyz="0 0 1"/><limit lower="0" upper="0.5" velocity="0.2"/></joint>
  <joint name="elbow" type="revolute"><parent link="l1"/><child link="l2"/>
    <origin xyz="0.2 0 0" rpy="1.5707963267948966 0 0"/><axis xyz="0 0 1"/><limit lower="-2" upper="2" velocity="1"/></joint>
  <joint name="fix" type="fixed"><parent link="l2"/><child link="tool"/><origin xyz="0.3 0 0"/></joint>
</robot>"""


def rz(t):
    T = np.eye(4)
    T[:2, :2] = [[np.cos(t), -np.sin(t)], [np.sin(t), np.cos(t)]]
    return T


def tr(x, y, z):
    T = np.eye(4)
    T[:3, 3] = [x, y, z]
    return T


def test_planar_fk_is_translation_times_rz(oracle_mod):
    ch = chain_from_urdf(MOBILE, "odom", "tool")
    assert ch.dof == 5
    assert ch.joint_type.tolist() == [robots.PLANAR_X, robots.PLANAR_Y, robots.PLANAR_THETA, robots.PRISMATIC,
                                      robots.REVOLUTE]
    assert ch.bounded.tolist() == [1, 1, 0, 1, 1] and ch.qmin[:2].tolist() == [-2, -2] and ch.vmax[0] == 0.7
    o = oracle_mod.Oracle(ch)
    rng = np.random.default_rng(0)
    q = rng.uniform(-1, 1, size=(50, 5))
    q[:, 3] = np.abs(q[:, 3]) * 0.5
    got = o.fk(q)
    for i in range(len(q)):
        x, y, th, lift, el = q[i]
        T = (_iso([0.5, -0.25, 0.1], [0, 0, 0.4]) @ tr(x, y, 0) @ rz(th) @ _iso([0.1, 0, 0.3], [0, 0, 0]) @ tr(0, 0, lift)
             @ _iso([0.2, 0, 0], [np.pi / 2, 0, 0]) @ rz(el) @ tr(0.3, 0, 0))
        np.testing.assert_allclose(got[i, :3], T[:3, 3], rtol=0, atol=1e-14)
        # orientation through the rotation's action on two vectors
        qw, qx, qy, qz = got[i, 3:]
        R = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)],
                      [2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)],
                      [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)]])
        np.testing.assert_allclose(R, T[:3, :3], rtol=0, atol=1e-13)


def test_native_reader_agrees_and_rejects_broken_planar_blocks():
    import __graft_entry__ as g
    g.build()
    import pick_ik_amd as pk
    from pick_ik_amd.solver import urdf_extract
    native, names = urdf_extract(MOBILE, "odom", "tool")
    ref = chain_from_urdf(MOBILE, "odom", "tool")
    assert names == ["virtual/x", "virtual/y", "virtual/theta", "lift", "elbow"]
    np.testing.assert_allclose(native.origin_xyz_rpy, ref.origin_xyz_rpy, rtol=0, atol=1e-15)
    for f in ("axis", "joint_type", "qmin", "qmax", "vmax", "bounded"):
        np.testing.assert_array_equal(getattr(native, f), getattr(ref, f))
    # the same robot on a free-flying base: seven variables (MoveIt's names and bounds), both readers agree
    FLY = MOBILE.replace('type="planar"', 'type="floating"')
    native, names = urdf_extract(FLY, "odom", "tool")
    ref = chain_from_urdf(FLY, "odom", "tool")
    assert names == [f"virtual/{v}" for v in ("trans_x", "trans_y", "trans_z", "rot_x", "rot_y", "rot_z", "rot_w")] + \
        ["lift", "elbow"]
    assert list(native.joint_type[:7]) == [5, 6, 7, 8, 9, 10, 11] and list(native.bounded[:7]) == [0, 0, 0, 1, 1, 1, 1]
    assert list(native.qmin[3:7]) == [-1.0] * 4 and list(native.qmax[3:7]) == [1.0] * 4
    np.testing.assert_allclose(native.origin_xyz_rpy, ref.origin_xyz_rpy, rtol=0, atol=1e-15)
    for f in ("axis", "joint_type", "qmin", "qmax", "vmax", "bounded"):
        np.testing.assert_array_equal(getattr(native, f), getattr(ref, f))
    with pytest.raises(pk.PickIkAmdError, match="not supported"):
        urdf_extract(MOBILE.replace('type="planar"', 'type="helical"'), "odom", "tool")


def test_host_model_extraction_checks_the_block(oracle_mod):
    """x, y, theta must be three consecutive variables (C ABI build_chain, compiled for the host)"""
    import dataclasses
    from tests.test_host_math_cpu import build, run
    ch = chain_from_urdf(MOBILE, "odom", "tool")
    o = oracle_mod.Oracle(ch)
    q = np.random.default_rng(1).uniform(-1, 1, size=(16, 5))
    goal = o.fk(q + 0.01)
    for strict in (False, True):
        out = run(build(strict), ch, (0.0, 0.0, 0.0), q, goal, q)
        fk = np.array(out["fk"], dtype=float)
        with oracle_mod.math_mode("portable" if strict else "libm"):
            ofk = o.fk(q)
        if strict:
            np.testing.assert_array_equal(fk, ofk)
        else:
            np.testing.assert_allclose(fk[:, :3], ofk[:, :3], rtol=0, atol=1e-12)
    bad = dataclasses.replace(ch, joint_type=np.array([2, 3, 0, 1, 0], np.int32))
    import subprocess
    with pytest.raises(AssertionError):
        run(build(False), bad, (0.0, 0.0, 0.0), q, goal, q)
    del subprocess
