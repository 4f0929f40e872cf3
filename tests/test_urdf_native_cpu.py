"""The NATIVE robot-description reader of the C ABI (pikamd_urdf_extract, pick_ik_amd/csrc/pik_urdf.hpp;
Robot::from / get_link_indices / get_active_variable_indices of reference src/robot.cpp:44-160 over
URDF) against the Python reader on every fixture of tests/test_urdf_cpu.py: same joints, same
folded origins, same limits, same variable order; same error behaviour.  No GPU needed."""
import numpy as np
import pytest

import pick_ik_amd as pk
from pick_ik_amd import robots
from pick_ik_amd.solver import urdf_extract
from pick_ik_amd.urdf import chain_from_urdf, chain_to_urdf, multi_chain_from_urdf
from tests.test_urdf_cpu import DUAL, PANDA_TAIL_URDF


@pytest.fixture(scope="module", autouse=True)
def built():
    import __graft_entry__ as g
    g.build()


def same_chain(a, b):
    assert a.dof == b.dof
    np.testing.assert_allclose(a.origin_xyz_rpy, b.origin_xyz_rpy, rtol=0, atol=1e-15)
    np.testing.assert_array_equal(a.axis, b.axis)
    np.testing.assert_array_equal(a.joint_type, b.joint_type)
    np.testing.assert_allclose(a.tip_xyz_rpy, b.tip_xyz_rpy, rtol=0, atol=1e-15)
    for f in ("qmin", "qmax", "vmax", "bounded"):
        np.testing.assert_array_equal(getattr(a, f), getattr(b, f))


@pytest.mark.parametrize("name", ["panda", "ur5", "rr"])
def test_tables_round_trip(name, oracle_mod):
    text = chain_to_urdf(robots.by_name(name))
    native, names = urdf_extract(text, "base", "tip")
    same_chain(native, chain_from_urdf(text, "base", "tip"))
    assert names == [f"joint{i + 1}" for i in range(native.dof)]
    # and through the oracle's FK: the native reader's chain IS the table's chain
    ch = robots.by_name(name)
    q = np.random.default_rng(1).uniform(ch.qmin, ch.qmax, size=(32, ch.dof))
    np.testing.assert_allclose(oracle_mod.Oracle(native).fk(q), oracle_mod.Oracle(ch).fk(q), atol=1e-12)


def test_fixed_prismatic_subchains_continuous_mimic():
    for base, tip in (("b", "hand"), ("b", "finger"), ("hand", "finger")):
        native, _ = urdf_extract(PANDA_TAIL_URDF, base, tip)
        same_chain(native, chain_from_urdf(PANDA_TAIL_URDF, base, tip))
    urdf = """<?xml version="1.0"?>
    <!-- a comment with <tags/> inside -->
    <robot name="m"><link name="a"/><link name="b"/><link name="c"/><link name="d"/>
      <joint name="j1" type="continuous"><parent link="a"/><child link="b"/><axis xyz="0 0 1"/>
        <limit velocity="3" effort="1"/></joint>
      <joint name='j2' type="revolute"><parent link="b"/><child link="c"/><origin xyz="1 0 0"/>
        <mimic joint="j1" multiplier="0" offset="0.25"/><limit lower="-1" upper="1" velocity="1" effort="1"/></joint>
      <joint name="j3" type="revolute"><parent link="c"/><child link="d"/><origin xyz="0 2 0" rpy="0.1 1.5707963267948966 -0.3"/>
        <axis xyz="0 1 0"/><limit lower="-1" upper="1" velocity="1" effort="1"/>
        <dynamics damping="0.1"/></joint>
      <material name="grey"><color rgba="0.5 0.5 0.5 1"/></material></robot>"""
    native, names = urdf_extract(urdf, "a", "d")
    same_chain(native, chain_from_urdf(urdf, "a", "d"))  # incl. the gimbal-lock rpy of j3's origin
    assert names == ["j1", "j3"] and list(native.bounded) == [0, 1]
    # a mimic joint that follows its master: both readers describe the same extra step
    follow = urdf.replace(' multiplier="0" offset="0.25"', ' multiplier="1.5" offset="-0.1"')
    native_f, _ = urdf_extract(follow, "a", "d")
    py_f = chain_from_urdf(follow, "a", "d")
    same_chain(native_f, py_f)
    assert len(native_f.mimic) == 1 and len(py_f.mimic) == 1
    for a, b in zip(native_f.mimic, py_f.mimic):
        assert (a.tip, a.after_variable, a.master_variable, a.joint_type, a.multiplier, a.offset) == \
               (b.tip, b.after_variable, b.master_variable, b.joint_type, b.multiplier, b.offset)
        np.testing.assert_array_equal(a.origin_xyz_rpy, b.origin_xyz_rpy)
        np.testing.assert_array_equal(a.axis, b.axis)
    for bad, msg in ((follow.replace('mimic joint="j1"', 'mimic joint="jx"'), "not a variable of the path"),
                     (urdf.replace('<joint name="j3" ', "<joint "), "no name")):
        with pytest.raises(Exception, match=msg):
            urdf_extract(bad, "a", "d")
        with pytest.raises(ValueError, match=msg):
            chain_from_urdf(bad, "a", "d")
    xml = """<robot name="r"><link name="a"/><link name="b"/>
      <joint name="j" type="revolute"><parent link="a"/><child link="b"/>
        <origin xyz="0 0 1"/><axis xyz="0 0 1"/><limit upper="1" velocity="1"/></joint></robot>"""
    native, _ = urdf_extract(xml, "a", "b")
    assert native.bounded[0] == 1 and native.qmin[0] == 0.0 and native.qmax[0] == 1.0


def test_several_tips():
    native, names = urdf_extract(DUAL, "base", ["lhand", "rhand"])
    ref, ref_names = multi_chain_from_urdf(DUAL, "base", ["lhand", "rhand"])
    assert names == ref_names == ["torso_yaw", "l_sh", "l_el", "r_sh", "r_sl"]
    assert native.n_tips == 2 and native.dof == 5
    for a, b in zip(native.tips, ref.tips):
        np.testing.assert_array_equal(a.variable, b.variable)
        np.testing.assert_allclose(a.origin_xyz_rpy, b.origin_xyz_rpy, rtol=0, atol=1e-15)
        np.testing.assert_array_equal(a.axis, b.axis)
        np.testing.assert_array_equal(a.joint_type, b.joint_type)
        np.testing.assert_allclose(a.tip_xyz_rpy, b.tip_xyz_rpy, rtol=0, atol=1e-15)
    for f in ("qmin", "qmax", "vmax", "bounded"):
        np.testing.assert_array_equal(getattr(native, f), getattr(ref, f))


def test_errors_are_codes_not_crashes():
    E = pk.PickIkAmdError
    with pytest.raises(E, match="link not found: nope"):
        urdf_extract(DUAL, "base", "nope")
    with pytest.raises(E, match="not a descendant"):
        urdf_extract(DUAL, "lhand", "base")
    with pytest.raises(E, match="no actuated joint"):
        urdf_extract(PANDA_TAIL_URDF, "l1", "hand")
    with pytest.raises(E, match="expected 3 numbers"):
        urdf_extract(DUAL.replace('xyz="0 0 0.4"', 'xyz="0 0"'), "base", "lhand")
    with pytest.raises(E, match="not a number"):
        urdf_extract(DUAL.replace('xyz="0 0 0.4"', 'xyz="0 zero 0.4"'), "base", "lhand")
    # (tip order never matters for a tree: paths from one base share a prefix, so new variables
    #  always come after the shared ones)
    assert urdf_extract(DUAL, "base", ["rhand", "lhand", "rhand"])[1] == ["torso_yaw", "r_sh", "r_sl", "l_sh", "l_el"]
    with pytest.raises(E, match="not supported"):
        urdf_extract(DUAL.replace('name="l_el" type="continuous"', 'name="l_el" type="helical"'), "base", "lhand")
    for junk in ("", "<hello", "<robot", "<robot><link name='a'></robot>", "<a><b></a></b>", "<notrobot/>",
                 '<robot><joint name="j" type="fixed"/></robot>'):
        with pytest.raises(E):
            urdf_extract("<x/>" if junk == "" else junk, "a", "b")
    with pytest.raises(E, match="n_tips"):
        urdf_extract(DUAL, "base", ["lhand"] * 9)
