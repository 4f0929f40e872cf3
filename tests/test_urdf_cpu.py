"""Chain extraction from URDF (pick_ik_amd/urdf.py) against the embedded tables, through the oracle's
FK: round trip, folded fixed joints (the Panda's joint8 + hand joint as they appear in the real
URDF), continuous / prismatic / mimic joints, error behaviour."""
import math

import numpy as np
import pytest

from pick_ik_amd import robots
from pick_ik_amd.urdf import chain_from_urdf, chain_to_urdf

PI = math.pi

PANDA_TAIL_URDF = """
<robot name="two_fixed">
  <link name="b"/><link name="l1"/><link name="l8"/><link name="hand"/><link name="finger"/>
  <joint name="j7" type="revolute"><parent link="b"/><child link="l1"/>
    <origin xyz="0.088 0 0" rpy="1.5707963267948966 0 0"/><axis xyz="0 0 1"/>
    <limit lower="-2.8973" upper="2.8973" velocity="2.61" effort="12"/></joint>
  <joint name="j8" type="fixed"><parent link="l1"/><child link="l8"/><origin xyz="0 0 0.107" rpy="0 0 0"/></joint>
  <joint name="hand_joint" type="fixed"><parent link="l8"/><child link="hand"/>
    <origin xyz="0 0 0" rpy="0 0 -0.7853981633974483"/></joint>
  <joint name="finger_joint" type="prismatic"><parent link="hand"/><child link="finger"/>
    <origin xyz="0 0 0.0584"/><axis xyz="0 1 0"/><limit lower="0" upper="0.04" velocity="0.2" effort="20"/></joint>
</robot>"""


@pytest.mark.parametrize("name", ["panda", "ur5", "rr"])
def test_round_trip_matches_table(oracle_mod, name):
    ch = robots.by_name(name)
    back = chain_from_urdf(chain_to_urdf(ch), "base", "tip")
    assert back.dof == ch.dof
    np.testing.assert_array_equal(back.qmin, ch.qmin)
    np.testing.assert_array_equal(back.bounded, ch.bounded)
    o1, o2 = oracle_mod.Oracle(ch), oracle_mod.Oracle(back)
    q = np.random.default_rng(0).uniform(ch.qmin, ch.qmax, size=(64, ch.dof))
    np.testing.assert_allclose(o2.fk(q)[:, :3], o1.fk(q)[:, :3], atol=1e-12)
    np.testing.assert_allclose(o2.variables(), o1.variables(), atol=1e-15)


def test_fixed_joints_fold_into_tip_and_subchains(oracle_mod):
    ch = chain_from_urdf(PANDA_TAIL_URDF, "b", "hand")
    assert ch.dof == 1 and ch.bounded[0] == 1 and ch.vmax[0] == 2.61
    # joint8 (z 0.107) then hand joint (yaw -pi/4), exactly the tip of the embedded Panda table
    np.testing.assert_allclose(ch.tip_xyz_rpy, robots.panda().tip_xyz_rpy, atol=1e-15)
    # a longer tip selects the prismatic finger joint too
    ch2 = chain_from_urdf(PANDA_TAIL_URDF, "b", "finger")
    assert ch2.dof == 2 and ch2.joint_type[1] == robots.PRISMATIC
    np.testing.assert_allclose(ch2.origin_xyz_rpy[1], [0, 0, 0.107 + 0.0584, 0, 0, -PI / 4], atol=1e-15)
    o = oracle_mod.Oracle(ch2)
    p0, p1 = o.fk([0.3, 0.0])[0], o.fk([0.3, 0.02])[0]
    assert np.linalg.norm(p1[:3] - p0[:3]) == pytest.approx(0.02, abs=1e-15)
    # a sub-chain that starts at an inner link
    ch3 = chain_from_urdf(PANDA_TAIL_URDF, "hand", "finger")
    assert ch3.dof == 1 and ch3.joint_type[0] == robots.PRISMATIC


def test_continuous_and_mimic_and_errors():
    urdf = """<robot name="m"><link name="a"/><link name="b"/><link name="c"/><link name="d"/>
      <joint name="j1" type="continuous"><parent link="a"/><child link="b"/><axis xyz="0 0 1"/>
        <limit velocity="3" effort="1"/></joint>
      <joint name="j2" type="revolute"><parent link="b"/><child link="c"/><origin xyz="1 0 0"/>
        <mimic joint="j1" multiplier="0" offset="0"/><limit lower="-1" upper="1" velocity="1" effort="1"/></joint>
      <joint name="j3" type="revolute"><parent link="c"/><child link="d"/><origin xyz="0 2 0"/>
        <axis xyz="0 1 0"/><limit lower="-1" upper="1" velocity="1" effort="1"/></joint></robot>"""
    ch = chain_from_urdf(urdf, "a", "d")
    assert ch.dof == 2 and list(ch.bounded) == [0, 1]          # continuous -> unbounded; mimic skipped
    np.testing.assert_allclose(ch.origin_xyz_rpy[1][:3], [1, 2, 0])  # constant mimic joint folded at its offset
    # a constant mimic joint at a non-zero offset is a fixed rotation about its axis (default axis x)
    ch2 = chain_from_urdf(urdf.replace('offset="0"', 'offset="0.5"'), "a", "d")
    np.testing.assert_allclose(ch2.origin_xyz_rpy[1], [1, 2 * np.cos(0.5), 2 * np.sin(0.5), 0.5, 0, 0], atol=1e-15)
    # a mimic joint that FOLLOWS its master (the reference's FK moves it, src/fk_moveit.cpp:22): one more step of
    # the chain behind its master's joint, the next joint's origin starts behind it
    ch3 = chain_from_urdf(urdf.replace(' multiplier="0" offset="0"', ' multiplier="-0.5" offset="0.2"'), "a", "d")
    assert ch3.dof == 2 and len(ch3.mimic) == 1
    m = ch3.mimic[0]
    assert (m.after_variable, m.master_variable, m.multiplier, m.offset, m.joint_type) == (0, 0, -0.5, 0.2, 0)
    np.testing.assert_allclose(m.origin_xyz_rpy, [1, 0, 0, 0, 0, 0])
    np.testing.assert_allclose(m.axis, [1, 0, 0])
    np.testing.assert_allclose(ch3.origin_xyz_rpy[1], [0, 2, 0, 0, 0, 0])
    # ... whose master must be a variable of the path
    with pytest.raises(ValueError, match="not a variable of the path"):
        chain_from_urdf(urdf.replace(' multiplier="0" offset="0"', "").replace('mimic joint="j1"', 'mimic joint="jx"'), "a", "d")
    with pytest.raises(ValueError, match="no name"):
        chain_from_urdf(urdf.replace('<joint name="j3" ', "<joint "), "a", "d")
    with pytest.raises(ValueError, match="link not found: nope"):
        chain_from_urdf(urdf, "a", "nope")
    with pytest.raises(ValueError, match="not a descendant"):
        chain_from_urdf(urdf, "d", "a")
    with pytest.raises(ValueError, match="no actuated joint"):
        chain_from_urdf(PANDA_TAIL_URDF, "l1", "hand")


DUAL = """<robot name="dual">
  <link name="base"/><link name="torso"/><link name="l1"/><link name="l2"/><link name="lhand"/>
  <link name="r1"/><link name="r2"/><link name="rhand"/>
  <joint name="torso_yaw" type="revolute"><parent link="base"/><child link="torso"/>
    <origin xyz="0 0 0.4"/><axis xyz="0 0 1"/><limit lower="-1.5" upper="1.5" velocity="1"/></joint>
  <joint name="l_sh" type="revolute"><parent link="torso"/><child link="l1"/>
    <origin xyz="0 0.2 0.3" rpy="1.5707963267948966 0 0"/><axis xyz="0 0 1"/><limit lower="-2" upper="2" velocity="2"/></joint>
  <joint name="l_el" type="continuous"><parent link="l1"/><child link="l2"/>
    <origin xyz="0.3 0 0"/><axis xyz="0 1 0"/><limit velocity="2"/></joint>
  <joint name="l_fix" type="fixed"><parent link="l2"/><child link="lhand"/><origin xyz="0.25 0 0"/></joint>
  <joint name="r_sh" type="revolute"><parent link="torso"/><child link="r1"/>
    <origin xyz="0 -0.2 0.3" rpy="-1.5707963267948966 0 0"/><axis xyz="0 0 1"/><limit lower="-2" upper="2" velocity="2"/></joint>
  <joint name="r_sl" type="prismatic"><parent link="r1"/><child link="r2"/>
    <origin xyz="0.3 0 0"/><axis xyz="1 0 0"/><limit lower="0" upper="0.2" velocity="0.5"/></joint>
  <joint name="r_fix" type="fixed"><parent link="r2"/><child link="rhand"/><origin xyz="0.1 0 0" rpy="0 0 0.5"/></joint>
</robot>"""


def test_multi_tip_from_urdf(oracle_mod):
    """Two tips sharing the torso joint: variables = union of the paths' joints, in path order."""
    from pick_ik_amd.urdf import multi_chain_from_urdf
    O = oracle_mod
    mc, names = multi_chain_from_urdf(DUAL, "base", ["lhand", "rhand"])
    assert names == ["torso_yaw", "l_sh", "l_el", "r_sh", "r_sl"]
    assert mc.n_tips == 2 and mc.dof == 5
    assert mc.tips[0].variable.tolist() == [0, 1, 2] and mc.tips[1].variable.tolist() == [0, 3, 4]
    assert mc.bounded.tolist() == [1, 1, 0, 1, 1] and mc.tips[1].joint_type.tolist() == [0, 0, robots.PRISMATIC]
    # each tip of the tree is the single chain to that tip
    o = O.Oracle(mc)
    rng = np.random.default_rng(0)
    q = rng.uniform(-1, 1, size=(20, 5))
    q[:, 4] = np.abs(q[:, 4]) * 0.2
    f = o.fk(q)
    for k, (tip, cols) in enumerate((("lhand", [0, 1, 2]), ("rhand", [0, 3, 4]))):
        single = O.Oracle(chain_from_urdf(DUAL, "base", tip))
        np.testing.assert_array_equal(f[:, k], single.fk(q[:, cols]))
    with pytest.raises(ValueError):
        multi_chain_from_urdf(DUAL, "base", ["lhand", "nohand"])


def test_limit_defaults_and_malformed_attributes():
    """<limit> without `lower` (urdfdom default 0) is still a bounded joint; an xyz / rpy / axis
    attribute with the wrong number of values raises ValueError."""
    from pick_ik_amd.urdf import chain_from_urdf
    xml = """<robot name="r"><link name="a"/><link name="b"/>
      <joint name="j" type="revolute"><parent link="a"/><child link="b"/>
        <origin xyz="0 0 1"/><axis xyz="0 0 1"/><limit upper="1" velocity="1"/></joint></robot>"""
    ch = chain_from_urdf(xml, "a", "b")
    assert ch.bounded[0] == 1 and ch.qmin[0] == 0.0 and ch.qmax[0] == 1.0
    with pytest.raises(ValueError, match="expected 3 numbers"):
        chain_from_urdf(xml.replace('xyz="0 0 1"/><axis', 'xyz="0 0"/><axis'), "a", "b")
