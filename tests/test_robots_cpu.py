"""The robot tables `bench.py` and the tests name (pick_ik_amd/robots.py) against the CPU oracle -- no GPU: every robot
`by_name` knows builds, its home pose has its length, and the oracle's forward kinematics of the composed robots is
what their parts give (the nine-variable Panda on a torso: torso joints at zero = the Panda shifted by the torso's
fixed transforms; reference for the Panda itself: tests/goal_tests.cpp:177, FK(ready).z = 0.59027)."""
import numpy as np
import pytest

import pick_ik_amd as pk
from pick_ik_amd import robots

NAMES = ("panda", "ur5", "rr", "dual_ur5", "torso_dual_arm", "floating_panda", "panda_on_torso")


@pytest.mark.parametrize("name", NAMES)
def test_robot_tables_are_consistent(name):
    ch = robots.by_name(name)
    d = ch.dof
    assert 1 <= d <= 16
    assert ch.qmin.shape == ch.qmax.shape == (d,) and np.all(ch.qmin <= ch.qmax)
    assert ch.bounded.shape == (d,)


def test_home_poses_have_their_robots_length():
    assert robots.PANDA_HOME.shape == (7,) and robots.UR5_HOME.shape == (6,)
    assert robots.PANDA_ON_TORSO_HOME.shape == (robots.panda_on_torso().dof,) == (9,)
    assert robots.FLOATING_PANDA_HOME.shape == (robots.floating_panda().dof,) == (14,)


def test_panda_on_torso_is_the_panda_behind_two_more_joints(oracle_mod):
    O = oracle_mod
    arm, whole = O.Oracle(robots.panda()), O.Oracle(robots.panda_on_torso())
    rng = np.random.default_rng(9)
    q = rng.uniform(robots.panda().qmin, robots.panda().qmax, size=(16, 7))
    a = arm.fk(q)
    w = whole.fk(np.concatenate([np.zeros((16, 2)), q], axis=1))
    # torso straight: yaw about z at 0.4 m, pitch about y 0.25 m above it, the arm's base 0.1 m forward and 0.2 m up --
    # the Panda's own first origin (0 0 0.333) is replaced by that mount, so the tip moves by (0.1, 0, 0.85 - 0.333)
    np.testing.assert_allclose(w[:, :3], a[:, :3] + np.array([0.1, 0.0, 0.4 + 0.25 + 0.2 - 0.333]), rtol=0, atol=1e-12)
    np.testing.assert_allclose(np.abs((w[:, 3:] * a[:, 3:]).sum(axis=1)), 1.0, rtol=0, atol=1e-12)
    # ready pose of the reference's tests
    z = arm.fk(robots.PANDA_HOME[None])[0, 2]
    assert abs(z - 0.59027) < 5e-5  # (the reference prints five digits)
