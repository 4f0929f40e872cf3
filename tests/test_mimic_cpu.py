"""Mimic joints on the path (CPU).  A mimic joint is no variable (reference src/robot.cpp:144-150) but the
reference's forward kinematics moves it with its master (RobotState::setJointGroupPositions -> updateMimicJoints,
src/fk_moveit.cpp:22).  The oracle's chain product with such a step must equal, BIT FOR BIT, the product of the
chain in which that joint is an ordinary variable set to multiplier * master + offset -- in every math mode."""
import dataclasses

import numpy as np
import pytest

from pick_ik_amd import robots


def with_mimic(rng, full, k, master, mult, off):
    """the chain `full` with its joint k turned into a mimic of variable `master` (k != master)"""
    keep = [j for j in range(full.dof) if j != k]
    new_index = {j: i for i, j in enumerate(keep)}
    m = robots.MimicJoint(after_variable=(new_index[k - 1] if k > 0 else -1), master_variable=new_index[master],
                          origin_xyz_rpy=tuple(full.origin_xyz_rpy[k]), axis=tuple(full.axis[k]), multiplier=mult,
                          offset=off, joint_type=int(full.joint_type[k]))
    ch = robots._chain(full.name + "_mimic", full.origin_xyz_rpy[keep], full.axis[keep], full.tip_xyz_rpy, full.qmin[keep],
                       full.qmax[keep], full.vmax[keep], bounded=full.bounded[keep], joint_type=full.joint_type[keep])
    return dataclasses.replace(ch, mimic=(m,)), keep


def expand(q, keep, k, master_full, mult, off, dof):
    out = np.zeros((len(q), dof))
    out[:, keep] = q
    out[:, k] = mult * out[:, master_full] + off
    return out


CASES = [("panda", 3, 1, -0.6, 0.2), ("panda", 0, 4, 0.5, -0.1), ("ur5", 5, 2, 1.0, 0.0), ("panda", 6, 6 - 1, 2.0, 0.3)]


@pytest.mark.parametrize("name,k,master,mult,off", CASES)
@pytest.mark.parametrize("mode", ["libm", "portable", "fma"])
def test_oracle_mimic_step_equals_the_joint_as_a_variable(oracle_mod, name, k, master, mult, off, mode):
    O = oracle_mod
    full = robots.by_name(name)
    rng = np.random.default_rng(k)
    ch, keep = with_mimic(rng, full, k, master, mult, off)
    q = rng.uniform(ch.qmin, ch.qmax, size=(50, ch.dof))
    with O.math_mode(mode):
        a = O.Oracle(ch).fk(q)
        b = O.Oracle(full).fk(expand(q, keep, k, master, mult, off, full.dof))
    np.testing.assert_array_equal(a, b)


def test_prismatic_mimic_and_two_in_a_row(oracle_mod):
    O = oracle_mod
    origins = [[0, 0, 0.2, 0, 0, 0], [0.1, 0, 0, 0.3, 0, 0], [0, 0.2, 0, 0, 0.4, 0], [0.3, 0, 0, 0, 0, 0.5], [0, 0, 0.1, 0.2, 0, 0]]
    axes = [[0, 0, 1], [1, 0, 0], [0, 1, 0.2], [0, 1, 0], [0, 0, 1]]
    jt = np.array([0, 1, 0, 0, 0], np.int32)
    full = robots._chain("f", origins, axes, [0.1, 0, 0, 0, 0, 0], [-1.5] * 5, [1.5] * 5, [1] * 5, joint_type=jt)
    # joints 1 (prismatic) and 2 both follow variable 0; variables: 0, 3, 4
    keep = [0, 3, 4]
    ms = (robots.MimicJoint(0, 0, tuple(origins[1]), tuple(axes[1]), 0.1, 0.05, joint_type=1),
          robots.MimicJoint(0, 0, tuple(origins[2]), tuple(axes[2]), -1.2, 0.0, joint_type=0))
    ch = robots._chain("m", np.array(origins)[keep], np.array(axes)[keep], [0.1, 0, 0, 0, 0, 0], [-1.5] * 3, [1.5] * 3, [1] * 3)
    ch = dataclasses.replace(ch, mimic=ms)
    rng = np.random.default_rng(0)
    q = rng.uniform(-1.5, 1.5, size=(40, 3))
    qf = np.zeros((40, 5))
    qf[:, keep] = q
    qf[:, 1] = 0.1 * q[:, 0] + 0.05
    qf[:, 2] = -1.2 * q[:, 0] + 0.0
    for mode in ("portable", "fma"):
        with O.math_mode(mode):
            np.testing.assert_array_equal(O.Oracle(ch).fk(q), O.Oracle(full).fk(qf))
