"""Serial-chain descriptions (data only) for the robots BASELINE.json names.

PROVENANCE: these tables are NOT in the reference tree.  pick_ik loads the Panda from
``moveit_resources`` (``loadTestingRobotModel("panda")``, reference tests/ik_tests.cpp:241-242) and
builds the 2-link RR arm with ``RobotModelBuilder`` (tests/ik_tests.cpp:15-48).  The numbers below
are the public URDF values (franka_description / universal_robot ur5) recalled by the builder; the
only check the reference offers is FK(ready).z = 0.59027 (tests/goal_tests.cpp:177), which
``tests/test_oracle_golden.py`` asserts.

A chain is base -> tip: ``dof`` actuated joints, each with the fixed origin transform (URDF
``<origin xyz rpy>``) of its child link and a joint axis, followed by one fixed tip transform (all
fixed joints after the last actuated joint collapsed into a single xyz/rpy).
"""
from __future__ import annotations

import dataclasses
import math

import numpy as np

PI = math.pi

REVOLUTE = 0
PRISMATIC = 1
#: the three variables of a planar joint (consecutive; PLANAR_X carries the joint's origin)
PLANAR_X, PLANAR_Y, PLANAR_THETA = 2, 3, 4
#: the seven variables of a floating joint (consecutive, MoveIt's order trans_x trans_y trans_z rot_x rot_y
#: rot_z rot_w; FLOATING_TX carries the joint's origin): ONE transform Translation(t) * Quaternion(w, x, y, z)
FLOATING_TX, FLOATING_TY, FLOATING_TZ, FLOATING_RX, FLOATING_RY, FLOATING_RZ, FLOATING_RW = 5, 6, 7, 8, 9, 10, 11
FLOATING = (FLOATING_TX, FLOATING_TY, FLOATING_TZ, FLOATING_RX, FLOATING_RY, FLOATING_RZ, FLOATING_RW)


@dataclasses.dataclass(frozen=True)
class MimicJoint:
    """A mimic joint on a tip path: no variable (reference src/robot.cpp:144-150), moved with its master by the
    reference's forward kinematics (RobotState::setJointGroupPositions -> updateMimicJoints, src/fk_moveit.cpp:22).
    One more step of the chain product behind the joint of variable `after_variable` (-1: in front of the first), at
    multiplier * q[master_variable] + offset; its origin = the fixed transform from the previous moving joint of the
    path, and the next joint's origin starts behind it."""
    after_variable: int
    master_variable: int
    origin_xyz_rpy: tuple  # 6
    axis: tuple  # 3
    multiplier: float = 1.0
    offset: float = 0.0
    joint_type: int = 0  # REVOLUTE / PRISMATIC
    tip: int = 0


@dataclasses.dataclass(frozen=True)
class Chain:
    name: str
    origin_xyz_rpy: np.ndarray  # [dof][6]
    axis: np.ndarray  # [dof][3]
    joint_type: np.ndarray  # [dof] int32
    tip_xyz_rpy: np.ndarray  # [6]
    qmin: np.ndarray
    qmax: np.ndarray
    vmax: np.ndarray
    bounded: np.ndarray  # [dof] uint8
    mimic: tuple = ()  # of MimicJoint

    @property
    def dof(self) -> int:
        return int(self.origin_xyz_rpy.shape[0])


@dataclasses.dataclass(frozen=True)
class TipPath:
    """The joints between the base and one tip link (see MultiChain)."""
    variable: np.ndarray  # [n] int32: index of each joint's variable, strictly increasing
    origin_xyz_rpy: np.ndarray  # [n][6]
    axis: np.ndarray  # [n][3]
    joint_type: np.ndarray  # [n] int32
    tip_xyz_rpy: np.ndarray  # [6]


@dataclasses.dataclass(frozen=True)
class MultiChain:
    """Several tip links over one vector of `dof` active variables -- the plugin's `tip_frames`
    (reference src/pick_ik_plugin.cpp:57-69; the variables are the union of the joints on the way
    to any tip, src/robot.cpp:130-160).  Each tip is described by the joints on ITS path, like a
    serial chain of its own; a joint shared by several tips (a torso) appears in each path with the
    same variable index.  Goals / FK outputs hold n_tips poses per problem."""
    name: str
    tips: tuple  # of TipPath
    qmin: np.ndarray
    qmax: np.ndarray
    vmax: np.ndarray
    bounded: np.ndarray  # [dof] uint8
    mimic: tuple = ()  # of MimicJoint (their `tip` says which path)

    @property
    def dof(self) -> int:
        return int(self.qmin.shape[0])

    @property
    def n_tips(self) -> int:
        return len(self.tips)


def multi_chain(name, paths, qmin, qmax, vmax, bounded=None) -> MultiChain:
    """paths: [(variable, origins, axes, joint_type or None, tip_xyz_rpy), ...]"""
    d = len(qmin)
    tips = []
    for variable, origins, axes, jt, tip in paths:
        n = len(variable)
        tips.append(TipPath(
            variable=np.ascontiguousarray(variable, dtype=np.int32),
            origin_xyz_rpy=np.ascontiguousarray(origins, dtype=np.float64).reshape(n, 6),
            axis=np.ascontiguousarray(axes, dtype=np.float64).reshape(n, 3),
            joint_type=np.ascontiguousarray(jt if jt is not None else [REVOLUTE] * n, dtype=np.int32),
            tip_xyz_rpy=np.ascontiguousarray(tip, dtype=np.float64).reshape(6)))
    return MultiChain(
        name=name, tips=tuple(tips),
        qmin=np.ascontiguousarray(qmin, dtype=np.float64),
        qmax=np.ascontiguousarray(qmax, dtype=np.float64),
        vmax=np.ascontiguousarray(vmax, dtype=np.float64),
        bounded=np.ascontiguousarray(bounded if bounded is not None else [1] * d, dtype=np.uint8))


def side_by_side(name, chains, mounts) -> MultiChain:
    """Independent arms on one base (a dual-arm cell): arm k (a Chain) is mounted at the pure
    translation mounts[k] = (x, y, z); its variables follow those of arm k-1.  The first joint
    origin of every arm must be rotation-free, so that mount + origin is again a URDF origin."""
    paths, qmin, qmax, vmax, bounded = [], [], [], [], []
    off = 0
    for ch, m in zip(chains, mounts):
        o = ch.origin_xyz_rpy.copy()
        assert np.all(o[0, 3:] == 0.0), "first joint origin must be rotation-free"
        o[0, :3] = o[0, :3] + np.asarray(m, dtype=np.float64)[:3]
        paths.append((np.arange(off, off + ch.dof), o, ch.axis, ch.joint_type, ch.tip_xyz_rpy))
        qmin += list(ch.qmin)
        qmax += list(ch.qmax)
        vmax += list(ch.vmax)
        bounded += list(ch.bounded)
        off += ch.dof
    return multi_chain(name, paths, qmin, qmax, vmax, bounded)


def torso_dual_arm() -> MultiChain:
    """A 9-variable tree: a torso yaw joint shared by two 4-joint arms (left / right shoulder
    offsets), one tip per hand.  Synthetic geometry -- exercises a joint that moves several tips."""
    torso = ([0, 0, 0.4, 0, 0, 0], [0, 0, 1])

    def arm(side):
        y = 0.2 * side
        origins = [torso[0], [0.0, y, 0.3, PI / 2 * side, 0, 0], [0, 0, 0, 0, -PI / 2, 0],
                   [0.3, 0, 0, 0, 0, 0], [0.25, 0, 0, 0, PI / 2, 0]]
        axes = [torso[1], [0, 0, 1], [0, 1, 0], [0, 1, 0], [0, 0, 1]]
        return origins, axes

    lo, la = arm(+1.0)
    ro, ra = arm(-1.0)
    paths = [([0, 1, 2, 3, 4], lo, la, None, [0, 0, 0.1, 0, 0, 0]),
             ([0, 5, 6, 7, 8], ro, ra, None, [0, 0, 0.1, 0, 0, PI / 3])]
    qmin = [-1.5] + [-2.5, -2.0, -2.3, -2.8] * 2
    qmax = [1.5] + [2.5, 2.0, 2.3, 2.8] * 2
    vmax = [1.0] + [2.0, 2.0, 2.5, 3.0] * 2
    return multi_chain("torso_dual_arm", paths, qmin, qmax, vmax)


def _chain(name, origins, axes, tip, qmin, qmax, vmax, bounded=None, joint_type=None) -> Chain:
    d = len(origins)
    return Chain(
        name=name,
        origin_xyz_rpy=np.ascontiguousarray(origins, dtype=np.float64).reshape(d, 6),
        axis=np.ascontiguousarray(axes, dtype=np.float64).reshape(d, 3),
        joint_type=np.ascontiguousarray(
            joint_type if joint_type is not None else [REVOLUTE] * d, dtype=np.int32),
        tip_xyz_rpy=np.ascontiguousarray(tip, dtype=np.float64).reshape(6),
        qmin=np.ascontiguousarray(qmin, dtype=np.float64),
        qmax=np.ascontiguousarray(qmax, dtype=np.float64),
        vmax=np.ascontiguousarray(vmax, dtype=np.float64),
        bounded=np.ascontiguousarray(
            bounded if bounded is not None else [1] * d, dtype=np.uint8),
    )


def panda() -> Chain:
    """Franka Emika Panda, panda_link0 -> panda_hand (group "panda_arm", tip "panda_hand")."""
    origins = [
        [0, 0, 0.333, 0, 0, 0],
        [0, 0, 0, -PI / 2, 0, 0],
        [0, -0.316, 0, PI / 2, 0, 0],
        [0.0825, 0, 0, PI / 2, 0, 0],
        [-0.0825, 0.384, 0, -PI / 2, 0, 0],
        [0, 0, 0, PI / 2, 0, 0],
        [0.088, 0, 0, PI / 2, 0, 0],
    ]
    axes = [[0, 0, 1]] * 7
    # fixed joint8 (0 0 0.107 | 0 0 0) then fixed hand joint (0 0 0 | 0 0 -pi/4)
    tip = [0, 0, 0.107, 0, 0, -PI / 4]
    qmin = [-2.8973, -1.7628, -2.8973, -3.0718, -2.8973, -0.0175, -2.8973]
    qmax = [2.8973, 1.7628, 2.8973, -0.0698, 2.8973, 3.7525, 2.8973]
    vmax = [2.175, 2.175, 2.175, 2.175, 2.61, 2.61, 2.61]
    return _chain("panda", origins, axes, tip, qmin, qmax, vmax)


#: the "ready"/home pose the reference tests use (tests/ik_tests.cpp:247-248)
PANDA_HOME = np.array([0.0, -PI / 4, 0.0, -3.0 * PI / 4, 0.0, PI / 2, PI / 4])


def ur5() -> Chain:
    """Universal Robots UR5 (classic universal_robot URDF), base_link -> ee_link."""
    origins = [
        [0, 0, 0.089159, 0, 0, 0],
        [0, 0.13585, 0, 0, PI / 2, 0],
        [0, -0.1197, 0.425, 0, 0, 0],
        [0, 0, 0.39225, 0, PI / 2, 0],
        [0, 0.093, 0, 0, 0, 0],
        [0, 0, 0.09465, 0, 0, 0],
    ]
    axes = [[0, 0, 1], [0, 1, 0], [0, 1, 0], [0, 1, 0], [0, 0, 1], [0, 1, 0]]
    tip = [0, 0.0823, 0, 0, 0, PI / 2]
    lim = 2 * PI
    qmin = [-lim, -lim, -PI, -lim, -lim, -lim]
    qmax = [lim, lim, PI, lim, lim, lim]
    vmax = [3.15, 3.15, 3.15, 3.2, 3.2, 3.2]
    return _chain("ur5", origins, axes, tip, qmin, qmax, vmax)


#: a mid-range UR5 seed ("a fixed mid-range pose")
UR5_HOME = np.array([0.0, -PI / 2, PI / 2, -PI / 2, -PI / 2, 0.0])


def rr(l1: float = 2.0, l2: float = 1.0) -> Chain:
    """Planar 2-link RR arm of the reference tests (tests/ik_tests.cpp:15-48: links 2 and 1;
    tests/robot_tests.cpp:9-38: links 1 and 1).  RobotModelBuilder joints have no position
    limits that bind in the tests; +-pi bounds with velocity 1 are used here."""
    origins = [[0, 0, 0, 0, 0, 0], [l1, 0, 0, 0, 0, 0]]
    axes = [[0, 0, 1], [0, 0, 1]]
    tip = [l2, 0, 0, 0, 0, 0]
    return _chain("rr", origins, axes, tip, [-PI, -PI], [PI, PI], [1.0, 1.0])


def on_floating_base(ch: Chain, origin=(0, 0, 0, 0, 0, 0), reach: float = 0.5) -> Chain:
    """`ch` mounted on a free-flying base: a floating joint (seven variables) in front of the chain's own
    joints.  MoveIt's FloatingJointModel bounds: translations within the virtual joint's box (here
    +-reach, bounded), quaternion components in [-1, 1]."""
    d = ch.dof
    origins = np.concatenate([np.array([origin] + [[0.0] * 6] * 6, dtype=np.float64), ch.origin_xyz_rpy])
    axes = np.concatenate([np.tile([0.0, 0.0, 1.0], (7, 1)), ch.axis])
    jt = np.concatenate([np.array(FLOATING, dtype=np.int32), ch.joint_type])
    qmin = np.concatenate([[-reach] * 3 + [-1.0] * 4, ch.qmin])
    qmax = np.concatenate([[reach] * 3 + [1.0] * 4, ch.qmax])
    vmax = np.concatenate([[1.0] * 7, ch.vmax])
    bounded = np.concatenate([np.ones(7, np.uint8), ch.bounded])
    assert d + 7 <= 16
    return _chain(ch.name + "_floating", origins, axes, ch.tip_xyz_rpy, qmin, qmax, vmax, bounded=bounded,
                  joint_type=jt)


def floating_panda() -> Chain:
    """the Panda on a free-flying base: 14 variables"""
    return on_floating_base(panda(), origin=(0.1, -0.2, 0.3, 0.2, -0.1, 0.4))


#: identity base pose + the ready pose of the arm
FLOATING_PANDA_HOME = np.concatenate([[0, 0, 0, 0, 0, 0, 1.0], PANDA_HOME])


def panda_on_torso() -> Chain:
    """The Panda on a two-joint torso (yaw about z at 0.4 m, pitch about y 0.25 m above it): nine revolute variables,
    every axis exactly +y or +z -- a chain of class 2 that is LONGER than the eight variables the exact flavour's
    specialised forms are instantiated for.  Synthetic geometry (a mobile manipulator's torso)."""
    a = panda()
    origins = np.concatenate([np.array([[0, 0, 0.4, 0, 0, 0], [0, 0, 0.25, 0, 0, 0]], dtype=np.float64), a.origin_xyz_rpy])
    origins[2] = [0.1, 0, 0.2, 0, 0, 0]  # the arm's base on the torso's upper link
    axes = np.concatenate([np.array([[0, 0, 1], [0, 1, 0]], dtype=np.float64), a.axis])
    qmin = np.concatenate([[-2.0, -0.6], a.qmin])
    qmax = np.concatenate([[2.0, 0.9], a.qmax])
    vmax = np.concatenate([[1.0, 1.0], a.vmax])
    return _chain("panda_on_torso", origins, axes, a.tip_xyz_rpy, qmin, qmax, vmax)


#: torso straight, the arm's ready pose
PANDA_ON_TORSO_HOME = np.concatenate([[0.0, 0.0], PANDA_HOME])


def dual_ur5() -> MultiChain:
    """Two UR5 arms 0.9 m apart on one base: 12 variables, two tips."""
    return side_by_side("dual_ur5", [ur5(), ur5()], [(0, 0.45, 0), (0, -0.45, 0)])


def by_name(name: str):
    return {"panda": panda, "ur5": ur5, "rr": rr, "dual_ur5": dual_ur5,
            "torso_dual_arm": torso_dual_arm, "floating_panda": floating_panda,
            "panda_on_torso": panda_on_torso}[name]()
