"""Serial chain extraction from a URDF robot description.

What pick_ik obtains from MoveIt's RobotModel -- `Robot::from`, `get_link_indices`,
`get_active_variable_indices` (reference src/robot.cpp:44-160) and the link transforms `make_fk_fn`
walks (src/fk_moveit.cpp:11-35) -- reduced to what the solver needs: the actuated joints on the path
base_link -> tip_link with their origins, axes and limits.  Fixed joints are folded into the next
joint's origin (or the tip transform); mimic joints are not variables in pick_ik (src/robot.cpp:144-150):
one with multiplier 0 is a constant joint and is folded at its offset, one that follows its master
cannot be expressed in a chain description and is refused when it lies on the path; `continuous` joints are unbounded variables (position_bounded_ = false).
"""
from __future__ import annotations

import math
import xml.etree.ElementTree as ET

import numpy as np

from .robots import (FLOATING, PLANAR_THETA, PLANAR_X, PLANAR_Y, PRISMATIC, REVOLUTE, Chain, MimicJoint, MultiChain,
                     multi_chain)


def _rpy_matrix(rpy):
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def _matrix_rpy(R):
    pitch = math.atan2(-R[2, 0], math.hypot(R[0, 0], R[1, 0]))
    if abs(abs(pitch) - math.pi / 2) < 1e-12:  # gimbal lock: put everything into roll
        return [math.atan2(-R[1, 2], R[1, 1]) if pitch < 0 else math.atan2(R[0, 1], R[1, 1]), pitch, 0.0]
    return [math.atan2(R[2, 1], R[2, 2]), pitch, math.atan2(R[1, 0], R[0, 0])]


def _iso(xyz, rpy):
    T = np.eye(4)
    T[:3, :3] = _rpy_matrix(rpy)
    T[:3, 3] = xyz
    return T


def _floats(text, n, default):
    if text is None:
        return list(default)
    v = [float(x) for x in text.split()]
    if len(v) != n:
        raise ValueError(f"expected {n} numbers, got {text!r}")
    return v


def _path_description(root, base_link, tip_link):
    """the actuated joints between base_link and tip_link: (names, origins, axes, types, limits, tip)"""
    links = {l.get("name") for l in root.findall("link")}
    for ln in (base_link, tip_link):
        if ln not in links:
            raise ValueError(f"link not found: {ln}")
    by_child = {}
    for j in root.findall("joint"):
        by_child[j.find("child").get("link")] = j
    path = []
    link = tip_link
    while link != base_link:
        j = by_child.get(link)
        if j is None:
            raise ValueError(f"{tip_link} is not a descendant of {base_link}")
        path.append(j)
        link = j.find("parent").get("link")
    path.reverse()

    names, origins, axes, types, qmin, qmax, vmax, bounded = [], [], [], [], [], [], [], []
    mimics = []  # joints that follow a variable of the path: (name, master name, after, origin6, axis, type, mult, off)
    pending = np.eye(4)
    for j in path:
        o = j.find("origin")
        xyz = _floats(o.get("xyz") if o is not None else None, 3, (0, 0, 0))
        rpy = _floats(o.get("rpy") if o is not None else None, 3, (0, 0, 0))
        pending = pending @ _iso(xyz, rpy)
        jt = j.get("type")
        if jt == "fixed":
            continue
        if not j.get("name"):
            raise ValueError("a joint on the path has no name attribute")
        mm = j.find("mimic")
        if mm is not None:
            # no variable (src/robot.cpp:144-150), but not fixed either: MoveIt sets it to multiplier *
            # master + offset.  A joint that follows another one is refused; multiplier 0 is a constant.
            mult, off = float(mm.get("multiplier", 1.0)), float(mm.get("offset", 0.0))
            if jt not in ("revolute", "continuous", "prismatic"):
                raise ValueError(f"joint {j.get('name')}: a mimic joint must be revolute or prismatic")
            a = j.find("axis")
            ax = np.array(_floats(a.get("xyz") if a is not None else None, 3, (1, 0, 0)), dtype=float)
            if mult != 0.0:
                # it follows its master: one more step of the chain product (robots.MimicJoint), its origin = what
                # has been folded since the previous moving joint; the next joint's origin starts behind it
                if mm.get("joint") is None:
                    raise ValueError(f"joint {j.get('name')}: <mimic> without a joint attribute")
                mimics.append((j.get("name"), mm.get("joint"), len(names) - 1,
                               list(pending[:3, 3]) + _matrix_rpy(pending[:3, :3]), list(ax),
                               PRISMATIC if jt == "prismatic" else REVOLUTE, mult, off))
                pending = np.eye(4)
                continue
            ax = ax / np.linalg.norm(ax)
            J = np.eye(4)
            if jt == "prismatic":
                J[:3, 3] = ax * off
            else:
                c, sn = math.cos(off), math.sin(off)
                x, y, z = ax
                t1 = 1.0 - c
                J[:3, :3] = [[t1 * x * x + c, t1 * x * y - z * sn, t1 * x * z + y * sn],
                             [t1 * x * y + z * sn, t1 * y * y + c, t1 * y * z - x * sn],
                             [t1 * x * z - y * sn, t1 * y * z + x * sn, t1 * z * z + c]]
            pending = pending @ J
            continue
        if jt == "planar":
            # moveit::core::PlanarJointModel: variables <joint>/x, /y, /theta, transform
            # Translation(x, y, 0) * AngleAxis(theta, UnitZ) in the joint frame (the URDF <axis> is not
            # used by MoveIt); x / y take the <limit> when there is one, theta is unbounded
            lim = j.find("limit")
            has = lim is not None and lim.get("lower") is not None and lim.get("upper") is not None
            v = float(lim.get("velocity", 0.0)) if lim is not None else 0.0
            for k, (suffix, t) in enumerate((("x", PLANAR_X), ("y", PLANAR_Y), ("theta", PLANAR_THETA))):
                names.append(f"{j.get('name')}/{suffix}")
                origins.append((list(pending[:3, 3]) + _matrix_rpy(pending[:3, :3])) if k == 0 else [0.0] * 6)
                axes.append([1.0 if k == 0 else 0.0, 1.0 if k == 1 else 0.0, 1.0 if k == 2 else 0.0])
                types.append(t)
                b = has and k < 2
                bounded.append(1 if b else 0)
                qmin.append(float(lim.get("lower")) if b else (-math.pi if k == 2 else 0.0))
                qmax.append(float(lim.get("upper")) if b else (math.pi if k == 2 else 0.0))
                vmax.append(v)
            pending = np.eye(4)
            continue
        if jt == "floating":
            # moveit::core::FloatingJointModel: variables <joint>/trans_x .. rot_w, transform
            # Translation(t) * Quaterniond(w, x, y, z); translations unbounded, quaternion components in [-1, 1]
            for k, suffix in enumerate(("trans_x", "trans_y", "trans_z", "rot_x", "rot_y", "rot_z", "rot_w")):
                names.append(f"{j.get('name')}/{suffix}")
                origins.append((list(pending[:3, 3]) + _matrix_rpy(pending[:3, :3])) if k == 0 else [0.0] * 6)
                axes.append([0.0, 0.0, 1.0])
                types.append(FLOATING[k])
                bounded.append(1 if k >= 3 else 0)
                qmin.append(-1.0 if k >= 3 else 0.0)
                qmax.append(1.0 if k >= 3 else 0.0)
                vmax.append(0.0)
            pending = np.eye(4)
            continue
        if jt not in ("revolute", "continuous", "prismatic"):
            raise ValueError(f"joint {j.get('name')}: type {jt} is not supported")
        a = j.find("axis")
        axis = _floats(a.get("xyz") if a is not None else None, 3, (1, 0, 0))
        lim = j.find("limit")
        names.append(j.get("name"))
        origins.append(list(pending[:3, 3]) + _matrix_rpy(pending[:3, :3]))
        axes.append(axis)
        types.append(PRISMATIC if jt == "prismatic" else REVOLUTE)
        # urdfdom / MoveIt: <limit lower= upper=> are optional and default to 0; a revolute or
        # prismatic joint is position-bounded whenever it has a <limit> element
        is_bounded = jt != "continuous" and lim is not None
        bounded.append(1 if is_bounded else 0)
        qmin.append(float(lim.get("lower", 0.0)) if lim is not None and is_bounded else 0.0)
        qmax.append(float(lim.get("upper", 0.0)) if lim is not None and is_bounded else 0.0)
        vmax.append(float(lim.get("velocity", 0.0)) if lim is not None else 0.0)
        pending = np.eye(4)
    tip = list(pending[:3, 3]) + _matrix_rpy(pending[:3, :3])
    out_m = []
    for name, master, after, o6, ax, t, mult, off in mimics:
        if master not in names:
            raise ValueError(f"joint {name} mimics {master}, which is not a variable of the path to {tip_link}")
        out_m.append((after, names.index(master), o6, ax, t, mult, off))
    return names, origins, axes, types, (qmin, qmax, vmax, bounded), tip, out_m


def _root(urdf: str):
    text = urdf if urdf.lstrip().startswith("<") else open(urdf).read()
    return ET.fromstring(text)


def chain_from_urdf(urdf: str, base_link: str, tip_link: str, name: str | None = None) -> Chain:
    """`urdf` is the XML text or a path to a file.  Raises ValueError for unknown links (the
    reference throws std::invalid_argument for an unknown tip, src/pick_ik_plugin.cpp:65-67) or
    when tip_link is not a descendant of base_link."""
    root = _root(urdf)
    _, origins, axes, types, (qmin, qmax, vmax, bounded), tip, mim = _path_description(root, base_link, tip_link)
    if not origins:
        raise ValueError("no actuated joint between base_link and tip_link")
    d = len(origins)
    return Chain(name=name or root.get("name", "urdf"),
                 origin_xyz_rpy=np.array(origins, dtype=np.float64).reshape(d, 6),
                 axis=np.array(axes, dtype=np.float64).reshape(d, 3),
                 joint_type=np.array(types, dtype=np.int32),
                 tip_xyz_rpy=np.array(tip, dtype=np.float64),
                 qmin=np.array(qmin), qmax=np.array(qmax), vmax=np.array(vmax),
                 bounded=np.array(bounded, dtype=np.uint8),
                 mimic=tuple(MimicJoint(after_variable=a, master_variable=m, origin_xyz_rpy=tuple(o6), axis=tuple(ax),
                                        multiplier=mult, offset=off, joint_type=t) for a, m, o6, ax, t, mult, off in mim))


def multi_chain_from_urdf(urdf: str, base_link: str, tip_links, name: str | None = None) -> MultiChain:
    """Several tip links (the plugin's tip_frames): the active variables are the joints on the way
    to ANY tip (get_active_variable_indices, reference src/robot.cpp:130-160), numbered in the order
    they are first met walking tip_links[0]'s path, then the new joints of tip_links[1]'s path, ...
    so that a path's variable indices increase along it.  Returns (MultiChain, variable names) --
    the names give the order of the joint vector."""
    root = _root(urdf)
    index, limits, paths, all_mimic = {}, [], [], []
    for k_tip, tip_link in enumerate(tip_links):
        names, origins, axes, types, (qmin, qmax, vmax, bounded), tip, mim = _path_description(root, base_link, tip_link)
        var = []
        for i, n in enumerate(names):
            if n not in index:
                index[n] = len(index)
                limits.append((qmin[i], qmax[i], vmax[i], bounded[i]))
            var.append(index[n])
        if any(b <= a for a, b in zip(var, var[1:])):
            raise ValueError("tip_links order makes a path's variables non-increasing; list the tips "
                             "so that shared joints are met first")
        paths.append((var, np.array(origins, dtype=np.float64).reshape(len(var), 6),
                      np.array(axes, dtype=np.float64).reshape(len(var), 3), types, tip))
        all_mimic += [MimicJoint(after_variable=(-1 if a < 0 else var[a]), master_variable=var[m], origin_xyz_rpy=tuple(o6),
                                 axis=tuple(ax), multiplier=mult, offset=off, joint_type=t, tip=k_tip)
                      for a, m, o6, ax, t, mult, off in mim]
    if not index:
        raise ValueError("no actuated joint between base_link and the tips")
    lim = np.array(limits, dtype=np.float64)
    mc = multi_chain(name or root.get("name", "urdf"), paths, lim[:, 0], lim[:, 1], lim[:, 2],
                     lim[:, 3].astype(np.uint8))
    if all_mimic:
        import dataclasses
        mc = dataclasses.replace(mc, mimic=tuple(all_mimic))
    return mc, list(index)


def chain_to_urdf(chain: Chain, base_link: str = "base", tip_link: str = "tip") -> str:
    """Writes a chain back as URDF text (round-trip tests, hand-off to MoveIt setups)."""
    def f(x):
        return repr(float(x))

    out = [f'<robot name="{chain.name}">', f'  <link name="{base_link}"/>']
    parent = base_link
    for j in range(chain.dof):
        child = f"link{j + 1}"
        o = chain.origin_xyz_rpy[j]
        jt = "prismatic" if chain.joint_type[j] == PRISMATIC else ("revolute" if chain.bounded[j] else "continuous")
        out += [f'  <link name="{child}"/>',
                f'  <joint name="joint{j + 1}" type="{jt}">',
                f'    <parent link="{parent}"/><child link="{child}"/>',
                f'    <origin xyz="{f(o[0])} {f(o[1])} {f(o[2])}" rpy="{f(o[3])} {f(o[4])} {f(o[5])}"/>',
                f'    <axis xyz="{f(chain.axis[j][0])} {f(chain.axis[j][1])} {f(chain.axis[j][2])}"/>',
                f'    <limit lower="{f(chain.qmin[j])}" upper="{f(chain.qmax[j])}" velocity="{f(chain.vmax[j])}" effort="1"/>'
                if chain.bounded[j] else f'    <limit velocity="{f(chain.vmax[j])}" effort="1"/>',
                '  </joint>']
        parent = child
    t = chain.tip_xyz_rpy
    out += [f'  <link name="{tip_link}"/>', f'  <joint name="tip_fixed" type="fixed">',
            f'    <parent link="{parent}"/><child link="{tip_link}"/>',
            f'    <origin xyz="{f(t[0])} {f(t[1])} {f(t[2])}" rpy="{f(t[3])} {f(t[4])} {f(t[5])}"/>', '  </joint>', '</robot>']
    return "\n".join(str(x) for x in out)
