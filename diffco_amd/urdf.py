"""URDF kinematic trees as a fused FK feed: parse a URDF, restate the reference's link/joint
bookkeeping, and compile the tree into a DCX_FK_TREE description that `dcx_fkine` and the fused score kernel
execute on the GPU.

Host logic only (XML, graph walking, constant folding); no numerical FK happens here.  What is restated
(paths under /root/reference/diffco):

* `URDFRobot.__init__` collision_interfaces/urdf_interface.py:378-434 — one RigidBody per <link> in file order,
  a degree of freedom per non-fixed, non-mimic joint in LINK order, mimic joints follow their driver, joint
  limits (+-pi when a revolute/prismatic joint has none, +-2pi for continuous);
* `get_body_parameters_from_urdf` :565-620 — joint origin (xyz, rpy), axis, mimic multiplier/offset;
* `RigidBody.forward_kinematics` collision_interfaces/rigid_body.py:82-140 — joint pose = origin followed by the
  joint motion; a revolute axis is read as +-x, +-y, else +-z (:103-108), a prismatic axis as a vector (:114-118);
* `ForwardKinematicsDiffCo.__init__` collision_checkers.py:345-360 — the features are the origins of the links
  whose joint origin has a non-zero translation ("unique position links"), stacked as [B, 3, L] (:386-393).

The reference loads URDFs through yourdfpy (requirements.txt, unpinned), which is not available here; its
documented behaviour for the attributes used above is restated: missing <origin> = identity, missing <axis> =
(1, 0, 0), `rpy` = fixed-axis roll-pitch-yaw (R = Rz(yaw) Ry(pitch) Rx(roll)), <mimic multiplier=1 offset=0>.
"""
import math
import os
import re
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import torch

from . import _fkdesc as fd
from . import _ops


# ----------------------------------------------------------------------------- parsed model
@dataclass
class Joint:
    name: str
    type: str
    parent: str
    child: str
    xyz: np.ndarray
    rpy: np.ndarray
    axis: np.ndarray
    lower: Optional[float] = None
    upper: Optional[float] = None
    mimic_joint: Optional[str] = None
    mimic_multiplier: float = 1.0
    mimic_offset: float = 0.0


@dataclass
class Tree:
    links: List[str]
    joints: List[Joint]
    joint_of_child: dict = field(default_factory=dict)  # link name -> Joint whose child it is
    children: dict = field(default_factory=dict)        # link name -> [child link names], file order of joints


def _floats(text, n, what):
    vals = [float(v) for v in text.split()]
    if len(vals) != n:
        raise ValueError(f"URDF: {what} needs {n} numbers, got {text!r}")
    return np.array(vals, dtype=np.float64)


def parse_urdf(source) -> Tree:
    """`source`: a path or the XML text.  Only the kinematic content is read."""
    if isinstance(source, (bytes, str)) and not str(source).lstrip().startswith("<"):
        if not os.path.exists(source):
            raise FileNotFoundError(source)
        with open(source, "rb") as f:
            source = f.read()
    text = source.decode() if isinstance(source, bytes) else source
    try:
        root = ET.fromstring(text)
    except ET.ParseError as e:
        if "unbound prefix" not in str(e):
            raise
        # simulator blocks with undeclared namespace prefixes (<sensor:camera> inside <gazebo>, as in the reference's
        # fetch.urdf): the prefixes carry no kinematic content, so they are flattened instead of rejected
        text = re.sub(r"<(/?)([A-Za-z_][\w.-]*):", r"<\1\2_", text)
        text = re.sub(r"(\s)([A-Za-z_][\w.-]*):([A-Za-z_][\w.-]*)=", r"\1\2_\3=", text)
        root = ET.fromstring(text)
    if root.tag != "robot":
        raise ValueError("URDF: the root element must be <robot>")
    links = [ln.get("name") for ln in root.findall("link")]
    if len(set(links)) != len(links):
        raise ValueError("URDF: duplicate link names")
    joints = []
    for j in root.findall("joint"):
        origin = j.find("origin")
        xyz = _floats(origin.get("xyz", "0 0 0"), 3, "origin xyz") if origin is not None else np.zeros(3)
        rpy = _floats(origin.get("rpy", "0 0 0"), 3, "origin rpy") if origin is not None else np.zeros(3)
        ax = j.find("axis")
        axis = _floats(ax.get("xyz", "1 0 0"), 3, "axis xyz") if ax is not None else np.array([1.0, 0.0, 0.0])
        lim, mim = j.find("limit"), j.find("mimic")
        jt = Joint(name=j.get("name"), type=j.get("type"), parent=j.find("parent").get("link"),
                   child=j.find("child").get("link"), xyz=xyz, rpy=rpy, axis=axis)
        if lim is not None:
            jt.lower, jt.upper = float(lim.get("lower", 0.0)), float(lim.get("upper", 0.0))
        if mim is not None:
            jt.mimic_joint = mim.get("joint")
            jt.mimic_multiplier = float(mim.get("multiplier", 1.0))
            jt.mimic_offset = float(mim.get("offset", 0.0))
        joints.append(jt)
    tree = Tree(links=links, joints=joints)
    for jt in joints:
        if jt.child not in links or jt.parent not in links:
            raise ValueError(f"URDF: joint {jt.name} refers to an unknown link")
        if jt.child in tree.joint_of_child:
            raise ValueError(f"URDF: link {jt.child} has two parent joints")
        tree.joint_of_child[jt.child] = jt
        tree.children.setdefault(jt.parent, []).append(jt.child)
    return tree


def _origin_matrix(xyz, rpy):
    """4x4 [Rz(yaw) Ry(pitch) Rx(roll) | xyz] in float64 (rigid_body.py:96-99 composes the same product)"""
    r, p, y = (float(v) for v in rpy)
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    T = np.eye(4)
    T[:3, :3] = [[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                 [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                 [-sp, cp * sr, cp * cr]]
    T[:3, 3] = xyz
    return T


def _flat34(T):
    return [float(v) for v in np.asarray(T)[:3, :4].reshape(-1)]


# ----------------------------------------------------------------------------- compilation to DCX_FK_TREE
def compile_tree(tree: Tree, base_transform=None, coord_major=True):
    """-> (FkDesc, info) where info has dof, controlled joint names (dof order), limits and feature link names."""
    chains, points, info = plan_tree(tree, base_transform)
    return fd.tree_desc(info["dof"], chains, points, coord_major=coord_major), info


def compile_trees(trees, base_transforms=None, coord_major=True):
    """Several robots side by side (the reference's MultiURDFRobot, urdf_interface.py:700-741): the configuration is
    the concatenation of the robots' configurations (split_configs :741-742) and the features are the robots'
    feature links one robot after the other (tensorized_fkine_multi_robot, collision_checkers.py:374-384)."""
    base_transforms = base_transforms if base_transforms is not None else [None] * len(trees)
    chains, points, infos, q0 = [], [], [], 0
    for tree, base in zip(trees, base_transforms):
        ch, pt, info = plan_tree(tree, base)
        for c in ch:
            for jt in c["joints"]:
                if jt["type"] != fd.DCX_J_FIXED:
                    jt["q"] += q0
        points += [(c + len(chains), f, off) for c, f, off in pt]
        chains += ch
        infos.append(info)
        q0 += info["dof"]
    info = dict(dof=q0, joint_names=[(i, n) for i, inf in enumerate(infos) for n in inf["joint_names"]],
                joint_limits=np.concatenate([inf["joint_limits"] for inf in infos], axis=0),
                feature_links=[(i, n) for i, inf in enumerate(infos) for n in inf["feature_links"]],
                n_chains=len(chains), dofs=[inf["dof"] for inf in infos])
    return fd.tree_desc(q0, chains, points, coord_major=coord_major), info


def plan_tree(tree: Tree, base_transform=None):
    """-> (chains, points, info): the arguments of `_fkdesc.tree_desc` for one robot.

    Every leaf of the tree becomes one serial chain; fixed joints are folded into the next movable joint's
    constant transform (float64 products, rounded once when stored), and a feature link behind a fixed joint
    becomes a control-point offset in the frame of the last movable joint before it."""
    f32 = np.float32
    roots = [ln for ln in tree.links if ln not in tree.joint_of_child]
    if len(roots) != 1:
        raise ValueError(f"URDF: expected exactly one root link, found {roots}")
    # ---- degrees of freedom, in link order (urdf_interface.py:388-409)
    dof_of_joint, controlled = {}, []
    for ln in tree.links:
        jt = tree.joint_of_child.get(ln)
        if jt is None or jt.type == "fixed" or jt.mimic_joint is not None:
            continue
        if jt.type not in ("revolute", "continuous", "prismatic"):
            raise ValueError(f"URDF: joint {jt.name} has type {jt.type!r}; the reference's forward kinematics "
                             "(rigid_body.py:100-126) only moves revolute, continuous and prismatic joints")
        dof_of_joint[jt.name] = len(controlled)
        controlled.append(jt)
    by_name = {jt.name: jt for jt in tree.joints}
    for jt in tree.joints:
        if jt.mimic_joint is not None and jt.type != "fixed":
            if jt.mimic_joint not in dof_of_joint:
                raise ValueError(f"URDF: joint {jt.name} mimics {jt.mimic_joint!r}, which is not a controlled joint")
            if jt.type not in ("revolute", "continuous", "prismatic"):
                raise ValueError(f"URDF: mimic joint {jt.name} has unsupported type {jt.type!r}")
            dof_of_joint[jt.name] = dof_of_joint[by_name[jt.mimic_joint].name]
    limits = np.zeros((len(controlled), 2), dtype=np.float32)
    for i, jt in enumerate(controlled):
        if jt.type == "continuous":
            limits[i] = (-2 * math.pi, 2 * math.pi)
        elif jt.lower is None:
            limits[i] = (-math.pi, math.pi)
        else:
            limits[i] = (jt.lower, jt.upper)
    # ---- feature links: joint origin translation not all zero in fp32 (collision_checkers.py:358-360)
    feature_links = [ln for ln in tree.links
                     if ln in tree.joint_of_child and np.any(tree.joint_of_child[ln].xyz.astype(f32) != 0)]
    slot = {ln: k for k, ln in enumerate(feature_links)}
    # ---- root-to-leaf paths, pruned behind their last feature link, kept only if they add a feature
    paths = []

    def walk(link, path):
        path = path + [link]
        kids = tree.children.get(link, [])
        if not kids:
            paths.append(path)
        for kid in kids:
            walk(kid, path)

    walk(roots[0], [])
    chains, points, placed = [], [None] * len(feature_links), set()
    for path in paths:
        while path and (path[-1] not in slot or path[-1] in placed):
            path = path[:-1]
        if not path:
            continue
        c = len(chains)
        joints, pending = [], np.eye(4)
        for ln in path[1:]:
            jt = tree.joint_of_child[ln]
            origin = _origin_matrix(jt.xyz.astype(f32).astype(np.float64), jt.rpy.astype(f32).astype(np.float64))
            if jt.type == "fixed":
                pending = pending @ origin
                if ln in slot and ln not in placed:
                    if not joints:  # nothing moves before this link: a constant feature in the base frame
                        joints.append(dict(type=fd.DCX_J_FIXED, fixed=fd.IDENTITY_BASE))
                    points[slot[ln]] = (c, len(joints) - 1, tuple(pending[:3, 3]))
                    placed.add(ln)
                continue
            mult, off = (jt.mimic_multiplier, jt.mimic_offset) if jt.mimic_joint is not None else (1.0, 0.0)
            entry = dict(q=dof_of_joint[jt.name], fixed=_flat34(pending @ origin))
            ax = jt.axis.astype(f32)
            if jt.type == "prismatic":
                if not np.any(ax != 0):
                    raise ValueError(f"URDF: prismatic joint {jt.name} has a zero axis")  # rigid_body.py:116
                entry.update(type=fd.DCX_J_PRISMATIC, axis=tuple(float(v) for v in ax), scale=mult, offset=off)
            else:
                if abs(ax[0]) == 1:
                    kind, sgn = fd.DCX_J_REV_X, float(np.sign(ax[0]))
                elif abs(ax[1]) == 1:
                    kind, sgn = fd.DCX_J_REV_Y, float(np.sign(ax[1]))
                else:
                    kind, sgn = fd.DCX_J_REV_Z, float(np.sign(ax[2]))
                entry.update(type=kind, scale=sgn * mult, offset=sgn * off)
            joints.append(entry)
            pending = np.eye(4)
            if ln in slot and ln not in placed:
                points[slot[ln]] = (c, len(joints) - 1, (0.0, 0.0, 0.0))
                placed.add(ln)
        base = np.eye(4) if base_transform is None else np.asarray(base_transform, dtype=np.float64).reshape(4, 4)
        chains.append(dict(base=_flat34(base), joints=joints))
    assert all(p is not None for p in points)
    info = dict(dof=len(controlled), joint_names=[jt.name for jt in controlled],
                controlled_links=[jt.child for jt in controlled], joint_limits=limits,
                feature_links=feature_links, n_chains=len(chains))
    return chains, points, info


# ----------------------------------------------------------------------------- robot facade
class URDFRobotFK:
    """The kinematic half of the reference's `URDFRobot` + `ForwardKinematicsDiffCo.tensorized_fkine_single_robot`:

        robot = URDFRobotFK("panda.urdf")
        robot.fkine(q)            # [B, 3, L] link-origin features, HIP forward + vjp; pass `robot.fkine` as
                                  # DiffCo's `transform` and the tree is fused into the score kernel

    Ground-truth collision geometry (meshes, FCL) is outside diffco_amd's scope."""

    def __init__(self, urdf, name="", base_transform=None, coord_major=True):
        self.name = name
        self.tree = parse_urdf(urdf)
        self.base_transform = None if base_transform is None else \
            torch.as_tensor(base_transform, dtype=torch.float64).cpu().numpy().reshape(4, 4)
        self._desc, info = compile_tree(self.tree, self.base_transform, coord_major=coord_major)
        self._adopt(info)

    def _adopt(self, info):
        self._n_dofs = self.dof = info["dof"]
        self.joint_names = info["joint_names"]
        self.joint_limits = torch.from_numpy(info["joint_limits"].copy())
        self.limits = self.joint_limits  # the name diffco_amd.model robots use
        self.unique_position_link_names = info["feature_links"]
        self.n_chains = info["n_chains"]
        self.fkine_backup = None

    def fk_desc(self):
        return self._desc

    def fkine(self, q, reuse=False):
        if reuse:
            return self.fkine_backup
        unsqueezed = q.ndim == 1
        out = _ops.fkine(self._desc, torch.reshape(q, (-1, self.dof)))
        self.fkine_backup = out[0] if unsqueezed else out
        return self.fkine_backup

    tensorized_fkine = fkine

    def link_positions(self, q):
        """{link name: [B, 3]} for the feature links (the translation part of
        compute_forward_kinematics_all_links, urdf_interface.py:516-553)"""
        X = self.fkine(torch.reshape(q, (-1, self.dof)))
        cm = bool(self._desc.t_coord_major)
        return {ln: (X[:, :, k] if cm else X[:, k, :]) for k, ln in enumerate(self.unique_position_link_names)}

    def rand_configs(self, num_cfgs):
        lo, hi = self.joint_limits[:, 0], self.joint_limits[:, 1]
        return torch.rand(num_cfgs, self.dof) * (hi - lo) + lo

    def wrap(self, q):
        return q


class MultiURDFRobotFK(URDFRobotFK):
    """Several URDF robots as one transform — the kinematic half of the reference's `MultiURDFRobot`
    (urdf_interface.py:700-862) with `ForwardKinematicsDiffCo.tensorized_fkine_multi_robot`
    (collision_checkers.py:374-384): q = [q_robot0 | q_robot1 | ...], features = robot 0's links, then robot 1's.
    `unique_position_link_names` holds (robot_index, link_name) pairs like the reference's."""

    def __init__(self, urdf_robots, name=None, coord_major=True):
        names = [r.name for r in urdf_robots]
        if len(set(names)) != len(names):
            raise ValueError("Robot names must be unique")  # urdf_interface.py:721
        self.urdf_robots = list(urdf_robots)
        self.name = "_".join(names) if name is None else name
        self._desc, info = compile_trees([r.tree for r in urdf_robots], [r.base_transform for r in urdf_robots],
                                         coord_major=coord_major)
        self._adopt(info)
        self._dofs = info["dofs"]

    def split_configs(self, q):
        return torch.split(q, self._dofs, dim=1)
