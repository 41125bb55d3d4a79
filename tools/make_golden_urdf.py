#!/usr/bin/env python3
"""Golden vectors for the URDF kinematic-tree feed from the REFERENCE.

Runs ONLY in the build container (needs /root/reference).  The reference loads URDFs through yourdfpy +
trimesh + fcl, none of which exist here, so `URDFRobot.__init__` cannot run.  What DOES run, unmodified, is the
numerical part this row replaces:

  * `RigidBody` (collision_interfaces/rigid_body.py) — the per-joint pose and the recursive FK,
  * `URDFRobot.compute_forward_kinematics_all_links` (collision_interfaces/urdf_interface.py:516-553),
  * `ForwardKinematicsDiffCo.tensorized_fkine_single_robot` (collision_checkers.py:386-393) and the
    unique-position-link selection of its constructor (:355-360, re-run here line for line because the
    constructor itself needs the collision world).

The script fills a bare `URDFRobot` object with `RigidBody` nodes whose parameters follow
`get_body_parameters_from_urdf` (:565-620) — read from the reference's own URDF files by diffco_amd's parser —
and calls those reference methods on seeded configurations, in fp32 (the reference's precision) and in fp64
(referee).  Only inputs (the URDF-derived joint table, q) and outputs (features, J^T g) are stored; no reference
source.  The joint table is stored so tests do not need /root/reference.

Usage:  python tools/make_golden_urdf.py [--out tests/golden]
"""
import argparse
import importlib
import importlib.abc
import importlib.machinery
import json
import os
import sys
import types
from collections import defaultdict

import numpy as np
import torch

REF = "/root/reference"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

ROBOTS = {
    "urdf_panda": "panda_description/urdf/panda.urdf",                      # prismatic fingers, mimic, 3 leaves
    "urdf_panda_nogripper": "panda_description/urdf/panda_no_gripper.urdf",
    "urdf_fetch_arm": "fetch_description/urdf/fetch_arm_no_gripper.urdf",  # x / y / z axes, continuous joints
    "urdf_iiwa7": "kuka_iiwa/urdf/iiwa7.urdf",
    "urdf_allegro": "allegro/urdf/allegro_hand_description_left.urdf",     # 4 fingers, negative axes
    "urdf_trifinger": "trifinger_edu_description/trifinger_edu.urdf",      # -x / y axes, 3 fingers
    "urdf_jaco": "kinova_description/urdf/jaco_clean.urdf",
    "urdf_2link": "2link_robot.urdf",
}


LATER = {"urdf_fetch": "fetch_description/urdf/fetch.urdf",        # whole robot: 14 dof, 21 links, 9 leaves, gazebo blocks
         "urdf_iiwa7_allegro": "kuka_iiwa/urdf/iiwa7_allegro.urdf"}  # arm + hand: 23 dof, 28 links (D = 84)
LATER_SEEDS = {"urdf_fetch": 5151, "urdf_iiwa7_allegro": 6161}

ABSENT = ("fcl", "trimesh", "yourdfpy", "rospy", "curobo")


class _Anything(types.ModuleType):
    """stand-in for an absent third-party module (SURVEY.md §8c stub recipe): importable as a package, any
    attribute is another stand-in that can be called or subclassed.  Nothing numerical goes through it."""
    __path__ = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        mod = _Anything(self.__name__ + "." + name)
        sys.modules[mod.__name__] = mod
        setattr(self, name, mod)
        return mod

    def __call__(self, *a, **k):
        return None

    def __mro_entries__(self, bases):
        return (object,)


class _AbsentFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in ABSENT:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _Anything(spec.name)

    def exec_module(self, module):
        pass


def import_reference():
    sys.meta_path.insert(0, _AbsentFinder())
    pkg = types.ModuleType("diffco")
    pkg.__path__ = [f"{REF}/diffco"]
    sys.modules["diffco"] = pkg
    ci = types.ModuleType("diffco.collision_interfaces")
    ci.__path__ = [f"{REF}/diffco/collision_interfaces"]
    sys.modules["diffco.collision_interfaces"] = ci
    pkg.collision_interfaces = ci
    sva = importlib.import_module("diffco.collision_interfaces.spatial_vector_algebra")
    rb = importlib.import_module("diffco.collision_interfaces.rigid_body")
    ui = importlib.import_module("diffco.collision_interfaces.urdf_interface")
    for name in ("RobotInterfaceBase", "URDFRobot", "MultiURDFRobot", "robot_description_folder"):
        setattr(ci, name, getattr(ui, name))
    for name in ("ROSRobotEnv", "CuRoboRobot", "CuRoboCollisionWorldEnv", "ShapeEnv", "PCDEnv"):
        setattr(ci, name, type(name, (), {}))
    for m in ("model", "kernel", "kernel_perceptrons"):
        setattr(pkg, m, importlib.import_module("diffco." + m))
    cc = importlib.import_module("diffco.collision_checkers")
    return types.SimpleNamespace(sva=sva, rb=rb, ui=ui, cc=cc)


R = import_reference()
from diffco_amd import urdf as U  # noqa: E402  (host-side parser only; no GPU needed)


def reference_robot(tree, dtype, base=None, name="golden"):
    """a reference URDFRobot whose _bodies were filled the way URDFRobot.__init__ (:378-418) fills them"""
    rob = object.__new__(R.ui.URDFRobot)
    R.ui.RobotInterfaceBase.__init__(rob, name=name, device="cpu")
    base = torch.eye(4, dtype=dtype) if base is None else torch.as_tensor(base, dtype=torch.float32).to(dtype)
    rob.base_transform = R.sva.CoordinateTransform(base[:3, :3], base[:3, 3], device="cpu")  # :365-366
    rob._n_dofs, rob._controlled_joints, rob._mimic_joints, rob._bodies = 0, [], defaultdict(list), []
    rob._body_name_to_idx_map = {}
    joint_by_name = {j.name: j for j in tree.joints}
    for link_idx, link in enumerate(tree.links):
        jt = tree.joint_of_child.get(link)
        p = {"link_idx": link_idx, "link_name": link}
        if jt is None:  # :573-581
            p.update(joint_rot_angles=torch.zeros(3, dtype=dtype), joint_trans=torch.zeros(3, dtype=dtype),
                     joint_name="base_joint", joint_type="fixed", joint_limits=None,
                     joint_axis=torch.zeros((1, 3), dtype=dtype), joint_mimic=None)
        else:           # :582-615; values pass through float32 first, as the reference's tensors do
            mimic = None
            if jt.mimic_joint is not None:
                mimic = types.SimpleNamespace(joint=jt.mimic_joint, multiplier=jt.mimic_multiplier, offset=jt.mimic_offset)
            p.update(joint_rot_angles=torch.tensor(jt.rpy, dtype=torch.float32).to(dtype),
                     joint_trans=torch.tensor(jt.xyz, dtype=torch.float32).to(dtype),
                     joint_name=jt.name, joint_type=jt.type,
                     joint_limits=None if jt.lower is None else {"lower": jt.lower, "upper": jt.upper},
                     joint_axis=(torch.tensor(jt.axis, dtype=torch.float32).to(dtype).reshape(1, 3)
                                 if jt.type != "fixed" else torch.zeros((1, 3), dtype=dtype)),
                     joint_mimic=mimic)
        body = R.rb.RigidBody(rigid_body_params=p, device="cpu")
        body.dof_idx = None
        if body.joint_type != "fixed":  # :397-405
            if body.joint_mimic is None:
                body.dof_idx = rob._n_dofs
                rob._n_dofs += 1
                rob._controlled_joints.append(link_idx)
            else:
                rob._mimic_joints[joint_by_name[body.joint_mimic.joint].child].append(body.name)
        rob._bodies.append(body)
        rob._body_name_to_idx_map[body.name] = link_idx
    for body in rob._bodies:  # :412-418
        if body.joint_name == "base_joint":
            continue
        parent = rob._bodies[rob._body_name_to_idx_map[joint_by_name[body.joint_name].parent]]
        body.set_parent(parent)
        parent.add_child(body)
    return rob


def reference_checker(rob):
    """ForwardKinematicsDiffCo with only the state tensorized_fkine_single_robot reads"""
    chk = object.__new__(R.cc.ForwardKinematicsDiffCo)
    chk.robot = rob
    chk.unique_position_link_names = []
    for link_body in rob._bodies:  # collision_checkers.py:358-360
        if torch.any(link_body.joint_trans() != 0):
            chk.unique_position_link_names.append(link_body.name)
    return chk


def gen_multi(out, gen):
    """two Pandas facing each other with the base transforms of the reference's own URDF test
    (examples/tests/test_urdf_robot.py:59-74), through MultiURDFRobot.compute_forward_kinematics_all_links
    (urdf_interface.py:857-862) and ForwardKinematicsDiffCo.tensorized_fkine_multi_robot (collision_checkers.py:374-384)"""
    import math
    rel = ROBOTS["urdf_panda"]
    tree = U.parse_urdf(open(os.path.join(REF, "diffco", "robot_data", rel)).read())

    def base(x, pitch):  # tf.translation_matrix([x, 0, 0.8]) with tf.euler_matrix(0, pitch, 0) as rotation
        c, s = math.cos(pitch), math.sin(pitch)
        return np.array([[c, 0, s, x], [0, 1, 0, 0.0], [-s, 0, c, 0.8], [0, 0, 0, 1]], dtype=np.float64)

    bases = [base(0.1, math.pi / 2), base(-0.1, -math.pi / 2)]
    desc, info = U.compile_trees([tree, tree], bases)
    res = {}
    for dtype in (torch.float32, torch.float64):
        torch.set_default_dtype(dtype)
        multi = object.__new__(R.ui.MultiURDFRobot)
        multi.urdf_robots = [reference_robot(tree, dtype, b, name=f"panda{i + 1}") for i, b in enumerate(bases)]
        multi._bodies = [[body for body in robot._bodies] for robot in multi.urdf_robots]  # :739
        chk = object.__new__(R.cc.ForwardKinematicsDiffCo)
        chk.robot = multi
        chk.unique_position_link_names = []
        for robot_idx, link_body_list in enumerate(multi._bodies):  # collision_checkers.py:348-352
            for link_body in link_body_list:
                if torch.any(link_body.joint_trans() != 0):
                    chk.unique_position_link_names.append((robot_idx, link_body.name))
        assert chk.unique_position_link_names == info["feature_links"]
        if dtype == torch.float32:
            lim = torch.from_numpy(info["joint_limits"])
            q = torch.rand(48, info["dof"], generator=gen) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
            q[0] = 0.0
            gX = torch.randn(48, 3, len(info["feature_links"]), generator=gen)
        qd = q.detach().clone().to(dtype).requires_grad_(True)
        X = chk.tensorized_fkine_multi_robot(qd)
        (gq,) = torch.autograd.grad((X * gX.to(dtype)).sum(), qd)
        res[dtype] = (X.detach(), gq.detach())
    torch.set_default_dtype(torch.float32)
    np.savez_compressed(os.path.join(out, "fk_urdf_dual_panda.npz"), q=q.numpy(), x32=res[torch.float32][0].numpy(),
                        gq32=res[torch.float32][1].numpy(), x64=res[torch.float64][0].numpy(),
                        gq64=res[torch.float64][1].numpy(), gx=gX.numpy(), limits=info["joint_limits"],
                        bases=np.stack(bases))
    err = (res[torch.float32][0].double() - res[torch.float64][0]).abs().max().item()
    print(f"  wrote fk_urdf_dual_panda.npz  dof={info['dof']} L={len(info['feature_links'])} chains={info['n_chains']} "
          f"ref fp32-vs-fp64 {err:.2e}")
    return dict(source=[rel, rel], dof=info["dof"], features=len(info["feature_links"]), chains=info["n_chains"],
                ref_fp32_vs_fp64=err, stack="reference (multi-robot)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    args = ap.parse_args()
    out = os.path.abspath(args.out)
    gen = torch.Generator().manual_seed(4242)
    manifest = {}
    gen_main_after_robots = None
    # (robots added later draw from their own generator so that the fixtures above keep their bytes)
    for name, rel in list(ROBOTS.items()) + list(LATER.items()):
        if name in LATER:
            if gen_main_after_robots is None:
                gen_main_after_robots = gen  # the state gen_multi continues from (keeps fk_urdf_dual_panda.npz unchanged)
            gen = torch.Generator().manual_seed(LATER_SEEDS[name])
        text = open(os.path.join(REF, "diffco", "robot_data", rel)).read()
        tree = U.parse_urdf(text)
        desc, info = U.compile_tree(tree)
        res = {}
        for dtype in (torch.float32, torch.float64):
            # x_rot / y_rot allocate with the default dtype (spatial_vector_algebra.py:19, 33): the fp64 referee
            # pass runs the same reference code under a float64 default
            torch.set_default_dtype(dtype)
            rob = reference_robot(tree, dtype)
            chk = reference_checker(rob)
            assert rob._n_dofs == info["dof"], (name, rob._n_dofs, info["dof"])
            assert chk.unique_position_link_names == info["feature_links"], name
            if dtype == torch.float32:
                lim = torch.from_numpy(info["joint_limits"])
                q = torch.rand(48, rob._n_dofs, generator=gen) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
                q[0] = 0.0
                gX = torch.randn(48, 3, len(info["feature_links"]), generator=gen)
            qd = q.detach().clone().to(dtype).requires_grad_(True)
            try:
                X = chk.tensorized_fkine_single_robot(qd)
                stacked = "reference"
            except RuntimeError:
                # A feature link in front of every movable joint has a batch-1 pose and the reference's torch.stack
                # (collision_checkers.py:391) rejects the ragged list.  Its poses are still the reference's own
                # (compute_forward_kinematics_all_links); only the stack is redone here with the constant rows
                # broadcast over the batch, which is what diffco_amd does.
                fk_dict = chk.fkine(qd)
                X = torch.stack([pos.expand(len(qd), 3) for ln in chk.unique_position_link_names
                                 for pos, _ in fk_dict[ln]], dim=-1)
                stacked = "broadcast (reference stack raises on constant links)"
            assert X.dtype == dtype, (name, X.dtype, "reference FK did not stay in the requested dtype")
            (gq,) = torch.autograd.grad((X * gX.to(dtype)).sum(), qd)
            res[dtype] = (X.detach(), gq.detach())
        torch.set_default_dtype(torch.float32)
        # the URDF-derived joint table (inputs), so the tests can rebuild the description without the URDF file
        table = [dict(name=j.name, type=j.type, parent=j.parent, child=j.child, xyz=j.xyz.tolist(), rpy=j.rpy.tolist(),
                      axis=j.axis.tolist(), lower=j.lower, upper=j.upper, mimic_joint=j.mimic_joint,
                      mimic_multiplier=j.mimic_multiplier, mimic_offset=j.mimic_offset) for j in tree.joints]
        np.savez_compressed(
            os.path.join(out, f"fk_{name}.npz"), q=q.numpy(), x32=res[torch.float32][0].numpy(),
            gq32=res[torch.float32][1].numpy(), x64=res[torch.float64][0].numpy(), gq64=res[torch.float64][1].numpy(),
            gx=gX.numpy(), limits=info["joint_limits"],
            model=np.frombuffer(json.dumps(dict(links=tree.links, joints=table, feature_links=info["feature_links"],
                                                joint_names=info["joint_names"], source=rel)).encode(), dtype=np.uint8))
        err = (res[torch.float32][0].double() - res[torch.float64][0]).abs().max().item()
        manifest[name] = dict(source=rel, dof=info["dof"], features=len(info["feature_links"]), chains=info["n_chains"],
                              ref_fp32_vs_fp64=err, stack=stacked)
        print(f"  wrote fk_{name}.npz  dof={info['dof']} L={len(info['feature_links'])} chains={info['n_chains']} "
              f"ref fp32-vs-fp64 {err:.2e}")
    manifest["urdf_dual_panda"] = gen_multi(out, gen_main_after_robots)
    with open(os.path.join(out, "MANIFEST_urdf.json"), "w") as f:
        json.dump({"generator": "tools/make_golden_urdf.py", "torch": torch.__version__,
                   "reference": "ucsdarclab/diffco @ /root/reference (2025-03-21)", "robots": manifest}, f, indent=1)


if __name__ == "__main__":
    main()
