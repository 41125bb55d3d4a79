#!/usr/bin/env python3
"""An INDEPENDENT pin for diffco_amd's URDF reader.

tools/make_golden_urdf.py feeds the reference's FK with the joint table that diffco_amd.urdf.parse_urdf produced, so a
mis-read <origin> / <axis> / <mimic> would agree on both sides.  This script takes another route to the same numbers:
it reads the reference's URDF files with xml.etree directly — no diffco_amd, no reference Python — and composes the
link frames by the URDF definition in float64,

    T_child = T_parent . Trans(xyz) . Rz(yaw) Ry(pitch) Rx(roll) . M(q),   M = Rot(axis, q) | Trans(axis * q) | I,

with Rodrigues' formula for an arbitrary axis (the product code conjugates x / y axes onto z by signed permutations;
nothing of that is shared), mimic joints as multiplier * q_master + offset, and joint values addressed BY NAME.  It
stores every link's origin at q = 0 and at one seeded q inside the limits.  tests/test_urdf_tree.py maps the named
values onto the reader's dof order and compares the feature links' origins.

Usage (build container only, needs /root/reference):  python tools/make_golden_urdf_pin.py [--out tests/golden]
"""
import argparse
import json
import math
import os
import re
import xml.etree.ElementTree as ET

import numpy as np

REF_URDF = "/root/reference/examples/data/urdf" if os.path.isdir("/root/reference/examples/data/urdf") else None
ROBOTS = {
    "urdf_panda": "panda_description/urdf/panda.urdf",
    "urdf_panda_nogripper": "panda_description/urdf/panda_no_gripper.urdf",
    "urdf_fetch_arm": "fetch_description/urdf/fetch_arm_no_gripper.urdf",
    "urdf_iiwa7": "kuka_iiwa/urdf/iiwa7.urdf",
    "urdf_allegro": "allegro/urdf/allegro_hand_description_left.urdf",
    "urdf_trifinger": "trifinger_edu_description/trifinger_edu.urdf",
    "urdf_jaco": "kinova_description/urdf/jaco_clean.urdf",
    "urdf_2link": "2link_robot.urdf",
    "urdf_fetch": "fetch_description/urdf/fetch.urdf",
    "urdf_iiwa7_allegro": "kuka_iiwa/urdf/iiwa7_allegro.urdf",
}


def find_root():
    for base, _, files in os.walk("/root/reference"):
        if "2link_robot.urdf" in files:
            return base
    raise SystemExit("reference URDF directory not found")


def vec(text, default):
    return np.array([float(t) for t in text.split()], dtype=np.float64) if text else np.array(default, dtype=np.float64)


def rpy_matrix(r, p, y):
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def rodrigues(axis, angle):
    n = axis / np.linalg.norm(axis)
    K = np.array([[0, -n[2], n[1]], [n[2], 0, -n[0]], [-n[1], n[0], 0]])
    return np.eye(3) + math.sin(angle) * K + (1 - math.cos(angle)) * (K @ K)


def homog(R, t):
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, t
    return T


def read(path):
    text = open(path, encoding="utf-8", errors="replace").read()
    text = re.sub(r"<(/?)(\w+):", r"<\1\2_", text)          # simulator blocks with undeclared prefixes
    text = re.sub(r"\s(\w+):(\w+)=", r" \1_\2=", text)
    root = ET.fromstring(text)
    joints = []
    for j in root.findall("joint"):
        if j.find("parent") is None or j.find("child") is None:
            continue
        o, ax, lim, mim = j.find("origin"), j.find("axis"), j.find("limit"), j.find("mimic")
        joints.append(dict(
            name=j.get("name"), type=j.get("type"), parent=j.find("parent").get("link"), child=j.find("child").get("link"),
            xyz=vec(o.get("xyz") if o is not None else None, [0, 0, 0]), rpy=vec(o.get("rpy") if o is not None else None, [0, 0, 0]),
            axis=vec(ax.get("xyz") if ax is not None else None, [1, 0, 0]),
            lower=float(lim.get("lower")) if lim is not None and lim.get("lower") is not None else None,
            upper=float(lim.get("upper")) if lim is not None and lim.get("upper") is not None else None,
            mimic=None if mim is None else (mim.get("joint"), float(mim.get("multiplier", 1.0)), float(mim.get("offset", 0.0)))))
    links = [ln.get("name") for ln in root.findall("link")]
    return links, joints


def link_origins(links, joints, q_by_name):
    children = set(j["child"] for j in joints)
    roots = [ln for ln in links if ln not in children]
    assert len(roots) == 1, roots
    by_parent = {}
    for j in joints:
        by_parent.setdefault(j["parent"], []).append(j)
    pos = {}

    def value(j):
        if j["mimic"] is not None:
            master, mul, off = j["mimic"]
            return mul * q_by_name[master] + off
        return q_by_name[j["name"]]

    def walk(link, T):
        pos[link] = T[:3, 3].copy()
        for j in by_parent.get(link, []):
            Tj = homog(rpy_matrix(*j["rpy"]), j["xyz"])
            if j["type"] in ("revolute", "continuous"):
                Tj = Tj @ homog(rodrigues(j["axis"], value(j)), np.zeros(3))
            elif j["type"] == "prismatic":
                Tj = Tj @ homog(np.eye(3), j["axis"] / np.linalg.norm(j["axis"]) * value(j))
            elif j["type"] != "fixed":
                raise SystemExit(f"joint type {j['type']} not handled")
            walk(j["child"], T @ Tj)

    walk(roots[0], np.eye(4))
    return pos


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"))
    args = ap.parse_args()
    base = find_root()
    rng = np.random.default_rng(20260929)
    out = {}
    for name, rel in ROBOTS.items():
        links, joints = read(os.path.join(base, rel))
        movable = [j for j in joints if j["type"] != "fixed" and j["mimic"] is None]
        q = {}
        for j in movable:
            lo = j["lower"] if j["lower"] is not None and j["type"] != "continuous" else -math.pi
            hi = j["upper"] if j["upper"] is not None and j["type"] != "continuous" else math.pi
            if j["type"] != "continuous" and lo == hi == 0.0:  # a limit element without a range
                lo, hi = -math.pi, math.pi
            q[j["name"]] = float(lo + (hi - lo) * rng.uniform(0.15, 0.85))
        zero = {k: 0.0 for k in q}
        p0, p1 = link_origins(links, joints, zero), link_origins(links, joints, q)
        out[name] = dict(source=rel, q=q, links=links, at_zero={k: p0[k].tolist() for k in links},
                         at_q={k: p1[k].tolist() for k in links})
        print(f"{name}: {len(links)} links, {len(movable)} named joint values")
    path = os.path.join(os.path.abspath(args.out), "urdf_pin.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
