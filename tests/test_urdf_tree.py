"""URDF kinematic-tree feed without a GPU: the C oracle's DCX_FK_TREE restatement and diffco_amd's
URDF host logic (parser, dof / feature bookkeeping, tree flattening) pinned against the golden vectors that
tools/make_golden_urdf.py produced by running the reference's RigidBody / URDFRobot / ForwardKinematicsDiffCo code
on the reference's own URDF files."""
import numpy as np
import pytest

from helpers import URDF_NAMES, dual_panda_robot, load, relerr, urdf_model, urdf_robot, urdf_xml
from oracle import oracle

TOL64, TOL32 = 5e-8, 1e-6  # fp64: only the fp32 rounding of folded fixed joints separates the two; fp32: a few ulp


@pytest.mark.parametrize("name", URDF_NAMES)
def test_tree_oracle_matches_reference(name):
    d, rob = load("fk_" + name), urdf_robot(name)
    desc = rob.fk_desc()
    x64 = oracle.fkine(desc, d["q"], np.float64)
    assert x64.shape == d["x64"].shape  # [B, 3, L]: collision_checkers.py:390 stacks on the last axis
    assert relerr(x64, d["x64"]) < TOL64
    assert relerr(oracle.fkine_vjp(desc, d["q"], d["gx"], np.float64), d["gq64"]) < TOL64
    x32 = oracle.fkine(desc, d["q"], np.float32)
    assert relerr(x32, d["x64"]) < TOL32 and relerr(x32, d["x32"]) < TOL32
    g32 = oracle.fkine_vjp(desc, d["q"], d["gx"], np.float32)
    assert relerr(g32, d["gq64"]) < TOL32 and relerr(g32, d["gq32"]) < TOL32


@pytest.mark.parametrize("name", URDF_NAMES)
def test_urdf_bookkeeping_matches_reference(name):
    d, m, rob = load("fk_" + name), urdf_model(name), urdf_robot(name)
    assert rob.unique_position_link_names == m["feature_links"]  # collision_checkers.py:358-360
    assert rob.joint_names == m["joint_names"]                   # dof order = link order, urdf_interface.py:388-409
    assert np.array_equal(rob.joint_limits.numpy(), d["limits"])
    assert rob.dof == d["q"].shape[1] and rob.fk_desc().n_points == d["x64"].shape[2]
    q = rob.rand_configs(50)
    assert q.shape == (50, rob.dof)
    assert bool(((q >= rob.joint_limits[:, 0]) & (q <= rob.joint_limits[:, 1])).all())


def test_multi_robot_matches_reference():
    """MultiURDFRobot + tensorized_fkine_multi_robot (urdf_interface.py:857-862, collision_checkers.py:374-384)"""
    d, rob = load("fk_urdf_dual_panda"), dual_panda_robot()
    assert rob.dof == 16 and [i for i, _ in rob.unique_position_link_names] == [0] * 9 + [1] * 9
    assert np.array_equal(rob.joint_limits.numpy(), d["limits"])
    assert [t.shape[1] for t in rob.split_configs(rob.rand_configs(3))] == [8, 8]
    desc = rob.fk_desc()
    assert relerr(oracle.fkine(desc, d["q"], np.float64), d["x64"]) < TOL64
    assert relerr(oracle.fkine_vjp(desc, d["q"], d["gx"], np.float64), d["gq64"]) < TOL64
    x32 = oracle.fkine(desc, d["q"], np.float32)
    assert relerr(x32, d["x64"]) < TOL32 and relerr(x32, d["x32"]) < TOL32
    from diffco_amd.urdf import MultiURDFRobotFK
    with pytest.raises(ValueError, match="unique"):
        MultiURDFRobotFK([urdf_robot("urdf_2link"), urdf_robot("urdf_2link")])


def test_point_major_layout_is_a_transpose():
    d = load("fk_urdf_panda")
    cm, pm = urdf_robot("urdf_panda"), urdf_robot("urdf_panda", coord_major=False)
    a = oracle.fkine(cm.fk_desc(), d["q"], np.float64)          # [B, 3, L]
    b = oracle.fkine(pm.fk_desc(), d["q"], np.float64)          # [B, L, 3]
    assert np.array_equal(a.transpose(0, 2, 1), b)
    ga = oracle.fkine_vjp(cm.fk_desc(), d["q"], d["gx"], np.float64)
    gb = oracle.fkine_vjp(pm.fk_desc(), d["q"], d["gx"].transpose(0, 2, 1), np.float64)
    assert np.array_equal(ga, gb)


def test_base_transform_moves_every_feature():
    d = load("fk_urdf_iiwa7")
    base = np.eye(4)
    base[:3, :3] = [[0, -1, 0], [1, 0, 0], [0, 0, 1]]
    base[:3, 3] = [0.5, -0.25, 1.0]
    a = oracle.fkine(urdf_robot("urdf_iiwa7").fk_desc(), d["q"], np.float64)
    b = oracle.fkine(urdf_robot("urdf_iiwa7", base_transform=base).fk_desc(), d["q"], np.float64)
    want = np.einsum("rc,bcl->brl", base[:3, :3], a) + base[:3, 3][None, :, None]  # urdf_interface.py:541-544
    assert relerr(b, want) < 1e-12


def test_mimic_and_prismatic_joints():
    """panda.urdf: finger 2 mimics finger 1 (prismatic, axes +y / -y): one dof moves both fingertips apart"""
    rob = urdf_robot("urdf_panda")
    assert rob.dof == 8 and rob.joint_names[-1] == "panda_finger_joint1"
    q = np.zeros((2, 8))
    q[1, 7] = 0.03
    X = oracle.fkine(rob.fk_desc(), q, np.float64)
    li, ri = (rob.unique_position_link_names.index(n) for n in ("panda_leftfinger", "panda_rightfinger"))
    gap0 = np.linalg.norm(X[0, :, li] - X[0, :, ri])
    gap1 = np.linalg.norm(X[1, :, li] - X[1, :, ri])
    assert abs((gap1 - gap0) - 0.06) < 1e-6
    others = [k for k in range(X.shape[2]) if k not in (li, ri)]
    assert np.array_equal(X[0][:, others], X[1][:, others])


def test_constant_feature_links_are_broadcast():
    """trifinger: feature links in front of every movable joint (the reference's torch.stack raises on them)"""
    d, rob = load("fk_urdf_trifinger"), urdf_robot("urdf_trifinger")
    X = oracle.fkine(rob.fk_desc(), d["q"], np.float64)
    const = [k for k in range(X.shape[2]) if np.ptp(X[:, :, k], axis=0).max() == 0]
    assert const, "expected constant link origins"
    assert relerr(X, d["x64"]) < TOL64


def test_parser_rejects_what_the_reference_cannot_move():
    from diffco_amd.urdf import URDFRobotFK, parse_urdf
    m = urdf_model("urdf_2link")
    bad = urdf_xml(m).replace('type="revolute"', 'type="floating"', 1)
    with pytest.raises(ValueError, match="floating"):
        URDFRobotFK(bad)
    with pytest.raises(ValueError, match="root"):
        parse_urdf("<notrobot/>")
    two_roots = urdf_xml(m).replace("</robot>", '<link name="orphan"/></robot>')
    with pytest.raises(ValueError, match="one root"):
        URDFRobotFK(two_roots)
    with pytest.raises(FileNotFoundError):
        parse_urdf("/nonexistent/robot.urdf")


def test_limits_of_the_compiled_description():
    """more feature links than DCX_MAX_POINTS (32): a clear error, never a silent truncation"""
    from diffco_amd.urdf import URDFRobotFK
    links = ["base"] + [f"l{i}" for i in range(40)]
    joints = [dict(name=f"j{i}", type="revolute", parent=links[i], child=links[i + 1], xyz=[0, 0, 0.1], rpy=[0, 0, 0],
                   axis=[0, 0, 1], lower=-1.0, upper=1.0, mimic_joint=None, mimic_multiplier=1.0, mimic_offset=0.0)
              for i in range(40)]
    with pytest.raises(ValueError, match="compiled limits"):
        URDFRobotFK(urdf_xml(dict(links=links, joints=joints)))


@pytest.mark.parametrize("seed", range(12))
def test_random_trees_against_an_independent_fk(seed):
    """random kinematic trees (all joint types, signed axes, off-axis prismatic directions, mimic joints, branching):
    the compiled DCX_FK_TREE description evaluated by the fp64 oracle equals an independent recursive FK, and the
    oracle's J^T g equals a finite-difference derivative"""
    from helpers import random_urdf_model, reference_tree_fk
    from diffco_amd.urdf import URDFRobotFK
    m = random_urdf_model(seed, n_links=6 + seed % 7)
    try:
        rob = URDFRobotFK(urdf_xml(m))
    except ValueError as e:  # e.g. no movable joint at all: nothing to test
        pytest.skip(str(e))
    if rob.dof == 0 or not rob.unique_position_link_names:
        pytest.skip("degenerate tree")
    rng = np.random.default_rng(100 + seed)
    q = rng.uniform(-1.2, 1.2, (16, rob.dof))
    pos, dof = reference_tree_fk(m, q)
    assert dof == rob.dof
    want = np.stack([pos[ln] for ln in rob.unique_position_link_names], axis=-1)  # [B, 3, L]
    desc = rob.fk_desc()
    got = oracle.fkine(desc, q, np.float64)
    assert np.abs(got - want).max() < 2e-7 * max(1.0, np.abs(want).max())  # only the fp32 storage of folded constants differs
    g = rng.standard_normal(want.shape)
    gq = oracle.fkine_vjp(desc, q, g, np.float64)
    eps = 1e-6
    for i in range(rob.dof):
        dq = np.zeros_like(q)
        dq[:, i] = eps
        fd = ((oracle.fkine(desc, q + dq, np.float64) - oracle.fkine(desc, q - dq, np.float64)) * g).sum(axis=(1, 2)) / (2 * eps)
        assert np.abs(fd - gq[:, i]).max() < 1e-6 * max(1.0, np.abs(gq).max())


def test_undeclared_namespace_prefixes_are_tolerated():
    """simulator blocks such as <gazebo><sensor:camera>…</sensor:camera></gazebo> (the reference's fetch.urdf) use
    prefixes no xmlns declares; they carry no kinematics and must not stop the robot from loading"""
    from diffco_amd.urdf import URDFRobotFK
    m = urdf_model("urdf_2link")
    xml = urdf_xml(m).replace("</robot>", '<gazebo reference="x"><sensor:camera name="rgb" a:b="1"><hfov>50</hfov>'
                                          "</sensor:camera></gazebo></robot>")
    a, b = URDFRobotFK(xml), URDFRobotFK(urdf_xml(m))
    assert a.fk_desc().key() == b.fk_desc().key()
    with pytest.raises(Exception):
        URDFRobotFK("<robot><link name='a'></robot>")  # genuinely malformed XML still fails


# ---- an independent pin for the URDF reader (tools/make_golden_urdf_pin.py: xml.etree + Rodrigues, no diffco_amd) ------
def _pin():
    import json
    import os
    from helpers import GOLDEN
    return json.load(open(os.path.join(GOLDEN, "urdf_pin.json")))


def _check_against_pin(rob, pin):
    """feature-link origins of `rob` (diffco_amd's reader -> tree compiler -> C oracle, float64) against the hand-composed
    link origins, joint values addressed by NAME"""
    assert set(rob.joint_names) == set(pin["q"])                       # movable, non-mimic joints
    assert set(rob.unique_position_link_names) <= set(pin["links"])
    for key, qd in (("at_zero", {k: 0.0 for k in pin["q"]}), ("at_q", pin["q"])):
        q = np.array([[qd[n] for n in rob.joint_names]], dtype=np.float64)
        x = oracle.fkine(rob.fk_desc(), q, np.float64)[0]               # [3, L]
        want = np.array([pin[key][ln] for ln in rob.unique_position_link_names]).T
        assert x.shape == want.shape
        assert np.abs(x - want).max() < 2e-7 * max(1.0, np.abs(want).max()), (key, np.abs(x - want).max())
    # the pin moves: the two poses differ for every link behind a movable joint
    moved = [ln for ln in rob.unique_position_link_names
             if np.abs(np.array(pin["at_q"][ln]) - np.array(pin["at_zero"][ln])).max() > 1e-3]
    assert len(moved) >= max(1, len(rob.unique_position_link_names) // 2)


@pytest.mark.parametrize("name", URDF_NAMES)
def test_reader_against_hand_composed_link_origins(name):
    """the stored joint table (what diffco_amd.urdf.parse_urdf read from the reference's URDF when the fixtures were
    made) reproduces link origins that were composed from the URDF text by another program"""
    _check_against_pin(urdf_robot(name), _pin()[name])


@pytest.mark.parametrize("name", ["urdf_panda", "urdf_allegro", "urdf_fetch", "urdf_trifinger"])
def test_reader_on_the_original_urdf_text(name):
    """the same, with the reader run on the ORIGINAL file (build container only: /root/reference is not on the GPU box)"""
    import os
    pin = _pin()[name]
    path = None
    for base, _, files in os.walk("/root/reference"):
        if "2link_robot.urdf" in files:
            path = os.path.join(base, pin["source"])
    if path is None or not os.path.isfile(path):
        pytest.skip("reference URDF files not present")
    from diffco_amd.urdf import URDFRobotFK
    _check_against_pin(URDFRobotFK(path), pin)
