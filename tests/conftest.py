import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pr2():
    from bio_ik_amd import pr2_like
    return pr2_like()


@pytest.fixture(scope="session")
def snake31():
    from bio_ik_amd import snake
    return snake(31)


@pytest.fixture(scope="session")
def templates(pr2, snake31):
    """The three problem templates of BASELINE.json configs C2, C3, C4."""
    from bio_ik_amd import AvoidJointLimitsGoal, MinimalDisplacementGoal, PoseGoal, ProblemTemplate
    return {
        "c2": ProblemTemplate(pr2, "right_arm", [PoseGoal("r_wrist_roll_link")]),
        "c3": ProblemTemplate(pr2, "all", [PoseGoal("r_wrist_roll_link"), PoseGoal("l_wrist_roll_link"), MinimalDisplacementGoal()]),
        "c4": ProblemTemplate(snake31, "snake", [PoseGoal("tip"), AvoidJointLimitsGoal()]),
    }


@pytest.fixture(scope="session")
def oracles(templates):
    from oracle import orc
    return {k: orc.Oracle(t) for k, t in templates.items()}


def random_configuration(model, rng, n=None):
    """uniform in [min,max] per variable (continuous joints: [-pi,pi]); reference README.md:410-418"""
    lo = np.asarray(model.var_min)
    hi = np.asarray(model.var_max)
    shape = (model.n_variables,) if n is None else (n, model.n_variables)
    return lo + (hi - lo) * rng.random(shape)


def gnarly_robot():
    """A deliberately awkward tree (test fixture only): rotated joint origins, oblique axes, a prismatic joint inside a
    chain, a tip on a fixed link behind the last joint, a tip that hangs off the root without any joint, three
    branches, and a joint outside every goal chain (only reachable through a JointVariableGoal)."""
    from bio_ik_amd import RobotModel
    m = RobotModel("gnarly")
    m.add_link("root")
    m.add_link("plate", "root", "plate_joint", "fixed", xyz=(0.1, -0.2, 0.3), rpy=(0.3, -0.2, 0.5))
    m.add_link("waist", "plate", "waist_joint", "revolute", xyz=(0.0, 0.05, 0.2), rpy=(0.1, 0.2, -0.3), axis=(0.2, -0.3, 0.9), lower=-2.0, upper=2.5, velocity=1.5)
    m.add_link("lift", "waist", "lift_joint", "prismatic", xyz=(0.02, 0.0, 0.1), rpy=(0.0, 0.4, 0.0), axis=(0.1, 0.1, 1.0), lower=-0.1, upper=0.4, velocity=0.2)
    for side, sgn in (("a", 1.0), ("b", -1.0)):
        m.add_link(side + "1", "lift", side + "1_joint", "revolute", xyz=(0.0, sgn * 0.15, 0.05), rpy=(sgn * 0.7, 0.0, 0.2), axis=(0, 1, 0), lower=-1.5, upper=1.2, velocity=2.0)
        m.add_link(side + "1f", side + "1", side + "1f_joint", "fixed", xyz=(0.2, 0.0, 0.0), rpy=(0.0, 0.3, 0.0))
        m.add_link(side + "2", side + "1f", side + "2_joint", "continuous", xyz=(0.1, 0.0, 0.02), rpy=(0.2, 0.0, 0.0), axis=(1, 0, 0), velocity=3.0)
        m.add_link(side + "3", side + "2", side + "3_joint", "revolute", xyz=(0.15, 0.0, 0.0), rpy=(0.0, 0.0, sgn * 0.4), axis=(0.0, 0.6, 0.8), lower=-2.2, upper=0.3, velocity=2.5)
        m.add_link(side + "_tool", side + "3", side + "_tool_joint", "fixed", xyz=(0.05, 0.01, 0.12), rpy=(0.3, 0.3, 0.3))
    m.add_link("head", "waist", "head_joint", "revolute", xyz=(0.0, 0.0, 0.4), rpy=(0.0, 0.0, 0.0), axis=(0, 0, 1), lower=-1.0, upper=1.0, velocity=1.0)
    m.add_link("antenna", "root", "antenna_joint", "revolute", xyz=(0.3, 0.0, 0.0), axis=(0, 0, 1), lower=-0.5, upper=0.5, velocity=1.0)
    joints = ["waist_joint", "lift_joint", "a1_joint", "a2_joint", "a3_joint", "b1_joint", "b2_joint", "b3_joint", "head_joint", "antenna_joint"]
    m.add_group("body", joints=joints, tips=["a_tool", "b_tool", "head"])
    return m


def hand_robot(slide="revolute"):
    """A branching fixture with axis-aligned joint origins (test fixture only): an arm of four joints, a palm on a fixed link, three fingers of two joints with a
    fixed tip link each -- three tips behind a common chain, so the walk parks the palm's frame for the second and third finger.  slide: the type of the arm's third
    joint (a prismatic joint in the MIDDLE of a chain is where the joint program's folded constants round differently from the reference's frame by frame)."""
    from bio_ik_amd import RobotModel
    m = RobotModel("hand")
    m.add_link("base")
    m.add_link("a1", "base", "a1_joint", "revolute", xyz=(0, 0, 0.1), axis=(0, 0, 1), lower=-2.0, upper=2.0, velocity=1.0)
    m.add_link("a2", "a1", "a2_joint", "revolute", xyz=(0.2, 0, 0), axis=(0, 1, 0), lower=-1.5, upper=1.5, velocity=1.5)
    if slide == "prismatic":
        m.add_link("slide", "a2", "slide_joint", "prismatic", xyz=(0.1, 0, 0), axis=(1, 0, 0), lower=-0.05, upper=0.2, velocity=0.3)
    else:
        m.add_link("slide", "a2", "slide_joint", "revolute", xyz=(0.1, 0, 0), axis=(0, 0, 1), lower=-0.5, upper=0.7, velocity=0.3)
    m.add_link("a3", "slide", "a3_joint", "continuous", xyz=(0.15, 0, 0), axis=(1, 0, 0), velocity=2.0)
    m.add_link("palm", "a3", "palm_joint", "fixed", xyz=(0, 0, 0))
    joints = ["a1_joint", "a2_joint", "slide_joint", "a3_joint"]
    tips = []
    for i, (y, ax) in enumerate(((0.04, (0, 1, 0)), (0.0, (0, 0, 1)), (-0.04, (0, 1, 0)))):
        f = "f%d" % i
        m.add_link(f + "a", "palm", f + "a_joint", "revolute", xyz=(0.03, y, 0), axis=ax, lower=-1.0, upper=1.2, velocity=3.0)
        m.add_link(f + "b", f + "a", f + "b_joint", "revolute", xyz=(0.04, 0, 0), axis=(0, 1, 0), lower=-0.2, upper=1.6, velocity=3.0)
        m.add_link(f + "_tip", f + "b", f + "_tip_joint", "fixed", xyz=(0.03, 0, 0))
        joints += [f + "a_joint", f + "b_joint"]
        tips.append(f + "_tip")
    m.add_group("hand", joints=joints, tips=tips)
    return m


@pytest.fixture(scope="session")
def gnarly():
    return gnarly_robot()


def mobile_robot(base="floating", reach=1.0):
    """A 3-joint arm on a free base (test fixture only): the base joint is a MoveIt floating joint (7 variables: translation +
    quaternion) or a planar joint (x, y, theta); its variables are genes like any other (forward_kinematics.h:120-135).
    The translation variables are bounded to +-reach, as a joint_limits.yaml of a MoveIt configuration would do."""
    from bio_ik_amd import RobotModel
    m = RobotModel("mobile_" + base)
    m.add_link("world")
    m.add_link("base", "world", "base_joint", base, xyz=(0.0, 0.0, 0.1), velocity=1.0)
    m.add_link("l1", "base", "j1", "revolute", xyz=(0.1, 0.0, 0.2), axis=(0, 0, 1), lower=-2.5, upper=2.5, velocity=2.0)
    m.add_link("l2", "l1", "j2", "revolute", xyz=(0.0, 0.0, 0.1), axis=(0, 1, 0), lower=-1.8, upper=1.8, velocity=2.0)
    m.add_link("l3", "l2", "j3", "revolute", xyz=(0.3, 0.0, 0.0), axis=(0, 1, 0), lower=-2.2, upper=2.2, velocity=2.5)
    m.add_link("tool", "l3", "tool_joint", "fixed", xyz=(0.25, 0.0, 0.0))
    m.add_group("whole", joints=["base_joint", "j1", "j2", "j3"], tips=["tool", "base"])
    for i, name in enumerate(m.variable_names):
        if name.startswith("base_joint/") and name.split("/")[1] in ("trans_x", "trans_y", "trans_z", "x", "y"):
            m.var_min[i], m.var_max[i], m.var_bounded[i] = -reach, reach, 1
    m._keep = None
    return m


def stage_robot(mid="planar", with_base=False):
    """Test fixture only: an arm with a multi-variable joint in the MIDDLE of its chain -- a planar stage (x, y, theta) or a floating coupling (translation +
    quaternion) between the shoulder and the elbow --, optionally on a planar base as well (two such joints on one chain).  The reference's FK takes such
    joints wherever they are (forward_kinematics.h:120-135, 331-354); until round 5 the device took one, directly behind the root."""
    from bio_ik_amd import RobotModel
    m = RobotModel("stage_" + mid + ("_on_base" if with_base else ""))
    m.add_link("world")
    parent = "world"
    if with_base:
        m.add_link("base", "world", "base_joint", "planar", xyz=(0.0, 0.0, 0.05), velocity=1.0)
        parent = "base"
    m.add_link("l1", parent, "j1", "revolute", xyz=(0.0, 0.0, 0.2), axis=(0, 0, 1), lower=-2.5, upper=2.5, velocity=2.0)
    m.add_link("l2", "l1", "j2", "revolute", xyz=(0.0, 0.1, 0.1), axis=(0, 1, 0), lower=-1.8, upper=1.8, velocity=2.0)
    m.add_link("stage", "l2", "stage_joint", mid, xyz=(0.3, 0.0, 0.0), velocity=1.0)
    m.add_link("l3", "stage", "j3", "revolute", xyz=(0.1, 0.0, 0.0), axis=(0, 1, 0), lower=-2.2, upper=2.2, velocity=2.5)
    m.add_link("l4", "l3", "j4", "revolute", xyz=(0.25, 0.0, 0.0), axis=(1, 0, 0), lower=-3.0, upper=3.0, velocity=3.0)
    m.add_link("tool", "l4", "tool_joint", "fixed", xyz=(0.15, 0.0, 0.0))
    m.add_group("whole", joints=(["base_joint"] if with_base else []) + ["j1", "j2", "stage_joint", "j3", "j4"], tips=["tool", "stage"])
    for i, name in enumerate(m.variable_names):
        j, _, v = name.partition("/")
        if j in ("stage_joint", "base_joint") and v in ("trans_x", "trans_y", "trans_z", "x", "y"):
            m.var_min[i], m.var_max[i], m.var_bounded[i] = (-0.2, 0.2, 1) if j == "stage_joint" else (-1.0, 1.0, 1)
    m._keep = None
    return m


def mimic_robot(chain=None):
    """Axis-aligned 6-joint arm (test fixture only) with two mimic joints (MoveIt JointModel::getMimic): the second elbow
    follows the shoulder pitch (a gene), and a finger on the tip's chain follows a finger that is on no goal chain (so the
    followed joint is not a gene and keeps the seed's value, reference problem.cpp:191-204).
    chain = "chain": the first wrist joint follows the second elbow, which follows the shoulder (a mimic of a mimic: resolved to the joint at the end
    of the chain as MoveIt's RobotModel::buildMimic does); chain = "resolved": the same robot with that resolution written out by hand."""
    from bio_ik_amd import RobotModel
    m = RobotModel("mimic_arm")
    m.add_link("base")
    m.add_link("l1", "base", "s1", "revolute", xyz=(0.0, 0.0, 0.3), axis=(0, 0, 1), lower=-2.5, upper=2.5, velocity=2.0)
    m.add_link("l2", "l1", "s2", "revolute", xyz=(0.0, 0.1, 0.0), axis=(0, 1, 0), lower=-1.8, upper=1.8, velocity=2.0)
    m.add_link("l3", "l2", "e1", "revolute", xyz=(0.35, 0.0, 0.0), axis=(0, 1, 0), lower=-2.2, upper=2.2, velocity=2.5)
    m.add_link("l4", "l3", "e2", "revolute", xyz=(0.25, 0.0, 0.0), axis=(0, 1, 0), lower=-3.0, upper=3.0, velocity=2.5, mimic=("s2", -0.5, 0.1))
    w1_mimic = {None: None, "chain": ("e2", 0.8, -0.2), "resolved": ("s2", 0.8 * -0.5, -0.2 + 0.8 * 0.1)}[chain]
    m.add_link("l5", "l4", "w1", "revolute", xyz=(0.2, 0.0, 0.0), axis=(1, 0, 0), lower=-3.0, upper=3.0, velocity=3.0, mimic=w1_mimic)
    m.add_link("l6", "l5", "w2", "revolute", xyz=(0.1, 0.0, 0.0), axis=(0, 1, 0), lower=-2.0, upper=2.0, velocity=3.0)
    # (the children of l6 in the order a MoveIt-loaded model has them: alphabetical by joint name, bio_ik_amd/urdf.py)
    m.add_link("finger_l", "l6", "finger_l_joint", "prismatic", xyz=(0.05, 0.03, 0.0), axis=(0, 1, 0), lower=0.0, upper=0.04, velocity=0.1)
    m.add_link("finger_r", "l6", "finger_r_joint", "prismatic", xyz=(0.05, -0.03, 0.0), axis=(0, -1, 0), lower=0.0, upper=0.04, velocity=0.1,
               mimic=("finger_l_joint", 1.0, 0.0))
    m.add_link("finger_r_tip", "finger_r", "finger_r_tip_joint", "fixed", xyz=(0.04, 0.0, 0.0))
    m.add_link("tool", "l6", "tool_joint", "fixed", xyz=(0.08, 0.0, 0.0))
    m.add_group("arm", joints=["s1", "s2", "e1", "e2", "w1", "w2", "finger_l_joint", "finger_r_joint"], tips=["tool", "finger_r_tip"])
    return m


def balance_robot():
    """A small two-armed torso whose links carry URDF inertials (test fixture only): BalanceGoal reads the frame of EVERY link with a
    positive mass (goal_types.cpp:231-255) -- links on the arms, a link behind a fixed joint, the root itself, and a link on a branch
    no other goal touches (`ballast`, moved by a joint of the group)."""
    from bio_ik_amd import RobotModel
    m = RobotModel("balance")
    m.add_link("pelvis", mass=6.0, com=(0.01, 0.0, 0.05))
    m.add_link("torso", "pelvis", "waist", "revolute", xyz=(0.0, 0.0, 0.2), axis=(0, 0, 1), lower=-1.5, upper=1.5, velocity=1.5, mass=9.0, com=(0.0, 0.01, 0.25))
    m.add_link("pack", "torso", "pack_joint", "fixed", xyz=(-0.15, 0.0, 0.3), rpy=(0.0, 0.2, 0.0), mass=3.0, com=(-0.05, 0.0, 0.0))
    for side, sgn in (("a", 1.0), ("b", -1.0)):
        m.add_link(side + "_upper", "torso", side + "_shoulder", "revolute", xyz=(0.0, sgn * 0.2, 0.45), rpy=(sgn * 0.3, 0.0, 0.0), axis=(0, 1, 0), lower=-2.0, upper=2.0,
                   velocity=2.0, mass=2.0, com=(0.12, 0.0, 0.0))
        m.add_link(side + "_lower", side + "_upper", side + "_elbow", "revolute", xyz=(0.3, 0.0, 0.0), axis=(0, 0.6, 0.8), lower=-2.2, upper=0.4, velocity=2.5,
                   mass=1.2, com=(0.1, 0.01, 0.0))
        m.add_link(side + "_hand", side + "_lower", side + "_wrist", "continuous", xyz=(0.25, 0.0, 0.0), axis=(1, 0, 0), velocity=3.0, mass=0.4, com=(0.04, 0.0, 0.0))
        m.add_link(side + "_tool", side + "_hand", side + "_tool_joint", "fixed", xyz=(0.08, 0.0, 0.0))
    m.add_link("ballast", "pelvis", "ballast_slide", "prismatic", xyz=(0.0, 0.0, -0.1), axis=(1, 0, 0), lower=-0.3, upper=0.3, velocity=0.5, mass=4.0, com=(0.0, 0.0, -0.05))
    joints = ["waist", "a_shoulder", "a_elbow", "a_wrist", "b_shoulder", "b_elbow", "b_wrist", "ballast_slide"]
    m.add_group("body", joints=joints, tips=["a_tool", "b_tool"])
    return m


def gnarly_goals():
    """one goal of every device opcode, spread over four tips (one of them the joint-less `plate`)"""
    from bio_ik_amd import (AvoidJointLimitsGoal, CenterJointsGoal, ConeGoal, DirectionGoal, JointVariableGoal, LineGoal, LookAtGoal,
                            MaxDistanceGoal, MinDistanceGoal, MinimalDisplacementGoal, OrientationGoal, PlaneGoal, PoseGoal, PositionGoal,
                            RegularizationGoal, SideGoal)
    sec = MinimalDisplacementGoal(weight=0.7)
    sec.secondary_ = True
    sec2 = JointVariableGoal("head_joint", 0.2, weight=0.5)
    sec2.secondary_ = True
    return [
        PoseGoal("a_tool", (0.4, 0.2, 0.6), (0.1, 0.2, 0.3, 0.9), weight=1.0),
        PositionGoal("b_tool", (0.3, -0.3, 0.5), weight=0.8),
        OrientationGoal("b_tool", (0.0, 0.3, 0.1, 0.9), weight=0.6),
        LookAtGoal("head", (1, 0, 0), (1.0, 0.5, 0.7), weight=0.5),
        MaxDistanceGoal("a3", (0.2, 0.2, 0.2), 0.3, weight=1.1),
        MinDistanceGoal("a3", (0.25, 0.2, 0.5), 0.4, weight=0.9),
        LineGoal("b3", (0.0, 0.0, 0.5), (0.0, 1.0, 0.0), weight=0.4),
        PlaneGoal("b3", (0.0, 0.0, 0.6), (0.0, 0.0, 1.0), weight=0.3),
        SideGoal("a_tool", (0, 0, 1), (0, 1, 0), weight=0.7),
        DirectionGoal("b_tool", (0, 0, 1), (1, 0, 0), weight=0.2),
        ConeGoal("head", (1, 0, 0), (0, 0, 1), 0.3, weight=0.6, position=(0.1, 0.0, 0.9), position_weight=0.5),
        PositionGoal("plate", (0.1, -0.2, 0.31), weight=0.05),
        AvoidJointLimitsGoal(weight=0.3),
        CenterJointsGoal(weight=0.2),
        RegularizationGoal(weight=0.1),
        JointVariableGoal("antenna_joint", 0.1, weight=0.4),
        sec,
        sec2,
    ]


@pytest.fixture(scope="session")
def hostsim_lib():
    """TEST INFRASTRUCTURE: the kernel bodies of bio_ik_amd/csrc built for the host (tests/hostsim)."""
    import subprocess
    from bio_ik_amd import solver
    d = os.path.join(ROOT, "tests", "hostsim")
    subprocess.run(["make", "-C", d, "-s"], check=True)
    return solver.load_library(os.path.join(d, "libbioik_hostsim.so"))


@pytest.fixture(scope="session")
def hostsim_shim(hostsim_lib):
    """TEST INFRASTRUCTURE: the Python package's plugin shim (bio_ik_amd/cpp/src/plugin_shim.cpp) linked against the host simulator"""
    import subprocess
    d = os.path.join(ROOT, "tests", "hostsim")
    out = os.path.join(d, "libbio_ik_shim_hostsim.so")
    subprocess.run(["make", "-C", os.path.join(ROOT, "bio_ik_amd", "cpp"), "-s", "shim", "SOLVER_DIR=" + d, "SOLVER=bioik_hostsim", "SHIM_OUT=" + out], check=True)
    return out
