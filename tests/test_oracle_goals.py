"""Goal costs of the oracle (reference include/bio_ik/goal_types.h, src/problem.cpp:244-341): closed-form known
answers and an independent NumPy evaluation of every opcode."""
import numpy as np
import pytest

from bio_ik_amd import (AvoidJointLimitsGoal, CenterJointsGoal, ConeGoal, DirectionGoal, JointVariableGoal, LineGoal,
                        LookAtGoal, MaxDistanceGoal, MinDistanceGoal, MinimalDisplacementGoal, OrientationGoal, PlaneGoal,
                        PoseGoal, PositionGoal, ProblemTemplate, RegularizationGoal, SideGoal, abi)
from conftest import random_configuration
from np_fk import quat_to_rot64
from oracle import orc

TIP = "r_wrist_roll_link"


def one_goal_oracle(pr2, goal, group="right_arm"):
    t = ProblemTemplate(pr2, group, [goal])
    return t, orc.Oracle(t)


def test_pose_goal_known_answers(pr2):
    rng = np.random.default_rng(20)
    seed = random_configuration(pr2, rng)
    t, o = one_goal_oracle(pr2, PoseGoal(TIP))
    genes = seed[o.active_variables]
    tip = o.fk_genes(seed, genes)[0, 0]
    g = PoseGoal(TIP, tip[:3], tip[3:])
    # tip == goal -> 0
    assert o.fitness(abi.FK_EXACT, seed, g.params(), genes)[0][0] == pytest.approx(0.0, abs=1e-28)
    # q and -q give identical cost
    g2 = PoseGoal(TIP, tip[:3], -tip[3:])
    assert o.fitness(abi.FK_EXACT, seed, g2.params(), genes)[0][0] == pytest.approx(0.0, abs=1e-28)
    # pure translation by d -> |d|^2
    d = np.array([0.01, -0.02, 0.03])
    g3 = PoseGoal(TIP, tip[:3] + d, tip[3:])
    assert o.fitness(abi.FK_EXACT, seed, g3.params(), genes)[0][0] == pytest.approx(d @ d, rel=1e-12)
    # 180 degree rotation error: dot(q*, q) = 0 -> rs^2 * 2 = 0.5 at rs = 0.5 (goal_types.h:172)
    q = tip[3:]
    q180 = orc.quat_mul_quat(q, [1, 0, 0, 0])
    g4 = PoseGoal(TIP, tip[:3], q180)
    assert o.fitness(abi.FK_EXACT, seed, g4.params(), genes)[0][0] == pytest.approx(0.5, rel=1e-12)
    g4.setRotationScale(0.0)  # position_only_ik, kinematics_plugin.cpp:290-295
    assert o.fitness(abi.FK_EXACT, seed, g4.params(), genes)[0][0] == pytest.approx(0.0, abs=1e-28)
    # weight enters squared (problem.cpp:151, 248)
    g5 = PoseGoal(TIP, tip[:3] + d, tip[3:], weight=3.0)
    t5 = ProblemTemplate(pr2, "right_arm", [g5])
    o5 = orc.Oracle(t5)
    assert o5.fitness(abi.FK_EXACT, seed, g5.params(), genes)[0][0] == pytest.approx(9 * (d @ d), rel=1e-12)


def np_goal_cost(goal, frame, genes, seed, o, model):
    """independent evaluation with rotation matrices"""
    p, R = frame[:3], quat_to_rot64(frame[3:])
    q = frame[3:]
    P = goal.params()
    op = goal.opcode
    lo = np.asarray(model.var_min)[o.active_variables]
    hi = np.asarray(model.var_max)[o.active_variables]
    info = o.robot_info()[o.active_variables]
    bounded = info[:, 1] != np.finfo(float).max
    w = o.velocity_weights()
    if op == abi.GOAL_POSITION:
        return np.sum((p - P[:3]) ** 2)
    if op == abi.GOAL_ORIENTATION:
        return min(np.sum((P - q) ** 2), np.sum((P + q) ** 2))
    if op == abi.GOAL_POSE:
        return np.sum((p - P[:3]) ** 2) + P[7] ** 2 * min(np.sum((P[3:7] - q) ** 2), np.sum((P[3:7] + q) ** 2))
    if op == abi.GOAL_LOOK_AT:
        a = R @ P[:3]
        d = P[3:6] - p
        return np.sum((d / np.linalg.norm(d) - a / np.linalg.norm(a)) ** 2)
    if op == abi.GOAL_MAX_DISTANCE:
        return max(0.0, np.linalg.norm(p - P[:3]) - P[3]) ** 2
    if op == abi.GOAL_MIN_DISTANCE:
        return max(0.0, P[3] - np.linalg.norm(p - P[:3])) ** 2
    if op == abi.GOAL_LINE:
        pos, d = P[:3], P[3:6]
        return np.sum((pos - (p - d * np.dot(d, p - pos))) ** 2)
    if op == abi.GOAL_PLANE:
        return np.dot(p - P[:3], P[3:6]) ** 2
    if op == abi.GOAL_AVOID_JOINT_LIMITS:
        d = np.maximum(0.0, np.abs(genes - (lo + hi) * 0.5) * 2.0 - info[:, 2] * 0.5) * w
        return np.sum((d * bounded) ** 2)
    if op == abi.GOAL_CENTER_JOINTS:
        d = (genes - (lo + hi) * 0.5) * w
        return np.sum((d * bounded) ** 2)
    if op == abi.GOAL_REGULARIZATION:
        return np.sum((genes - seed[o.active_variables]) ** 2)
    if op == abi.GOAL_MINIMAL_DISPLACEMENT:
        return np.sum(((genes - seed[o.active_variables]) * w) ** 2)
    if op == abi.GOAL_JOINT_VARIABLE:
        vi = list(o.active_variables).index(model.variable_index(goal.variable_name()))
        return (P[0] - genes[vi]) ** 2
    if op == abi.GOAL_SIDE:
        return max(0.0, np.dot(R @ P[:3], P[3:6])) ** 2
    if op == abi.GOAL_DIRECTION:
        return np.sum((R @ P[:3] - P[3:6]) ** 2)
    if op == abi.GOAL_CONE:
        v = R @ P[4:7]
        ang = np.arccos(np.clip(np.dot(v, P[7:10]) / np.sqrt(np.dot(v, v) * np.dot(P[7:10], P[7:10])), -1, 1))
        return max(0.0, ang - P[10]) ** 2 + P[3] ** 2 * np.sum((P[:3] - p) ** 2)
    raise AssertionError(op)


GOALS = [
    lambda: PositionGoal(TIP, (0.5, -0.2, 0.8)),
    lambda: OrientationGoal(TIP, (0.1, 0.2, 0.3, 0.9)),
    lambda: PoseGoal(TIP, (0.5, -0.2, 0.8), (0.1, 0.2, 0.3, 0.9)),
    lambda: LookAtGoal(TIP, (1, 0, 0), (1.0, 0.3, 0.9)),
    lambda: MaxDistanceGoal(TIP, (0.4, -0.1, 0.7), 0.2),
    lambda: MinDistanceGoal(TIP, (0.4, -0.1, 0.7), 0.6),
    lambda: LineGoal(TIP, (0.4, -0.1, 0.7), (1, 1, 0)),
    lambda: PlaneGoal(TIP, (0.4, -0.1, 0.7), (0, 1, 1)),
    lambda: AvoidJointLimitsGoal(1.0, False),
    lambda: CenterJointsGoal(1.0, False),
    lambda: RegularizationGoal(1.0),
    lambda: MinimalDisplacementGoal(1.0, False),
    lambda: JointVariableGoal("r_elbow_flex_joint", -1.0),
    lambda: SideGoal(TIP, (0, 0, 1), (0, 0, 1)),
    lambda: DirectionGoal(TIP, (0, 0, 1), (0, 1, 0)),
    lambda: ConeGoal(TIP, (0, 0, 1), (0, 0, 1), 0.2, position=(0.4, -0.1, 0.7), position_weight=0.5),
]


@pytest.mark.parametrize("mk", GOALS)
def test_every_goal_opcode_against_numpy(pr2, mk):
    goal = mk()
    goal.setWeight(1.5)
    t = ProblemTemplate(pr2, "right_arm", [goal] + ([PoseGoal(TIP)] if goal.link_name() is None else []))
    o = orc.Oracle(t)
    rng = np.random.default_rng(21)
    for _ in range(5):
        seed = random_configuration(pr2, rng)
        genes = random_configuration(pr2, rng)[o.active_variables]
        frame = o.fk_genes(seed, genes)[0, 0]
        params = t.pack_params()
        if goal.link_name() is None:
            params[-1] = 0.0  # auxiliary pose goal with rotation_scale 0 ...
            params[-8:-5] = frame[:3]  # ... and zero position error: contributes exactly 0
        prim, sec = o.fitness(abi.FK_EXACT, seed, params, genes)
        expect = 1.5 ** 2 * np_goal_cost(goal, frame, genes, seed, o, pr2)
        got = sec[0] if goal.isSecondary() else prim[0]
        assert got == pytest.approx(expect, rel=1e-11, abs=1e-15)


def test_secondary_goals_are_summed_separately(oracles, templates):
    """ik_base.h:163-185: primary and secondary sums; MinimalDisplacementGoal is secondary by default (goal_types.h:450)."""
    o, t = oracles["c3"], templates["c3"]
    rng = np.random.default_rng(22)
    seed = random_configuration(t.model, rng)
    genes = random_configuration(t.model, rng)[o.active_variables]
    prim, sec = o.fitness(abi.FK_EXACT, seed, t.pack_params(), genes)
    w = o.velocity_weights()
    assert sec[0] == pytest.approx(np.sum(((genes - seed[o.active_variables]) * w) ** 2), rel=1e-12)
    assert prim[0] > 0


def test_check_solution_dtwist_and_dpos_drot(pr2):
    """problem.cpp:306-324: every twist component < dtwist (default 1e-5); dpos metres / drot DEGREES when set."""
    rng = np.random.default_rng(23)
    t, o = one_goal_oracle(pr2, PoseGoal(TIP))
    seed = random_configuration(pr2, rng)
    genes = seed[o.active_variables]
    tip = o.fk_genes(seed, genes)[0, 0]
    p = abi.default_solve_params()
    def shifted(dp, ang):
        d = np.concatenate([dp, [np.sin(ang / 2), 0, 0, np.cos(ang / 2)]])
        g = orc.frame_concat(tip, d)
        return np.concatenate([g, [0.5]])
    assert o.check(p, seed, shifted([0, 0, 0], 0.0), genes)[0] == 1
    assert o.check(p, seed, shifted([0.9e-5, 0, 0], 0.9e-5), genes)[0] == 1
    assert o.check(p, seed, shifted([1.1e-5, 0, 0], 0.0), genes)[0] == 0
    assert o.check(p, seed, shifted([0, 0, 0], 1.1e-5), genes)[0] == 0
    p2 = abi.default_solve_params(dtwist=-1.0, dpos=1e-3, drot=1.0)  # 1 mm, 1 degree
    assert o.check(p2, seed, shifted([0.9e-3, 0, 0], np.radians(0.9)), genes)[0] == 1
    assert o.check(p2, seed, shifted([1.1e-3, 0, 0], 0.0), genes)[0] == 0
    assert o.check(p2, seed, shifted([0, 0, 0], np.radians(1.1)), genes)[0] == 0
    # non-pose goals: fitness < min(dpos, dtwist)^2 (problem.cpp:327-334)
    t3 = ProblemTemplate(pr2, "right_arm", [LineGoal(TIP, tip[:3], (1, 0, 0))])
    o3 = orc.Oracle(t3)
    assert o3.check(p, seed, t3.pack_params(), genes)[0] == 1
    t4 = ProblemTemplate(pr2, "right_arm", [LineGoal(TIP, tip[:3] + [0, 2e-5, 0], (1, 0, 0))])
    o4 = orc.Oracle(t4)
    assert o4.check(p, seed, t4.pack_params(), genes)[0] == 0


def test_problem_errors(pr2):
    with pytest.raises(KeyError):
        ProblemTemplate(pr2, "right_arm", [PoseGoal("no_such_link")])
    with pytest.raises(KeyError):
        ProblemTemplate(pr2, "right_arm", [JointVariableGoal("no_such_joint", 0.0)])
    # a variable outside the group -> reference ERROR("joint variable not found") (problem.cpp:125)
    t = ProblemTemplate(pr2, "right_arm", [PoseGoal(TIP), JointVariableGoal("l_elbow_flex_joint", 0.0)])
    with pytest.raises(orc.OracleError):
        orc.Oracle(t)


def test_fixed_joints_option(pr2):
    """goal.h:124 + problem.cpp:104-114, 198-200: fixed joints drop out of the active variables."""
    t = ProblemTemplate(pr2, "right_arm", [PoseGoal(TIP)], fixed_joints=["r_forearm_roll_joint"])
    o = orc.Oracle(t)
    assert o.D == 6
    assert pr2.variable_index("r_forearm_roll_joint") not in list(o.active_variables)
    # goal variables come first in the active list (problem.cpp:145-148 before :201-204)
    t2 = ProblemTemplate(pr2, "right_arm", [JointVariableGoal("r_wrist_flex_joint", -0.5), PoseGoal(TIP)])
    o2 = orc.Oracle(t2)
    assert o2.active_variables[0] == pr2.variable_index("r_wrist_flex_joint") and o2.D == 7
