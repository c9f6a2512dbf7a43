"""Parity tests proper: the HIP path (libbioik_hip.so on a real MI355X, through the C-ABI) against the CPU oracle.

Three levels: function level (same genes -> same frames / fitness / tables / children / success flags),
trajectory level (same RNG streams -> the same solution, bit for bit: both sides use bioik_sincos and the explicitly fused IEEE
arithmetic), result level at BASELINE.json's full sizes (every reported success reproduces its goal pose under the
ORACLE's exact FK within 1e-4 m / 1e-3 rad, joints inside their limits, success rate equal to the oracle's on a sample)."""
import os

import numpy as np
import pytest

import parity_cases as pc
from bio_ik_amd import ProblemTemplate, abi
from bio_ik_amd.workload import make_queries
from conftest import gnarly_goals
from oracle import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

POS_TOL, ROT_TOL = 1e-4, 1e-3  # north-star tolerance on PoseGoal results [m], [rad]


@pytest.fixture(scope="module", autouse=True)
def shared_trigonometry():
    orc.set_trig_mode(1)
    yield
    orc.set_trig_mode(0)


@pytest.fixture(scope="module")
def gpus(templates):
    from bio_ik_amd.solver import HipSolver, device_count
    assert device_count() >= 1, "no HIP device: the product path has no CPU fallback"
    return {k: HipSolver(t, device=0) for k, t in templates.items()}


@pytest.mark.parametrize("cfg", ["c2", "c3", "c4"])
def test_function_level(gpus, oracles, templates, cfg):
    pc.function_level(gpus[cfg], oracles[cfg], templates[cfg].model, np.random.default_rng(1), n=3000, exact_bits=True)


def test_function_level_gnarly(gnarly):
    from bio_ik_amd.solver import HipSolver
    t = ProblemTemplate(gnarly, "body", gnarly_goals())
    pc.function_level(HipSolver(t), orc.Oracle(t), gnarly, np.random.default_rng(2), n=2000)
    t2 = ProblemTemplate(gnarly, "body", gnarly_goals(), fixed_joints=["lift_joint", "antenna_joint"])
    pc.function_level(HipSolver(t2), orc.Oracle(t2), gnarly, np.random.default_rng(3), n=2000)


def test_mimic_of_a_mimic():
    """a joint that follows a joint that itself follows a gene: resolved to the joint at the end of the chain as MoveIt's RobotModel::buildMimic does; the
    oracle's trajectories, and the same bits as the robot with the resolution written out by hand"""
    from bio_ik_amd import PoseGoal
    from bio_ik_amd.solver import HipSolver
    from conftest import mimic_robot
    sols = []
    for chain in ("chain", "resolved"):
        m = mimic_robot(chain)
        t = ProblemTemplate(m, "arm", [PoseGoal("tool")])
        h, o = HipSolver(t, device=0), orc.Oracle(t)
        assert h.D == o.D == 4
        pc.function_level(h, o, m, np.random.default_rng(6), n=40, exact_bits=True)
        pc.trajectory(h, o, t, n=4, pop=128, steps_list=(3,))
        seeds, params, _ = make_queries(t, h.active_variables, h.fk_genes, 64, seed=8)
        sols.append(h.solve_batch(abi.default_solve_params(population=128, max_steps=24, random_seed=2), seeds, params))
    assert all(np.array_equal(x, y) for x, y in zip(*sols))


def test_mimic_joints():
    """a joint that follows a gene and a joint that follows a joint outside every goal chain (MoveIt mimic joints,
    forward_kinematics.h:230-246, 623-636): function level, and whole solves bit for bit"""
    from bio_ik_amd import MinimalDisplacementGoal, PoseGoal, PositionGoal
    from bio_ik_amd.solver import HipSolver
    from conftest import mimic_robot
    m = mimic_robot()
    sec = MinimalDisplacementGoal(weight=0.5)
    sec.secondary_ = True
    t = ProblemTemplate(m, "arm", [PoseGoal("tool"), PositionGoal("finger_r_tip", weight=0.3), sec])
    h, o = HipSolver(t), orc.Oracle(t)
    assert h.D == o.D == 5
    pc.function_level(h, o, m, np.random.default_rng(5), n=500)
    t2 = ProblemTemplate(m, "arm", [PoseGoal("tool"), sec])
    h2, o2 = HipSolver(t2), orc.Oracle(t2)
    pc.function_level(h2, o2, m, np.random.default_rng(6), n=500, exact_bits=True)
    pc.trajectory(h2, o2, t2, n=16, pop=128, steps_list=(1, 6))
    pc.trajectory(h2, o2, t2, n=8, pop=70, steps_list=(3,), fk_mode=abi.FK_LINEAR)

def test_urdf_loaded_model_solves_on_the_device():
    """URDF + SRDF text -> bio_ik_amd.urdf.load_urdf -> flat model -> libbioik_hip.so: the device solves on a robot description that
    never went through a hand-built fixture (mimic joints, nested SRDF groups), function level and whole solves against the oracle"""
    from bio_ik_amd import MinimalDisplacementGoal, PoseGoal
    from bio_ik_amd.solver import HipSolver
    from bio_ik_amd.urdf import load_urdf
    from test_urdf import SRDF, URDF
    m = load_urdf(URDF, SRDF)
    sec = MinimalDisplacementGoal(weight=0.5)
    sec.secondary_ = True
    t = ProblemTemplate(m, "arm_chain", [PoseGoal("tool"), sec])
    h, o = HipSolver(t), orc.Oracle(t)
    assert h.D == o.D == 5
    pc.function_level(h, o, m, np.random.default_rng(15), n=400, exact_bits=True)
    pc.trajectory(h, o, t, n=16, pop=128, steps_list=(1, 5))
    with pc.oracle_arithmetic(0):
        seeds, params, _ = make_queries(t, o.active_variables, o.fk_genes, 512, seed=33)
    p = abi.default_solve_params(population=128, max_steps=64, random_seed=2)
    sol, fit, suc, steps = h.solve_batch(p, seeds, params)
    so = o.solve_batch(p, orc.RNG_COUNTER, seeds[:32], params[:32], n_threads=8)
    assert np.array_equal(so[0], sol[:32]) and np.array_equal(so[2], suc[:32])
    assert suc.mean() > 0.6  # (an arm whose second elbow follows the shoulder, with a secondary goal: about three in four within 64 steps)
    with pc.oracle_arithmetic(0):
        perr, rerr = pc.pose_errors(o, sol, params)
    assert perr[suc == 1].max() < POS_TOL and rerr[suc == 1].max() < ROT_TOL


def test_balance_goal():
    """BalanceGoal on the device (goal_types.cpp:231-272): ten links with mass = ten more tips, centre of mass accumulated along the
    chain walk.  Function level against the reference-pinned oracle arithmetic, and 512 FK -> IK -> FK round trips on pose + balance."""
    from bio_ik_amd import AvoidJointLimitsGoal, BalanceGoal, PoseGoal
    from bio_ik_amd.solver import BioIKError, HipSolver
    from conftest import balance_robot
    m = balance_robot()
    for goals in ([PoseGoal("a_tool"), BalanceGoal((0.02, -0.01, 0.0), weight=0.8)], [BalanceGoal((0.0, 0.0, 0.0))],
                  [BalanceGoal((0.01, 0.0, 0.0)), PoseGoal("b_tool"), AvoidJointLimitsGoal(weight=0.2)]):
        t = ProblemTemplate(m, "body", goals)
        h, o = HipSolver(t), orc.Oracle(t)
        assert h.T == o.T >= 10 and np.array_equal(h.tip_links, o.tip_links)
        for mode in (0, 1):
            with pc.oracle_arithmetic(mode):
                pc.function_level(h, o, m, np.random.default_rng(15), n=1000, frame_tol=1e-12, fit_rtol=1e-10)
    t = ProblemTemplate(m, "body", [PoseGoal("a_tool"), BalanceGoal(weight=1.0)])
    h, o = HipSolver(t), orc.Oracle(t)
    with pc.oracle_arithmetic(0):
        seeds, params, off = pc.balance_queries(t, o, 512, seed=8)
    sol, fit, suc, steps = h.solve_batch(abi.default_solve_params(population=128, max_steps=96, random_seed=4), seeds, params)
    assert suc.mean() > 0.8
    with pc.oracle_arithmetic(0):
        perr, rerr = pc.pose_errors(o, sol, params)
        berr = pc.balance_errors(t, o, sol, params, off)
    assert perr[suc == 1].max() < POS_TOL and rerr[suc == 1].max() < ROT_TOL and berr[suc == 1].max() < POS_TOL
    from bio_ik_amd import pr2_like
    with pytest.raises(BioIKError) as e:  # a model without inertials cannot carry a BalanceGoal
        HipSolver(ProblemTemplate(pr2_like(), "right_arm", [PoseGoal("r_wrist_roll_link"), BalanceGoal()]))
    assert e.value.code == abi.ERR_INVALID_ARGUMENT


@pytest.mark.parametrize("cfg", ["c2", "c3", "c4"])
def test_gradient_descent_and_jacobian_solvers(gpus, oracles, templates, cfg):
    """modes gd_c and jac (reference src/ik_gradient.cpp:136-251, 42-133) on the device, one wavefront per query (k_solve_point)"""
    h, o, t = gpus[cfg], oracles[cfg], templates[cfg]
    pc.point_solvers(h, o, t, n=64, exact_jac=True)  # (round 6: jac's twist goes through the shared acos, bioik_acos.h -- the same bits as every other mode)
    if cfg == "c2":  # result level: jac converges on tracking queries; every success reproduces its pose under the reference-pinned FK
        with pc.oracle_arithmetic(0):
            seeds, params, _ = make_queries(t, o.active_variables, o.fk_genes, 2048, seed=17, kind="tracking")
        sol, fit, suc, steps = h.solve_batch(abi.default_solve_params(mode="jac", max_steps=32), seeds, params)
        assert suc.mean() > 0.9
        with pc.oracle_arithmetic(0):
            perr, rerr = pc.pose_errors(o, sol, params)
        assert perr[suc == 1].max() < POS_TOL and rerr[suc == 1].max() < ROT_TOL


def test_no_active_variable(pr2):
    """every joint of the group fixed: D = 0, the solve runs its budget and returns the seed, as the oracle does"""
    from bio_ik_amd import PoseGoal
    from bio_ik_amd.solver import HipSolver
    t0 = ProblemTemplate(pr2, "right_arm", [PoseGoal("r_wrist_roll_link")])
    names = [pr2.variable_names[v] for v in HipSolver(t0).active_variables]
    t = ProblemTemplate(pr2, "right_arm", [PoseGoal("r_wrist_roll_link")], fixed_joints=names)
    h, o = HipSolver(t), orc.Oracle(t)
    assert h.D == o.D == 0
    seeds, params = np.tile(pr2.default_positions(), (64, 1)), np.tile(t.pack_params(), (64, 1))
    for pop, fk in ((16, abi.FK_EXACT), (128, abi.FK_EXACT), (16, abi.FK_LINEAR)):
        p = abi.default_solve_params(population=pop, max_steps=2, random_seed=1, fk_mode=fk)
        got, want = h.solve_batch(p, seeds, params), o.solve_batch(p, orc.RNG_COUNTER, seeds, params)
        assert all(np.array_equal(a, b) for a, b in zip(got, want))
        assert np.array_equal(got[0], seeds) and not got[2].any()


def test_more_than_32_joints():
    """48 moving joints on one chain: function level and whole solves bit for bit, every lane mapping the launcher picks
    for 16 / 70 / 128 children per species; 64 active variables are refused"""
    from bio_ik_amd import AvoidJointLimitsGoal, PoseGoal, snake
    from bio_ik_amd.solver import BioIKError, HipSolver
    m = snake(48)
    t = ProblemTemplate(m, "snake", [PoseGoal("tip"), AvoidJointLimitsGoal()])
    h, o = HipSolver(t), orc.Oracle(t)
    assert h.D == o.D == 48
    pc.function_level(h, o, m, np.random.default_rng(11), n=300, exact_bits=True)
    pc.trajectory(h, o, t, n=8, pop=16, steps_list=(1, 3))
    pc.trajectory(h, o, t, n=4, pop=128, steps_list=(2,))
    pc.trajectory(h, o, t, n=4, pop=70, steps_list=(2,), fk_mode=abi.FK_LINEAR)
    with pytest.raises(BioIKError):
        HipSolver(ProblemTemplate(snake(64), "snake", [PoseGoal("tip")]))


@pytest.mark.parametrize("mid,with_base", [("planar", False), ("floating", False), ("planar", True)])
def test_floating_and_planar_joints_anywhere(mid, with_base):
    """a planar stage / a floating coupling in the MIDDLE of the chain, and two multi-variable joints on one chain (round 5: forward_kinematics.h:120-135,
    331-354 take them wherever they are).  As for the free base below: the forward-difference Jacobian columns go through acos / sqrt -- since round 6 the
    shared acos of bioik_acos.h and the IEEE sqrt: everything bit for bit, memetic solves included, plus the result level at 48 steps."""
    from bio_ik_amd import PoseGoal, PositionGoal
    from bio_ik_amd.solver import HipSolver
    from conftest import stage_robot
    m = stage_robot(mid, with_base)
    t = ProblemTemplate(m, "whole", [PoseGoal("tool"), PositionGoal("stage", weight=0.2)])
    h, o = HipSolver(t), orc.Oracle(t)
    assert h.D == o.D == 4 + (7 if mid == "floating" else 3) + (3 if with_base else 0)
    pc.function_level(h, o, m, np.random.default_rng(9), n=500, exact_bits=True)
    pc.trajectory(h, o, t, n=16, pop=128, steps_list=(1, 5), mode="bio2")
    pc.trajectory(h, o, t, n=8, pop=64, steps_list=(3,))
    t1 = ProblemTemplate(m, "whole", [PoseGoal("tool")])
    h1, o1 = HipSolver(t1), orc.Oracle(t1)
    seeds, params, _ = make_queries(t1, o1.active_variables, o1.fk_genes, 256, seed=22)
    p = abi.default_solve_params(population=64, max_steps=48, random_seed=3)
    sol, fit, suc, steps = h1.solve_batch(p, seeds, params)
    so = o1.solve_batch(p, orc.RNG_COUNTER, seeds, params, n_threads=8)
    assert suc.mean() >= so[2].mean() - 0.03 and suc.mean() > 0.7  # (the oracle's own rate on this fixture at 48 steps: 0.79 ... 0.9)
    perr, rerr = pc.pose_errors(o1, sol, params, tip=0, off=0)
    assert perr[suc == 1].max() < POS_TOL and rerr[suc == 1].max() < ROT_TOL


@pytest.mark.parametrize("base", ["floating", "planar"])
def test_floating_and_planar_joints(base):
    """a free base in front of the arm (forward_kinematics.h:120-135, 695-726; ik_evolution_2.cpp:203-215, 320-324).  The Jacobian
    columns of these joints come from a forward difference through acos / sqrt (frame.h:240-259) -- since round 6 the shared acos of
    bioik_acos.h and the IEEE sqrt: tables and solves bit for bit, memetic solves included, plus the result level at 48 steps."""
    from bio_ik_amd import PoseGoal, PositionGoal
    from bio_ik_amd.solver import HipSolver
    from conftest import mobile_robot
    m = mobile_robot(base)
    t = ProblemTemplate(m, "whole", [PoseGoal("tool"), PositionGoal("base", weight=0.2)])
    h, o = HipSolver(t), orc.Oracle(t)
    assert h.D == o.D == (10 if base == "floating" else 6)
    pc.function_level(h, o, m, np.random.default_rng(8), n=500, exact_bits=True)
    pc.trajectory(h, o, t, n=16, pop=128, steps_list=(1, 5), mode="bio2")
    pc.trajectory(h, o, t, n=8, pop=64, steps_list=(3,))
    # result level on the tool pose alone (the two-goal problem converges slowly: a low-weight goal against dtwist = 1e-5)
    t1 = ProblemTemplate(m, "whole", [PoseGoal("tool")])
    h1, o1 = HipSolver(t1), orc.Oracle(t1)
    seeds, params, _ = make_queries(t1, o1.active_variables, o1.fk_genes, 256, seed=21)
    p = abi.default_solve_params(population=64, max_steps=48, random_seed=3)
    sol, fit, suc, steps = h1.solve_batch(p, seeds, params)
    so = o1.solve_batch(p, orc.RNG_COUNTER, seeds, params, n_threads=8)
    assert suc.mean() >= so[2].mean() - 0.03 and suc.mean() > 0.9
    perr, rerr = pc.pose_errors(o1, sol, params, tip=0, off=0)
    assert perr[suc == 1].max() < POS_TOL and rerr[suc == 1].max() < ROT_TOL


def test_success_check_near_threshold(gpus, oracles, templates):
    pc.success_check_near_goal(gpus["c2"], oracles["c2"], templates["c2"], np.random.default_rng(4), n=64)


@pytest.mark.parametrize("cfg,pop,kw", [
    ("c2", 16, {}),
    ("c2", 16, {"fk_mode": abi.FK_LINEAR}),
    ("c2", 128, {}),
    ("c3", 128, {}),
    ("c4", 512, {}),
    ("c2", 130, {"mode": "bio2"}),
    ("c4", 16, {"mode": "bio2_memetic_l", "fk_mode": abi.FK_LINEAR}),
    ("c2", 16, {"islands": 4}),
    ("c2", 16, {"no_wipeout": 1, "dpos": 1e-4, "drot": 0.05, "dtwist": -1.0}),
])
def test_trajectory_bit_exact(gpus, oracles, templates, cfg, pop, kw):
    pc.trajectory(gpus[cfg], oracles[cfg], templates[cfg], n=24, pop=pop, steps_list=(1, 4, 12), **kw)


@pytest.mark.parametrize("env", [
    {"BIOIK_SOLVE_THREADS": "64"},                                        # one wavefront, species one after the other
    {"BIOIK_SOLVE_THREADS": "128"},                                       # one wavefront per species, concurrently
    {"BIOIK_SOLVE_THREADS": "256"},                                       # two wavefronts per species
    {"BIOIK_SOLVE_THREADS": "256", "BIOIK_SOLVE_SPECIES_PARALLEL": "0"},  # four wavefronts, species one after the other
    {"BIOIK_SOLVE_THREADS": "128", "BIOIK_SOLVE_STORE_CHILDREN": "0"},    # winners re-derived from the RNG
    {"BIOIK_SOLVE_THREADS": "64", "BIOIK_SOLVE_STORE_CHILDREN": "0"},
    {"BIOIK_SOLVE_THREADS": "128", "BIOIK_SOLVE_CHILD_PAIRS": "0"},       # one child per trip instead of two
    {"BIOIK_SOLVE_THREADS": "128", "BIOIK_SOLVE_GENERAL": "1"},           # the kernel flavour that also carries floating / planar joints
    {"BIOIK_SOLVE_THREADS": "64", "BIOIK_SOLVE_SPECIES_PARALLEL": "1"},   # one wavefront, the species on its two halves
    {"BIOIK_SOLVE_THREADS": "64", "BIOIK_SOLVE_SPECIES_PARALLEL": "1", "BIOIK_SOLVE_STORE_CHILDREN": "0"},
    {"BIOIK_SOLVE_THREADS": "128", "BIOIK_SOLVE_COLUMNLESS": "1"},        # children computed where they are read: no genotype columns in LDS
    {"BIOIK_SOLVE_THREADS": "64", "BIOIK_SOLVE_COLUMNLESS": "1"},
    {"BIOIK_SOLVE_THREADS": "128", "BIOIK_SOLVE_COLUMNLESS": "2"},       # ... and scored two at a time
    {"BIOIK_SOLVE_THREADS": "64", "BIOIK_SOLVE_SPECIES_PARALLEL": "1", "BIOIK_SOLVE_COLUMNLESS": "2"},  # C3: the joint walk of both species' children
    {"BIOIK_SOLVE_THREADS": "64", "BIOIK_SOLVE_SPECIES_PARALLEL": "1", "BIOIK_SOLVE_COLUMNLESS": "2", "BIOIK_SOLVE_NO_JOINT": "1"},
    {"BIOIK_SOLVE_THREADS": "256", "BIOIK_SOLVE_COLUMNLESS": "1"},
    {"BIOIK_SOLVE_THREADS": "128", "BIOIK_SOLVE_COLUMNLESS": "0", "BIOIK_SOLVE_STORE_CHILDREN": "0"},
    {"BIOIK_SOLVE_TWO_PHASE": "1"},                                      # two launches: hand-over after the first step (first launch: the species
    {"BIOIK_SOLVE_TWO_PHASE": "2"},                                      # on the halves of a wavefront where the problem allows, else the usual mapping)
    {"BIOIK_SOLVE_TWO_PHASE": "2", "BIOIK_SOLVE_THREADS": "256"},        # ... both launches under a forced mapping
    {"BIOIK_SOLVE_TWO_PHASE": "1", "BIOIK_SOLVE_GENERAL": "1"},
    {"BIOIK_SOLVE_TWO_PHASE": "1,2,4"},                                  # a chain of hand-overs (four launches)
])
def test_trajectory_independent_of_workgroup_mapping(gpus, oracles, templates, env, monkeypatch):
    """the same solve under every lane <-> work mapping the launcher can choose"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    pc.trajectory(gpus["c2"], oracles["c2"], templates["c2"], n=16, pop=128, steps_list=(6,))
    pc.trajectory(gpus["c3"], oracles["c3"], templates["c3"], n=8, pop=100, steps_list=(3,))
    pc.trajectory(gpus["c4"], oracles["c4"], templates["c4"], n=4, pop=70, steps_list=(2,), fk_mode=abi.FK_LINEAR)


def test_two_launch_solve_equals_single_launch(gpus, templates, monkeypatch):
    """A batch that fills the chip several times over is solved in two launches (first steps under the half-wavefront mapping, the unsolved
    queries handed to the mapping with the fastest lone step): bit for bit the result of one launch — full size, with islands, with a
    step budget below the hand-over, and under a wall-clock timeout (which ends queries at launch-dependent steps: only its guarantees hold)."""
    h, t = gpus["c2"], templates["c2"]
    seeds, params, _ = make_queries(t, h.active_variables, h.fk_genes, 4096, seed=21)
    for kw in ({"max_steps": 64}, {"max_steps": 24, "islands": 2}, {"max_steps": 6}, {"max_steps": 40, "mode": "bio2"}):
        p = abi.default_solve_params(population=128, random_seed=4, **kw)
        monkeypatch.setenv("BIOIK_SOLVE_TWO_PHASE", "0")
        one = h.solve_batch(p, seeds, params)
        monkeypatch.delenv("BIOIK_SOLVE_TWO_PHASE")
        two = h.solve_batch(p, seeds, params)  # the launcher's own choice for 4096 queries
        monkeypatch.setenv("BIOIK_SOLVE_TWO_PHASE", "13")
        late = h.solve_batch(p, seeds, params)
        monkeypatch.delenv("BIOIK_SOLVE_TWO_PHASE")
        assert all(np.array_equal(a, b) for a, b in zip(one, two)) and all(np.array_equal(a, b) for a, b in zip(one, late)), kw
    p = abi.default_solve_params(population=128, max_steps=4096, random_seed=4, timeout=0.004)
    sol, fit, suc, steps = h.solve_batch(p, seeds, params)
    assert steps.min() >= 1 and steps.max() < 4096 and suc.mean() > 0.5


def test_full_batch_c2_result_level(gpus, oracles, templates):
    """BASELINE.json configs[1]: 4096 PoseGoals, pop=128.  FK -> IK -> FK round trip (reference README.md:404-447)."""
    h, o, t = gpus["c2"], oracles["c2"], templates["c2"]
    n = 4096
    with pc.oracle_arithmetic(0):  # goal poses from the reference-pinned arithmetic, not from the device's own FK
        seeds, params, _ = make_queries(t, o.active_variables, o.fk_genes, n, seed=0xB101C)
    p = abi.default_solve_params(population=128, max_steps=64, random_seed=1)
    sol, fit, suc, steps = h.solve_batch(p, seeds, params)
    assert suc.mean() >= 0.99
    with pc.oracle_arithmetic(0):  # ... and the returned poses verified under it
        perr, rerr = pc.pose_errors(o, sol, params)
    assert perr[suc == 1].max() < POS_TOL and rerr[suc == 1].max() < ROT_TOL
    info = o.robot_info()
    bounded = info[:, 1] != np.finfo(float).max
    assert np.all(sol[:, bounded] >= info[bounded, 3] - 1e-12) and np.all(sol[:, bounded] <= info[bounded, 4] + 1e-12)
    inactive = np.setdiff1d(np.arange(h.V), h.active_variables)
    assert np.array_equal(sol[:, inactive], seeds[:, inactive])  # variables outside the group come back untouched
    # the oracle, given the same streams, returns the same answers (sample: CPU time)
    k = 96
    so = o.solve_batch(p, orc.RNG_COUNTER, seeds[:k], params[:k], n_threads=8)
    assert np.array_equal(so[0], sol[:k]) and np.array_equal(so[2], suc[:k]) and np.array_equal(so[3], steps[:k])
    # determinism / idempotence: a second launch returns identical bits
    sol2, fit2, suc2, steps2 = h.solve_batch(p, seeds, params)
    assert np.array_equal(sol, sol2) and np.array_equal(steps, steps2)


def test_full_batch_c3_c4_result_level(gpus, oracles, templates):
    """configs[2] (two tips + MinimalDisplacement) and configs[3] (31-DOF snake + AvoidJointLimits, pop=512)"""
    for cfg, pop, n, max_steps, min_rate in (("c3", 128, 4096, 128, 0.55), ("c4", 512, 4096, 64, 0.99)):  # (C3 on the device: 0.32 ... 0.34 within 64 steps, 0.60 ... 0.62 within 128, profiles/r05_c3_rates.log)
        h, o, t = gpus[cfg], oracles[cfg], templates[cfg]
        with pc.oracle_arithmetic(0):
            seeds, params, _ = make_queries(t, o.active_variables, o.fk_genes, n, seed=0xB101C)
        p = abi.default_solve_params(population=pop, max_steps=max_steps, random_seed=1)
        sol, fit, suc, steps = h.solve_batch(p, seeds, params)
        assert suc.mean() >= min_rate
        off = 0
        for tip in range(h.T):
            with pc.oracle_arithmetic(0):
                perr, rerr = pc.pose_errors(o, sol, params, tip=tip, off=off)
            assert perr[suc == 1].max() < POS_TOL and rerr[suc == 1].max() < ROT_TOL
            off += 8
        k = 8
        so = o.solve_batch(p, orc.RNG_COUNTER, seeds[:k], params[:k], n_threads=8)
        assert np.array_equal(so[0], sol[:k]) and np.array_equal(so[2], suc[:k])


def test_full_size_mixed_batch_on_one_gpu(gpus, oracles, templates):
    """BASELINE.json configs[4] on the one GPU of this box: a mixed batch of 8192 queries — 4096 PR2 right-arm (pop=128) and 4096
    31-DOF snake (pop=512) — sorted by model into two homogeneous blocks that bio_ik_amd.batch.solve_mixed runs concurrently on two
    HIP streams with all arrays resident in HBM.  Every block must come back bit-identical to its own solve."""
    import torch
    from bio_ik_amd.batch import solve_mixed
    blocks, want = [], []
    for cfg, pop, max_steps in (("c2", 128, 64), ("c4", 512, 32)):
        h, o, t = gpus[cfg], oracles[cfg], templates[cfg]
        with pc.oracle_arithmetic(0):
            seeds, params, _ = make_queries(t, o.active_variables, o.fk_genes, 4096, seed=0xB101C + pop)
        p = abi.default_solve_params(population=pop, max_steps=max_steps, random_seed=3)
        blocks.append((h, p, seeds, params))
        want.append(h.solve_batch(p, seeds, params))
    got = solve_mixed(blocks, device="cuda:0")
    torch.cuda.synchronize()
    for (h, p, seeds, params), g, w in zip(blocks, got, want):
        assert all(np.array_equal(a, b) for a, b in zip(g, w))
        assert g[2].mean() > 0.98
        with pc.oracle_arithmetic(0):
            perr, rerr = pc.pose_errors(oracles["c2" if h is gpus["c2"] else "c4"], g[0], params)
        assert perr[g[2] == 1].max() < POS_TOL and rerr[g[2] == 1].max() < ROT_TOL


def test_c5_full_size_on_one_gpu(gpus, oracles, templates):
    """BASELINE.json configs[4] at its full size on the one GPU of this box: 262 144 mixed queries (131 072 PR2 right-arm pop=128 +
    131 072 31-DOF snake pop=512) through solve_mixed.  Too many for the oracle, so the checks are the size-independent ones: a window of
    each block equals its own solve at the window's query offset (the sharding property, bit for bit), the success rates hold, and every
    success of a 4096-query sample reproduces its goal pose under the reference-pinned FK."""
    import torch
    from bio_ik_amd.batch import solve_mixed
    n, win, off = 131072, 2048, 77777
    blocks = []
    for cfg, pop, max_steps in (("c2", 128, 64), ("c4", 512, 32)):
        h, o, t = gpus[cfg], oracles[cfg], templates[cfg]
        with pc.oracle_arithmetic(0):
            seeds, params, _ = make_queries(t, o.active_variables, o.fk_genes, n, seed=0xC5 + pop)
        blocks.append((h, abi.default_solve_params(population=pop, max_steps=max_steps, random_seed=11), seeds, params))
    got = solve_mixed(blocks, device="cuda:0")
    torch.cuda.synchronize()
    for (h, p, seeds, params), g, cfg in zip(blocks, got, ("c2", "c4")):
        assert g[0].shape[0] == n and g[2].mean() > 0.98
        h.set_first_query(off)
        w = h.solve_batch(p, seeds[off:off + win], params[off:off + win])
        h.set_first_query(0)
        assert all(np.array_equal(a[off:off + win], b) for a, b in zip(g, w)), cfg
        idx = np.random.default_rng(5).choice(n, 4096, replace=False)
        with pc.oracle_arithmetic(0):
            perr, rerr = pc.pose_errors(oracles[cfg], g[0][idx], params[idx])
        ok = g[2][idx] == 1
        assert perr[ok].max() < POS_TOL and rerr[ok].max() < ROT_TOL


def test_solve_batch_multi_two_handles_on_one_gpu(gpus, templates):
    """the C-ABI form of the multi-GPU split (bioik_solve_batch_multi): on this one-GPU box two handles on device 0 take the two shards
    on their own host threads and streams; the result equals the single-handle solve bit for bit, and the caller's device is untouched"""
    import torch
    from bio_ik_amd.solver import HipSolver
    h, t = gpus["c2"], templates["c2"]
    other = HipSolver(t, device=0)
    seeds, params, _ = make_queries(t, h.active_variables, h.fk_genes, 4096, seed=77)
    p = abi.default_solve_params(population=128, max_steps=48, random_seed=6)
    h.set_first_query(12345)
    want = h.solve_batch(p, seeds, params)
    got = h.solve_batch_multi([other], p, seeds, params)
    h.set_first_query(0)
    assert all(np.array_equal(a, b) for a, b in zip(want, got))
    assert got[2].mean() > 0.98
    assert torch.cuda.current_device() == 0


def test_solve_batch_multi_on_distinct_devices(gpus, templates):
    """bioik_solve_batch_multi with one handle per GPU of the node (every visible device): the shards run on different devices and the
    result equals the single-device solve bit for bit.  Skipped on a one-GPU box (the two-handle test above covers the control flow)."""
    import torch
    from bio_ik_amd.solver import HipSolver, device_count
    nd = device_count()
    if nd < 2:
        pytest.skip("needs at least two HIP devices")
    h, t = gpus["c2"], templates["c2"]
    others = [HipSolver(t, device=d) for d in range(1, nd)]
    n = 4096 * nd
    seeds, params, _ = make_queries(t, h.active_variables, h.fk_genes, n, seed=78)
    p = abi.default_solve_params(population=128, max_steps=48, random_seed=6)
    h.set_first_query(777)
    want = h.solve_batch(p, seeds, params)
    got = h.solve_batch_multi(others, p, seeds, params)
    h.set_first_query(0)
    assert all(np.array_equal(a, b) for a, b in zip(want, got))
    assert torch.cuda.current_device() == 0


def test_submit_wait_pipelining_full_size(gpus, templates):
    """bioik_solve_batch_submit / _wait: five 4096-query batches through the handle's slots equal the synchronous solves bit for bit,
    whatever the order of waiting; the pipelined sequence is not slower than the one-at-a-time sequence"""
    import time
    h, t = gpus["c2"], templates["c2"]
    p = abi.default_solve_params(population=128, max_steps=64, random_seed=3)
    batches = [make_queries(t, h.active_variables, h.fk_genes, 4096, seed=900 + k)[:2] for k in range(5)]
    h.solve_batch(p, *batches[0])  # (warm-up: code object, arenas)
    t0 = time.perf_counter()
    want = [h.solve_batch(p, s, g) for s, g in batches]
    t_sync = time.perf_counter() - t0
    t0 = time.perf_counter()
    tickets = [h.submit_batch(p, s, g) for s, g in batches]
    got = {k: h.wait_batch(tickets[k]) for k in (4, 0, 2, 1, 3)}
    t_pipe = time.perf_counter() - t0
    for k in range(5):
        assert all(np.array_equal(a, b) for a, b in zip(want[k], got[k]))
    assert t_pipe < 1.05 * t_sync, (t_pipe, t_sync)
    print("five 4096-query batches: one at a time %.1f ms, pipelined %.1f ms" % (t_sync * 1e3, t_pipe * 1e3))


def test_selection_ties_are_decided_by_position(monkeypatch):
    """joints without any range: every child of a generation has the same fitness, the elitist selection is decided by position alone
    (ik_evolution_2.cpp:410-431) -- the tie path of the wavefront-minimum top-2 and of the merging butterfly, bit for bit against the oracle"""
    from bio_ik_amd import PoseGoal, snake
    from bio_ik_amd.solver import HipSolver
    t = ProblemTemplate(snake(4, limit=0.0), "snake", [PoseGoal("tip")])
    o = orc.Oracle(t)
    for env in ({"BIOIK_SOLVE_THREADS": "128"}, {"BIOIK_SOLVE_THREADS": "256"}, {"BIOIK_SOLVE_THREADS": "64", "BIOIK_SOLVE_SPECIES_PARALLEL": "1"}, {}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        pc.trajectory(HipSolver(t, device=0), o, t, n=3, pop=128 if env else 16, steps_list=(3,))
        for k in env:
            monkeypatch.delenv(k)


def test_preselection_by_selection_gives_what_the_sort_gives(gpus, oracles, templates, monkeypatch):
    """The pre-selection's survivors by selection of the k-th least key (select_threshold: the kernels with a wavefront per species, k_solve_lean_cl4 / cl4h) against the sort of all
    children (BIOIK_SOLVE_PRESELECT=0): the oracle's trajectories both ways, and whole batches of the two-armed problem and of the 31-joint chain bit for bit
    the same -- also with the children's fitness made coarse (BIOIK_SOLVE_TIE_TEST_BITS, a test switch), when most generations have several best children and
    the reference's stable order among them is what decides (ik_evolution_2.cpp:366-378, 410-423)"""
    for mode in ("0", "1"):
        monkeypatch.setenv("BIOIK_SOLVE_PRESELECT", mode)
        pc.trajectory(gpus["c4"], oracles["c4"], templates["c4"], n=4, pop=512, steps_list=(3,))
        pc.trajectory(gpus["c3"], oracles["c3"], templates["c3"], n=8, pop=128, steps_list=(4,))
        pc.trajectory(gpus["c3"], oracles["c3"], templates["c3"], n=300, pop=100, steps_list=(2,))
    for name, n, pop, steps, kw in (("c4", 2048, 512, 8, {}), ("c4", 3, 512, 6, {"mode": "bio2"}), ("c4", 700, 512, 10, {"mode": "bio2"}), ("c3", 4096, 128, 24, {})):
        h, t = gpus[name], templates[name]
        seeds, params, _ = make_queries(t, h.active_variables, h.fk_genes, n, seed=77)
        p = abi.default_solve_params(population=pop, max_steps=steps, random_seed=5, **kw)
        out = {}
        for bits in ("0", "44", "50"):
            monkeypatch.setenv("BIOIK_SOLVE_TIE_TEST_BITS", bits)
            for mode in ("0", "1"):
                monkeypatch.setenv("BIOIK_SOLVE_PRESELECT", mode)
                out[bits, mode] = h.solve_batch(p, seeds, params)
            assert all(np.array_equal(x, y) for x, y in zip(out[bits, "0"], out[bits, "1"])), (name, n, bits)
        if name == "c4":  # (the joint walk of the two-armed problem -- four keys per lane on half a wavefront -- keeps the sort: the cheaper of the two there)
            assert not np.array_equal(out["0", "1"][0], out["50", "1"][0])  # (the coarse values did take the search elsewhere)
    monkeypatch.delenv("BIOIK_SOLVE_TIE_TEST_BITS")
    monkeypatch.delenv("BIOIK_SOLVE_PRESELECT")


def test_secondary_goals_of_every_kind_in_whole_solves(templates):
    """Five secondary goals at once -- MinimalDisplacementGoal, AvoidJointLimitsGoal, CenterJointsGoal, a RegularizationGoal made secondary (sums over the joint
    values: the lanes of the line search share their terms, solve_body's secondary_shared) and a JointVariableGoal between them, which also puts its variable
    in front of the chains' so that the genes do NOT follow the ops (the sums then run in gene order, as the reference's do: goal_eval_joint_set_x) --: the
    oracle's trajectories bit for bit on the seven-joint arm and on both arms with the torso, and the same without the JointVariableGoal (the lean kernels)"""
    from bio_ik_amd import AvoidJointLimitsGoal, CenterJointsGoal, JointVariableGoal, MinimalDisplacementGoal, PoseGoal, RegularizationGoal
    from bio_ik_amd.solver import HipSolver
    model = templates["c2"].model
    reg = RegularizationGoal(weight=0.6)
    reg.secondary_ = True
    sec = [MinimalDisplacementGoal(weight=0.7), AvoidJointLimitsGoal(weight=0.3), JointVariableGoal("r_elbow_flex_joint", -1.0, weight=0.5, secondary=True),
           CenterJointsGoal(weight=0.2), reg]
    for goals in (sec, [g for g in sec if not isinstance(g, JointVariableGoal)]):
        t = ProblemTemplate(model, "right_arm", [PoseGoal("r_wrist_roll_link")] + goals)
        h, o = HipSolver(t), orc.Oracle(t)
        pc.function_level(h, o, model, np.random.default_rng(5), n=500, exact_bits=True)
        pc.trajectory(h, o, t, n=24, pop=16, steps_list=(6,))
        pc.trajectory(h, o, t, n=8, pop=128, steps_list=(4,))
        pc.trajectory(h, o, t, n=8, pop=40, steps_list=(5,), fk_mode=abi.FK_LINEAR)
        pc.trajectory(h, o, t, n=8, pop=16, steps_list=(6,), mode="bio2_memetic_l")
        t2 = ProblemTemplate(model, "all", [PoseGoal("r_wrist_roll_link"), PoseGoal("l_wrist_roll_link")] + goals)
        pc.trajectory(HipSolver(t2), orc.Oracle(t2), t2, n=16, pop=128, steps_list=(4,))


def test_goal_sets_beyond_one_goal_per_tip(templates):
    """parity_cases.goal_sets_beyond_one_goal_per_tip on the device"""
    from bio_ik_amd.solver import HipSolver
    pc.goal_sets_beyond_one_goal_per_tip(templates["c2"].model, lambda t: HipSolver(t))


def test_branching_hand():
    """parity_cases.branching_hand on the device"""
    from bio_ik_amd.solver import HipSolver
    pc.branching_hand(lambda t: HipSolver(t))


def test_exact_joint_program(templates, monkeypatch):
    """parity_cases.exact_joint_program on the device"""
    from bio_ik_amd.solver import HipSolver
    monkeypatch.setenv("BIOIK_COMPILE_EXACT", "1")
    pc.exact_joint_program(lambda t: HipSolver(t), templates)


def test_line_search_step_without_bound(monkeypatch):
    """parity_cases.line_search_step_without_bound (quirk Q7: the reference's candidate at +-DBL_MAX; no joint value of magnitude 1e300 leaves the product)"""
    from bio_ik_amd.solver import HipSolver
    monkeypatch.setenv("BIOIK_COMPILE_EXACT", "1")
    pc.line_search_step_without_bound(lambda t: HipSolver(t))


def test_line_search_on_a_flat_model(monkeypatch):
    """parity_cases.line_search_on_a_flat_model (quirk Q5: the reference's NaN candidate)"""
    from bio_ik_amd.solver import HipSolver
    monkeypatch.setenv("BIOIK_COMPILE_EXACT", "1")
    pc.line_search_on_a_flat_model(lambda t: HipSolver(t))


def test_four_wavefront_build_of_the_computed_children_kernel(gpus, oracles, templates, monkeypatch):
    """C4 at its full population runs under the 128-register build of the computed-children kernel (k_solve_lean_cl4: the launcher's
    residency rule); its trajectories equal the oracle's and those of the 168-register build bit for bit"""
    h, o, t = gpus["c4"], oracles["c4"], templates["c4"]
    pc.trajectory(h, o, t, n=3, pop=512, steps_list=(2,))
    seeds, params, _ = make_queries(t, h.active_variables, h.fk_genes, 256, seed=31)
    p = abi.default_solve_params(population=512, max_steps=6, random_seed=4)
    a = h.solve_batch(p, seeds, params)
    monkeypatch.setenv("BIOIK_SOLVE_THREE_WAVES", "1")
    b = h.solve_batch(p, seeds, params)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_joint_walk_of_both_species_children_full_size(gpus, oracles, templates, monkeypatch):
    """C3 at its full population runs k_solve_lean_clj4 (both species on one wavefront, their pre-selected children walked as one list, the
    128-register build: fitness values parked per species in LDS): trajectories equal to the oracle's, and a 1024-query batch equal to the
    three-wavefront build of the same walk (k_solve_lean_clj) and to the one-half-per-species form bit for bit"""
    h, o, t = gpus["c3"], oracles["c3"], templates["c3"]
    seeds, params, _ = make_queries(t, h.active_variables, h.fk_genes, 1024, seed=33)
    p = abi.default_solve_params(population=128, max_steps=5, random_seed=6)
    a = h.solve_batch(p, seeds, params)
    with pc.oracle_arithmetic(1):
        w = o.solve_batch(p, orc.RNG_COUNTER, seeds[:24], params[:24], n_threads=8)
    assert all(np.array_equal(x[:24], y) for x, y in zip(a, w))
    monkeypatch.setenv("BIOIK_SOLVE_THREE_WAVES", "1")
    b = h.solve_batch(p, seeds, params)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    monkeypatch.delenv("BIOIK_SOLVE_THREE_WAVES")
    monkeypatch.setenv("BIOIK_SOLVE_NO_JOINT", "1")
    b = h.solve_batch(p, seeds, params)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    for pop in (9, 33, 70, 200):  # (lists shorter than a half-wavefront, over several trips, odd tails)
        monkeypatch.delenv("BIOIK_SOLVE_NO_JOINT", raising=False)
        pc.trajectory(h, o, t, n=4, pop=pop, steps_list=(3,))


def test_throughput_schedule_changes_no_result(gpus, oracles, templates, monkeypatch):
    """bioik_solve_params::schedule = BIOIK_SCHEDULE_THROUGHPUT: a full-size batch solved in one launch with both species of a query on one
    wavefront (k_solve_lean_cl64w4: the 128-register build; BIOIK_SOLVE_THREE_WAVES: the 168-register one) -- the oracle's trajectories, and the
    results of the default (latency) schedule bit for bit"""
    h, o, t = gpus["c2"], oracles["c2"], templates["c2"]
    pc.trajectory(h, o, t, n=16, pop=128, steps_list=(1, 6), schedule=abi.SCHEDULE_THROUGHPUT)
    seeds, params, _ = make_queries(t, h.active_variables, h.fk_genes, 4096, seed=35)
    a = h.solve_batch(abi.default_solve_params(population=128, max_steps=64, random_seed=2), seeds, params)
    b = h.solve_batch(abi.default_solve_params(population=128, max_steps=64, random_seed=2, schedule="throughput"), seeds, params)
    assert all(np.array_equal(x, y) for x, y in zip(a, b)) and a[2].mean() > 0.99
    monkeypatch.setenv("BIOIK_SOLVE_THREE_WAVES", "1")
    c = h.solve_batch(abi.default_solve_params(population=128, max_steps=64, random_seed=2, schedule="throughput"), seeds, params)
    assert all(np.array_equal(x, y) for x, y in zip(a, c))


def test_helper_wavefronts(gpus, oracles, templates, monkeypatch):
    """k_solve_lean_cl4h on the device: k_solve_lean_cl4's two wavefronts plus two helpers that walk half of every generation's children, handing over through
    words in LDS -- the oracle's trajectories; small launches and the stragglers of a chip-filling call equal to what k_solve_lean_cl4 itself returns
    (BIOIK_SOLVE_HELPED=0), bit for bit"""
    h, o, t = gpus["c2"], oracles["c2"], templates["c2"]
    pc.trajectory(h, o, t, n=16, pop=128, steps_list=(1, 6))
    pc.trajectory(h, o, t, n=8, pop=200, steps_list=(4,), islands=2, island_sync=1)
    for n, kw in ((1, {}), (300, {}), (64, {"islands": 16, "island_sync": 1}), (1024, {}), (4096, {})):
        seeds, params, _ = make_queries(t, h.active_variables, h.fk_genes, n, seed=61)
        p = abi.default_solve_params(population=128, max_steps=64, random_seed=7, **kw)
        monkeypatch.setenv("BIOIK_SOLVE_HELPED", "0")
        a = h.solve_batch(p, seeds, params)
        monkeypatch.setenv("BIOIK_SOLVE_HELPED", "1024")
        b = h.solve_batch(p, seeds, params)
        assert all(np.array_equal(x, y) for x, y in zip(a, b)), n
    monkeypatch.delenv("BIOIK_SOLVE_HELPED")


def test_measured_mapping_choice_changes_no_result(templates, monkeypatch):
    """bioik_hip.hip: solve_dispatch -- a handle's first chip-filling call under the latency schedule runs once per eligible lane mapping and keeps the fastest;
    whatever it keeps, the answers are those of the rules alone (BIOIK_SOLVE_AUTOTUNE=0), for a problem of BASELINE.json and for one outside every fitted
    threshold (a 12-joint chain, 64 children per species); the device-pointer entry then uses the handle's choice without waiting"""
    from bio_ik_amd import PoseGoal, snake
    from bio_ik_amd.solver import HipSolver
    for t, pop in ((templates["c2"], 128), (ProblemTemplate(snake(12), "snake", [PoseGoal("tip")]), 64)):
        p = abi.default_solve_params(population=pop, max_steps=32, random_seed=4)
        monkeypatch.setenv("BIOIK_SOLVE_AUTOTUNE", "0")
        h0 = HipSolver(t)
        seeds, params, _ = make_queries(t, h0.active_variables, h0.fk_genes, 4096, seed=51)
        a = h0.solve_batch(p, seeds, params)
        monkeypatch.setenv("BIOIK_SOLVE_AUTOTUNE", "1")
        h1 = HipSolver(t)
        b = h1.solve_batch(p, seeds, params)   # (times the presets)
        c = h1.solve_batch(p, seeds, params)   # (runs the one it kept)
        assert all(np.array_equal(x, y) for x, y in zip(a, b)) and all(np.array_equal(x, y) for x, y in zip(a, c))
        monkeypatch.delenv("BIOIK_SOLVE_AUTOTUNE")


def test_stragglers_that_leave_when_the_chip_runs_empty(gpus, oracles, templates, monkeypatch):
    """SolveArgs::resident on the device: a chip-filling call of the latency schedule starts under the dense kernel and hands its stragglers to
    k_solve_lean_cl4 when fewer than BIOIK_SOLVE_DRAIN_BELOW wavefronts are left -- which query leaves at which step is a matter of timing, the results
    are not: equal to the one-kernel solve (threshold 0) bit for bit, for thresholds that hand over nearly everything and nearly nothing, with and
    without islands; and the test pattern of the host simulator's suite (every unit at its own step) against the oracle"""
    h, o, t = gpus["c2"], oracles["c2"], templates["c2"]
    seeds, params, _ = make_queries(t, h.active_variables, h.fk_genes, 4096, seed=41)
    p = abi.default_solve_params(population=128, max_steps=64, random_seed=9)
    monkeypatch.setenv("BIOIK_SOLVE_DRAIN_BELOW", "0")
    a = h.solve_batch(p, seeds, params)
    for below in ("1024", "100000", "16"):
        monkeypatch.setenv("BIOIK_SOLVE_DRAIN_BELOW", below)
        b = h.solve_batch(p, seeds, params)
        assert all(np.array_equal(x, y) for x, y in zip(a, b)), below
    p2 = abi.default_solve_params(population=128, max_steps=24, random_seed=9, islands=2, island_sync=1)
    monkeypatch.setenv("BIOIK_SOLVE_DRAIN_BELOW", "0")
    a = h.solve_batch(p2, seeds[:2048], params[:2048])
    monkeypatch.setenv("BIOIK_SOLVE_DRAIN_BELOW", "2048")
    b = h.solve_batch(p2, seeds[:2048], params[:2048])
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    monkeypatch.delenv("BIOIK_SOLVE_DRAIN_BELOW")
    monkeypatch.setenv("BIOIK_SOLVE_DRAIN_TEST", "6")
    pc.trajectory(h, o, t, n=48, pop=128, steps_list=(10,), schedule=abi.SCHEDULE_THROUGHPUT)
    pc.trajectory(gpus["c3"], oracles["c3"], templates["c3"], n=16, pop=128, steps_list=(8,))


def _graph_fixture(h, t, n, seed):
    import torch
    dev = torch.device("cuda", 0)
    seeds, params, _ = make_queries(t, h.active_variables, h.fk_genes, n, seed=seed)
    ds, dp = torch.from_numpy(seeds).to(dev), torch.from_numpy(params).to(dev)
    o = (torch.empty((n, h.V), dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.float64, device=dev),
         torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev))
    return seeds, params, ds, dp, o


def test_hipgraph_capture_of_a_chip_filling_call(gpus, templates, monkeypatch):
    """bioik_solve_batch_device enqueues on the caller's stream without synchronising: a chip-filling call of the latency schedule -- TWO launches (the dense
    kernel, then the stragglers under k_solve_lean_cl4 when the chip runs empty), the hand-over list in the handle's scratch, the resident words -- is
    captured into a hipGraph and replayed FOUR times, the third and fourth on new queries written into the captured input arrays; every replay equals the
    eager solve of the same queries bit for bit.  (Until round 4 the second replay of such a graph went wrong: the runtime's memset NODE in front of the
    kernels, see bioik_hip.hip: be_fill_async; the words a solve resets are now written by a kernel of the library.)  The same with a hand-over after a
    fixed step (BIOIK_SOLVE_TWO_PHASE), which hands nearly every unit over."""
    import torch
    h, t = gpus["c2"], templates["c2"]
    n = 4096
    p = abi.default_solve_params(population=128, max_steps=64, random_seed=3)
    for two_phase in (None, "1"):
        if two_phase:
            monkeypatch.setenv("BIOIK_SOLVE_TWO_PHASE", two_phase)
        seeds, params, ds, dp, o = _graph_fixture(h, t, n, 43)
        seeds2, params2, _ = make_queries(t, h.active_variables, h.fk_genes, n, seed=44)
        ref, ref2 = h.solve_batch(p, seeds, params), h.solve_batch(p, seeds2, params2)
        s = torch.cuda.Stream(torch.device("cuda", 0))

        def enqueue():
            h.solve_batch_device(p, n, ds.data_ptr(), dp.data_ptr(), o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), s.cuda_stream)
        with torch.cuda.stream(s):
            enqueue()  # (warm: the scratch of the first call on this stream is allocated outside the capture)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            enqueue()
        for i in range(4):
            want = ref if i < 2 else ref2
            if i == 2:
                ds.copy_(torch.from_numpy(seeds2)), dp.copy_(torch.from_numpy(params2))
            o[0].zero_(), o[2].zero_(), o[3].zero_()
            g.replay()
            torch.cuda.synchronize()
            assert np.array_equal(o[0].cpu().numpy(), want[0]) and np.array_equal(o[2].cpu().numpy(), want[2]) and np.array_equal(o[3].cpu().numpy(), want[3]), (two_phase, i)
        del g
        monkeypatch.delenv("BIOIK_SOLVE_TWO_PHASE", raising=False)


def test_hipgraph_replay_with_islands_timeout_and_eager_solves_between(gpus, templates):
    """What else a captured solve depends on, replayed: the islands' first-success words (island_sync; a 0xff fill per replay), the launch clock of a
    timeout (zeroed per replay), k_select behind the solve kernels, and the scratch the graph's addresses point into -- an eager solve on the SAME handle
    and stream that needs MORE scratch (more queries, more islands) runs between the replays and must not take the graph's buffer away (a buffer a
    capture has used is pinned until the handle goes; bioik_hip.hip: bioik_problem::Scratch)."""
    import torch
    h, t = gpus["c2"], templates["c2"]
    n = 1536
    p = abi.default_solve_params(population=128, max_steps=48, random_seed=5, islands=2, island_sync=1, timeout=3600.0)
    seeds, params, ds, dp, o = _graph_fixture(h, t, n, 47)
    ref = h.solve_batch(p, seeds, params)
    big_n = 3000
    pb = abi.default_solve_params(population=128, max_steps=48, random_seed=6, islands=4, island_sync=1)
    bseeds, bparams, bds, bdp, bo = _graph_fixture(h, t, big_n, 48)
    bref = h.solve_batch(pb, bseeds, bparams)
    s = torch.cuda.Stream(torch.device("cuda", 0))

    def enqueue():
        h.solve_batch_device(p, n, ds.data_ptr(), dp.data_ptr(), o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), s.cuda_stream)

    def eager_big():
        with torch.cuda.stream(s):
            h.solve_batch_device(pb, big_n, bds.data_ptr(), bdp.data_ptr(), bo[0].data_ptr(), bo[1].data_ptr(), bo[2].data_ptr(), bo[3].data_ptr(), s.cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(bo[0].cpu().numpy(), bref[0]) and np.array_equal(bo[3].cpu().numpy(), bref[3])
    with torch.cuda.stream(s):
        enqueue()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        enqueue()
    for i in range(4):
        o[0].zero_(), o[2].zero_(), o[3].zero_()
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(o[0].cpu().numpy(), ref[0]) and np.array_equal(o[2].cpu().numpy(), ref[2]) and np.array_equal(o[3].cpu().numpy(), ref[3]), i
        if i < 3:
            eager_big()  # (needs more scratch than the graph's solve: grows on a buffer of its own)
    # round 6: a SECOND capture on the same stream, after eager calls have moved to a buffer of their own, takes the first capture's pinned buffer again
    # (the eager one stays free to grow): both graphs replay to the eager answer, with eager solves in between
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=s):
        enqueue()
    for gg in (g2, g, g2):
        o[0].zero_(), o[2].zero_(), o[3].zero_()
        gg.replay()
        torch.cuda.synchronize()
        assert np.array_equal(o[0].cpu().numpy(), ref[0]) and np.array_equal(o[2].cpu().numpy(), ref[2]) and np.array_equal(o[3].cpu().numpy(), ref[3])
        eager_big()


def test_best_island_three_ways(gpus, templates, monkeypatch):
    """round 6: the islands' reduction by a wavefront (inside the solve's launch, or a launch of its own) against the lane that walks the islands"""
    pc.island_selection_three_ways(gpus["c2"], templates["c2"], monkeypatch, n=5, pop=128, steps=12)
    pc.island_selection_three_ways(gpus["c2"], templates["c2"], monkeypatch, n=3, pop=16, steps=20, fk_mode=abi.FK_LINEAR)


def test_islands_that_stop_each_other(gpus, oracles, templates, monkeypatch):
    """bioik_solve_params::island_sync = 1 on the device: islands of a query that run in different workgroups at different times, the answer still the
    oracle's lock-step answer bit for bit (an island leaves early only when its result can no longer be chosen)"""
    pc.trajectory(gpus["c2"], oracles["c2"], templates["c2"], n=64, pop=128, steps_list=(24,), islands=4, island_sync=1, seed=3)
    pc.trajectory(gpus["c2"], oracles["c2"], templates["c2"], n=40, pop=16, steps_list=(40,), islands=3, island_sync=1, fk_mode=abi.FK_LINEAR, seed=4)
    pc.trajectory(gpus["c3"], oracles["c3"], templates["c3"], n=16, pop=128, steps_list=(8,), islands=2, island_sync=1)
    pc.trajectory(gpus["c2"], oracles["c2"], templates["c2"], n=16, pop=8, steps_list=(40,), islands=8, island_sync=1, mode="gd_c")
    pc.trajectory(gpus["c2"], oracles["c2"], templates["c2"], n=32, pop=128, steps_list=(24,), islands=2, island_sync=1, schedule="throughput")
    monkeypatch.setenv("BIOIK_SOLVE_TWO_PHASE", "2,5")
    pc.trajectory(gpus["c2"], oracles["c2"], templates["c2"], n=64, pop=128, steps_list=(24,), islands=4, island_sync=1, seed=3)
    monkeypatch.delenv("BIOIK_SOLVE_TWO_PHASE")
    # a full-size batch: fewer steps than with independent islands, the same successes, and the same answer twice
    h, t = gpus["c2"], templates["c2"]
    seeds, params, _ = make_queries(t, h.active_variables, h.fk_genes, 2048, seed=37)
    a = h.solve_batch(abi.default_solve_params(population=128, max_steps=48, random_seed=2, islands=4), seeds, params)
    b = h.solve_batch(abi.default_solve_params(population=128, max_steps=48, random_seed=2, islands=4, island_sync=1), seeds, params)
    c = h.solve_batch(abi.default_solve_params(population=128, max_steps=48, random_seed=2, islands=4, island_sync=1), seeds, params)
    assert np.array_equal(a[2], b[2]) and np.all(b[3] <= a[3]) and b[3].mean() < a[3].mean()
    assert all(np.array_equal(x, y) for x, y in zip(b, c))
    # islands = BIOIK_ISLANDS_AUTO: sized to the part of the chip the call leaves idle (min(16, 2048 / n), at least four up to 1024 queries, one beyond), stopping each other
    # (round 6: 64 islands up to eight queries, 32 up to sixteen -- MoveIt's one pose per call)
    for n, want in ((1, 64), (16, 32), (200, 10), (700, 4), (1024, 4), (1025, 1), (2048, 1)):
        d = h.solve_batch(abi.default_solve_params(population=128, max_steps=48, random_seed=2, islands=abi.ISLANDS_AUTO), seeds[:n], params[:n])
        e = h.solve_batch(abi.default_solve_params(population=128, max_steps=48, random_seed=2, islands=want, island_sync=1), seeds[:n], params[:n])
        assert all(np.array_equal(x, y) for x, y in zip(d, e)), n


def test_sharded_batch_equals_whole_batch(gpus, templates):
    """the multi-GPU split: shards solved separately with their query offsets reproduce the unsharded batch"""
    h, t = gpus["c2"], templates["c2"]
    n = 512
    seeds, params, _ = make_queries(t, h.active_variables, h.fk_genes, n, seed=5)
    p = abi.default_solve_params(population=128, max_steps=32, random_seed=9)
    h.set_first_query(0)
    whole = h.solve_batch(p, seeds, params)
    parts = []
    for r in range(4):
        h.set_first_query(r * 128)
        parts.append(h.solve_batch(p, seeds[r * 128:(r + 1) * 128], params[r * 128:(r + 1) * 128]))
    h.set_first_query(0)
    for i in range(4):
        assert np.array_equal(np.concatenate([q[i] for q in parts]), whole[i])


def test_device_pointer_entry_and_streamed_fitness(gpus, oracles, templates):
    """bioik_solve_batch_device / bioik_stream_fitness_device on arrays resident in HBM (torch only supplies memory)"""
    torch = pytest.importorskip("torch")
    h, o, t = gpus["c2"], oracles["c2"], templates["c2"]
    dev = torch.device("cuda", 0)
    n = 256
    seeds, params, _ = make_queries(t, h.active_variables, h.fk_genes, n, seed=6)
    p = abi.default_solve_params(population=128, max_steps=16, random_seed=2)
    ref = h.solve_batch(p, seeds, params)
    ds, dp = torch.from_numpy(seeds).to(dev), torch.from_numpy(params).to(dev)
    sol = torch.empty((n, h.V), dtype=torch.float64, device=dev)
    fit = torch.empty(n, dtype=torch.float64, device=dev)
    suc = torch.empty(n, dtype=torch.int32, device=dev)
    steps = torch.empty(n, dtype=torch.int32, device=dev)
    st = torch.cuda.Stream(dev)
    with torch.cuda.stream(st):
        h.solve_batch_device(p, n, ds.data_ptr(), dp.data_ptr(), sol.data_ptr(), fit.data_ptr(), suc.data_ptr(), steps.data_ptr(), st.cuda_stream)
    st.synchronize()
    assert np.array_equal(sol.cpu().numpy(), ref[0]) and np.array_equal(steps.cpu().numpy(), ref[3])
    # streamed generation: genes [unit][D][pop] in HBM -> fitness [unit][pop], against the oracle
    # (odd tails, several blocks per unit, a tree with parked branch frames)
    for cfg, pop, units in (("c2", 128, 8), ("c2", 77, 5), ("c2", 600, 3), ("c3", 130, 4)):
        h, o, t = gpus[cfg], oracles[cfg], templates[cfg]
        seeds, params, _ = make_queries(t, h.active_variables, h.fk_genes, units, seed=6)
        ds, dp = torch.from_numpy(seeds).to(dev), torch.from_numpy(params).to(dev)
        rng = np.random.default_rng(pop)
        genes = rng.uniform(-1, 1, size=(units, h.D, pop))
        dg = torch.from_numpy(genes).to(dev)
        df = torch.empty((units, pop), dtype=torch.float64, device=dev)
        h.stream_fitness_device(units, pop, ds.data_ptr(), dp.data_ptr(), dg.data_ptr(), df.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        torch.cuda.synchronize(dev)
        got = df.cpu().numpy()
        for u in range(units):
            want, _ = o.fitness(abi.FK_EXACT, seeds[u], params[u], genes[u].T)
            assert np.array_equal(got[u], want), (cfg, pop, u)


def test_two_launches_in_flight(gpus, templates):
    """one problem handle, two HIP streams, launches in flight together (islands > 1 use stream-ordered scratch): every launch
    returns what a lone launch returns"""
    import torch
    h, t = gpus["c2"], templates["c2"]
    dev = torch.device("cuda", 0)
    n = 512
    seeds, params, _ = make_queries(t, h.active_variables, h.fk_genes, n, seed=77)
    for islands in (1, 2):
        p = abi.default_solve_params(population=64, max_steps=16, random_seed=5, islands=islands)
        ref = h.solve_batch(p, seeds, params)
        ds, dp = torch.from_numpy(seeds).to(dev), torch.from_numpy(params).to(dev)
        streams = [torch.cuda.Stream(dev) for _ in range(2)]
        outs = [(torch.empty((n, h.V), dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.float64, device=dev),
                 torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev)) for _ in range(4)]
        torch.cuda.synchronize(dev)
        for i, o in enumerate(outs):
            h.solve_batch_device(p, n, ds.data_ptr(), dp.data_ptr(), o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(),
                                 streams[i % 2].cuda_stream)
        torch.cuda.synchronize(dev)
        for o in outs:
            assert np.array_equal(o[0].cpu().numpy(), ref[0]) and np.array_equal(o[2].cpu().numpy(), ref[2]) and np.array_equal(o[3].cpu().numpy(), ref[3])


def test_wall_clock_timeout(gpus, templates):
    """the caller's timeout (reference ik_parallel.h:160, kinematics_plugin.cpp:504): wall-clock budget of the call on the device
    clock; at least one step per query; a generous timeout changes no bit of the result"""
    import time
    h, t = gpus["c2"], templates["c2"]
    n = 2048
    seeds, params, _ = make_queries(t, h.active_variables, h.fk_genes, n, seed=123)
    far = params.copy()
    far[:, :3] += 10.0  # unreachable: no query can succeed, only a budget ends the call
    h.solve_batch(abi.default_solve_params(population=128, max_steps=2, random_seed=1), seeds, far)  # (first launch of the process)
    t0 = time.perf_counter()
    sol, fit, suc, steps = h.solve_batch(abi.default_solve_params(population=128, max_steps=1000000, random_seed=1, timeout=0.02), seeds, far)
    dt = time.perf_counter() - t0
    assert not suc.any() and steps.min() >= 1
    assert 0.015 < dt < 0.2, dt  # ~20 ms on the device + whatever the last single steps and the copies take; a million steps would take minutes
    assert steps.max() > 20      # the workgroups that started first used their time
    p0 = abi.default_solve_params(population=128, max_steps=24, random_seed=4)
    p1 = abi.default_solve_params(population=128, max_steps=24, random_seed=4, timeout=3600.0)
    a, b = h.solve_batch(p0, seeds, params), h.solve_batch(p1, seeds, params)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    # the timeout counts from the call's SUBMISSION (ik_parallel.h:160, 200), not from the moment its launch reaches the chip: three chip-filling solves
    # queued on one stream at (nearly) the same time -- each is past its deadline when the one before it ends, runs its one obligatory step and leaves
    import torch
    dev = torch.device("cuda", 0)
    ds, dp = torch.from_numpy(seeds).to(dev), torch.from_numpy(far).to(dev)
    outs = [(torch.empty((n, h.V), dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.float64, device=dev),
             torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev)) for _ in range(3)]
    s = torch.cuda.Stream(dev)
    pt = abi.default_solve_params(population=128, max_steps=1000000, random_seed=1, timeout=0.02)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    h.solve_batch_device(abi.default_solve_params(population=128, max_steps=2, random_seed=1), n, ds.data_ptr(), dp.data_ptr(), *[x.data_ptr() for x in outs[0]], s.cuda_stream)
    s.synchronize()
    with torch.cuda.stream(s):
        ev[0].record(s)
        for k in range(3):
            h.solve_batch_device(pt, n, ds.data_ptr(), dp.data_ptr(), *[x.data_ptr() for x in outs[k]], s.cuda_stream)
            ev[k + 1].record(s)
    s.synchronize()
    ends = [ev[0].elapsed_time(ev[k + 1]) for k in range(3)]  # ms from the first submission to the end of solve k
    assert 15.0 < ends[0] < 40.0, ends
    assert ends[2] < ends[0] + 15.0, ends          # (60 ms and more if every solve's clock started with its launch)
    for k in range(3):
        assert outs[k][3].min().item() >= 1 and not outs[k][2].any().item()
    assert outs[2][3].max().item() <= 4            # the third solve: the obligatory step (and what fits before the verdict crosses the workgroup)


def test_error_conventions(pr2):
    """status codes instead of exceptions/aborts (include/bioik_hip.h)"""
    from bio_ik_amd import PoseGoal, RobotModel
    from bio_ik_amd.solver import BioIKError, HipSolver
    m = RobotModel("float")  # a planar joint that mimics another joint: what the device still has no form for (a floating joint behind a moving joint runs since round 5)
    m.add_link("base")
    m.add_link("turret", "base", "yaw", "revolute", axis=(0, 0, 1), lower=-1.0, upper=1.0, velocity=1.0)
    m.add_link("body", "turret", "fj", "floating")
    m.add_link("sled", "body", "pj", "planar", mimic=("yaw", 1.0, 0.0))
    m.add_group("g", joints=["yaw", "fj", "pj"], tips=["sled"])
    with pytest.raises(BioIKError) as e:
        HipSolver(ProblemTemplate(m, "g", [PoseGoal("sled")]))
    assert e.value.code == abi.ERR_UNSUPPORTED
    m.add_group("g2", joints=["yaw", "fj"], tips=["body"])
    m._keep = None
    assert HipSolver(ProblemTemplate(m, "g2", [PoseGoal("body")])).D == 8
    with pytest.raises(BioIKError) as e:
        HipSolver(ProblemTemplate(pr2, "right_arm", [PoseGoal("r_wrist_roll_link")]), device=99)
    assert e.value.code == abi.ERR_NO_DEVICE
    from bio_ik_amd import JointVariableGoal
    with pytest.raises(BioIKError) as e:  # variable outside the group: reference ERROR("joint variable not found"), problem.cpp:125
        HipSolver(ProblemTemplate(pr2, "right_arm", [PoseGoal("r_wrist_roll_link"), JointVariableGoal("l_elbow_flex_joint", 0.0)]))
    assert e.value.code == abi.ERR_NOT_FOUND


@pytest.mark.gpu
def test_sharded_and_mixed_batch_device_path(tmp_path):
    """bio_ik_amd/batch.py with the exchange buffers in HBM (backend nccl = RCCL, one rank on this one-GPU box): the shard is
    solved by bioik_solve_batch_device on its own stream and gathered; equal, bit for bit, to the host-pointer solve."""
    import subprocess
    import sys
    out = str(tmp_path / "result.npy")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", BIOIK_WORKER_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", "29641",
           os.path.join(ROOT, "tests", "_gloo_worker.py"), out]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=150)
    except subprocess.TimeoutExpired:
        pytest.skip("the one-rank RCCL rendezvous did not complete within 150 s on this box")
    assert r.returncode == 0, r.stderr[-2000:]
    res = np.load(out)
    assert res[0] == 1 and res[1] == 1
