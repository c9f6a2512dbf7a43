"""CPU-side parity of the KERNEL BODIES (bio_ik_amd/csrc built with -DBIOIK_HOSTSIM, every lane a fibre of the calling thread) against
the oracle.  Same source as the gfx950 kernels, so the solver logic (selection order, RNG contexts, memetic phase,
species management, pre-selection by secondary goals, multi-wave reductions) is stepped against the reference
restatement on a machine without a GPU.  The GPU suite (test_gpu_parity.py) repeats these cases through
libbioik_hip.so at larger sizes."""
import os

import numpy as np
import pytest

import parity_cases as pc
from bio_ik_amd import PoseGoal, ProblemTemplate, abi
from bio_ik_amd.solver import HipSolver
from bio_ik_amd.workload import make_queries
from conftest import gnarly_goals
from oracle import orc


@pytest.fixture(scope="module", autouse=True)
def shared_trigonometry():
    """bit-exact comparisons need the oracle on the sincos it shares with the device (oracle/orc_model.h)"""
    orc.set_trig_mode(1)
    yield
    orc.set_trig_mode(0)


@pytest.fixture(autouse=True)
def no_divergent_collectives(hostsim_lib, request):
    """Every collective of the simulator (wavefront rendezvous, shuffles, ballots, the workgroup barrier) carries the source line it is called from;
    lanes that meet at DIFFERENT collectives -- a silent exchange of garbage on the device -- are counted.  No test may cause one."""
    import ctypes
    count = hostsim_lib.hostsim_divergent_collectives
    count.restype = ctypes.c_ulonglong
    before = count()
    yield
    if "divergence_selftest" not in request.node.name:
        assert count() == before, "lanes of a wavefront met at different collectives (see the [hostsim] lines on stderr)"


def test_divergence_selftest(hostsim_lib):
    """the detector itself: a launch whose odd lanes synchronise from another line than its even ones is reported, the same launch without is not"""
    import ctypes
    count = hostsim_lib.hostsim_divergent_collectives
    count.restype = ctypes.c_ulonglong
    before = count()
    hostsim_lib.hostsim_selftest_divergence(0)
    assert count() == before
    hostsim_lib.hostsim_selftest_divergence(1)
    assert count() == before + 32


@pytest.fixture(scope="module")
def sims(hostsim_lib, templates, oracles):
    return {k: HipSolver(t, lib=hostsim_lib) for k, t in templates.items()}


@pytest.mark.parametrize("cfg", ["c2", "c3", "c4"])
def test_function_level_bit_exact(sims, oracles, templates, cfg):
    """the fixtures have axis-aligned joint origins, for which folding fixed links into the joint program is exact"""
    pc.function_level(sims[cfg], oracles[cfg], templates[cfg].model, np.random.default_rng(1), n=70, exact_bits=True)


def test_function_level_gnarly(hostsim_lib, gnarly):
    """rotated origins, oblique axes, prismatic joint, branches with parked frames, root tip, all 16 goal opcodes"""
    t = ProblemTemplate(gnarly, "body", gnarly_goals())
    h, o = HipSolver(t, lib=hostsim_lib), orc.Oracle(t)
    pc.function_level(h, o, gnarly, np.random.default_rng(2), n=70)
    t2 = ProblemTemplate(gnarly, "body", gnarly_goals(), fixed_joints=["lift_joint", "antenna_joint"])
    h2, o2 = HipSolver(t2, lib=hostsim_lib), orc.Oracle(t2)
    assert h2.D == 8
    pc.function_level(h2, o2, gnarly, np.random.default_rng(3), n=70)


def test_success_check_near_threshold(sims, oracles, templates):
    pc.success_check_near_goal(sims["c2"], oracles["c2"], templates["c2"], np.random.default_rng(4), n=16)


@pytest.mark.parametrize("cfg,pop,kw", [
    ("c2", 16, {}),                                # one wavefront, exact FK per individual, memetic 'q'
    ("c2", 16, {"fk_mode": abi.FK_LINEAR}),        # the reference's linearised phenotypes
    ("c3", 24, {}),                                # two tips + secondary goal: pre-selection by secondary fitness
    ("c2", 130, {"mode": "bio2"}),                 # no memetic phase, 16 generations, several children per lane
    ("c4", 16, {"mode": "bio2_memetic_l", "fk_mode": abi.FK_LINEAR}),
])
def test_trajectory_bit_exact(sims, oracles, templates, cfg, pop, kw):
    pc.trajectory(sims[cfg], oracles[cfg], templates[cfg], n=2, pop=pop, steps_list=(1, 3), **kw)


def test_trajectory_two_wavefronts_and_islands(sims, oracles, templates, monkeypatch):
    """128 lanes: the two species run concurrently on one wavefront each, two children per lane kept in LDS"""
    monkeypatch.setenv("BIOIK_SOLVE_THREADS", "128")
    pc.trajectory(sims["c2"], oracles["c2"], templates["c2"], n=2, pop=128, steps_list=(2,), islands=2)


@pytest.mark.parametrize("env", [
    {"BIOIK_SOLVE_THREADS": "128"},                                      # species-parallel + secondary pre-selection + memetic
    {"BIOIK_SOLVE_THREADS": "128", "BIOIK_SOLVE_STORE_CHILDREN": "0"},   # winners re-derived from the RNG instead of read back
    {"BIOIK_SOLVE_THREADS": "128", "BIOIK_SOLVE_SPECIES_PARALLEL": "0"}, # two wavefronts, species one after the other
    {"BIOIK_SOLVE_THREADS": "64", "BIOIK_SOLVE_STORE_CHILDREN": "0"},
    {"BIOIK_SOLVE_THREADS": "128", "BIOIK_SOLVE_CHILD_PAIRS": "0"},      # one child per trip instead of two
    {"BIOIK_SOLVE_THREADS": "128", "BIOIK_SOLVE_GENERAL": "1"},           # the kernel flavour that also carries floating / planar joints
    {"BIOIK_SOLVE_THREADS": "64", "BIOIK_SOLVE_SPECIES_PARALLEL": "1"},   # one wavefront, the species on its two halves
    {"BIOIK_SOLVE_THREADS": "64", "BIOIK_SOLVE_SPECIES_PARALLEL": "1", "BIOIK_SOLVE_STORE_CHILDREN": "0"},
    {"BIOIK_SOLVE_THREADS": "128", "BIOIK_SOLVE_COLUMNLESS": "1"},       # children computed where they are read: no genotype columns in LDS
    {"BIOIK_SOLVE_THREADS": "64", "BIOIK_SOLVE_COLUMNLESS": "1"},
    {"BIOIK_SOLVE_THREADS": "128", "BIOIK_SOLVE_COLUMNLESS": "2"},       # ... and scored two at a time
    {"BIOIK_SOLVE_THREADS": "128", "BIOIK_SOLVE_COLUMNLESS": "0", "BIOIK_SOLVE_STORE_CHILDREN": "0"},
])
def test_trajectory_workgroup_mappings(sims, oracles, templates, monkeypatch, env):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    pc.trajectory(sims["c3"], oracles["c3"], templates["c3"], n=1, pop=70, steps_list=(2,))
    pc.trajectory(sims["c2"], oracles["c2"], templates["c2"], n=1, pop=70, steps_list=(2,), fk_mode=abi.FK_LINEAR)


@pytest.mark.parametrize("pop", [70, 130])
def test_preselection_with_computed_children_several_per_lane(sims, oracles, templates, monkeypatch, pop):
    """pre-selection on the secondary goals with the children computed where they are read, two (pop=70 on 32-lane groups) and four (pop=130)
    children per lane and trip, each with a ragged last trip; the bitonic order of a non-power-of-two population"""
    for k, v in {"BIOIK_SOLVE_THREADS": "64", "BIOIK_SOLVE_SPECIES_PARALLEL": "1", "BIOIK_SOLVE_COLUMNLESS": "1"}.items():
        monkeypatch.setenv(k, v)
    pc.trajectory(sims["c3"], oracles["c3"], templates["c3"], n=1, pop=pop, steps_list=(2,))
    pc.trajectory(sims["c4"], oracles["c4"], templates["c4"], n=1, pop=pop, steps_list=(1,))


@pytest.mark.parametrize("pop", [9, 70, 333])
def test_joint_walk_of_both_species_children(sims, oracles, templates, monkeypatch, pop):
    """both species on the halves of one wavefront, secondary goals, children in pairs: the 64 lanes walk the random prefixes of BOTH species'
    pre-selected children as one list (solve_body<.., JOINT>, k_solve_lean_clj) and find each species' two best over the whole wavefront --
    the same winners as with one half per species (BIOIK_SOLVE_NO_JOINT), which is the oracle's result"""
    for k, v in {"BIOIK_SOLVE_THREADS": "64", "BIOIK_SOLVE_SPECIES_PARALLEL": "1", "BIOIK_SOLVE_COLUMNLESS": "2"}.items():
        monkeypatch.setenv(k, v)
    pc.trajectory(sims["c3"], oracles["c3"], templates["c3"], n=2, pop=pop, steps_list=(1, 3))
    if pop <= 128:
        pc.trajectory(sims["c4"], oracles["c4"], templates["c4"], n=1, pop=pop, steps_list=(1,))
    monkeypatch.setenv("BIOIK_SOLVE_NO_JOINT", "1")
    pc.trajectory(sims["c3"], oracles["c3"], templates["c3"], n=1, pop=pop, steps_list=(2,))


@pytest.mark.parametrize("env", [
    {"BIOIK_SOLVE_TWO_PHASE": "1"},
    {"BIOIK_SOLVE_TWO_PHASE": "1", "BIOIK_SOLVE_GENERAL": "1"},
    {"BIOIK_SOLVE_TWO_PHASE": "1,2"},  # a chain of hand-overs (three launches)
])
def test_two_launch_solve(sims, oracles, templates, monkeypatch, env):
    """a solve split over two launches (SolveArgs::step_begin ...): the state handed over after K steps — elites, solution, bookkeeping —
    continues to the bit-identical result; with pop=128 and exact FK the first launch runs under the half-wavefront mapping with computed
    children, otherwise under the usual one; also with islands and with a secondary goal"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    pc.trajectory(sims["c2"], oracles["c2"], templates["c2"], n=1, pop=128, steps_list=(3,))
    pc.trajectory(sims["c3"], oracles["c3"], templates["c3"], n=1, pop=40, steps_list=(3,), islands=2)
    pc.trajectory(sims["c2"], oracles["c2"], templates["c2"], n=1, pop=24, steps_list=(1, 4), fk_mode=abi.FK_LINEAR, mode="bio2_memetic_l")


def test_edge_cases(sims, oracles, templates):
    h, o, t = sims["c2"], oracles["c2"], templates["c2"]
    p = abi.default_solve_params(population=16, max_steps=0)
    seeds = np.tile(t.model.default_positions(), (3, 1))
    params = np.tile(t.pack_params(), (3, 1))
    sol, fit, suc, steps = h.solve_batch(p, seeds, params)  # no budget: the seed comes back, fitness DBL_MAX (ik_parallel.h:208-209)
    assert np.array_equal(sol, seeds) and np.all(fit == np.finfo(float).max) and not suc.any() and not steps.any()
    sol, fit, suc, steps = h.solve_batch(p, seeds[:0], params[:0])  # empty batch
    assert sol.shape == (0, h.V)
    # a query that starts at its goal succeeds after the first step
    from bio_ik_amd.workload import make_queries
    s2, p2, targets = make_queries(t, o.active_variables, o.fk_genes, 2, seed=3)
    s2[:, o.active_variables] = targets
    p = abi.default_solve_params(population=16, max_steps=5)
    sol, fit, suc, steps = h.solve_batch(p, s2, p2)
    assert suc.all() and np.all(steps == 1)



def test_secondary_goals_of_every_kind_in_whole_solves(hostsim_lib, templates):
    """Whole solves with five secondary goals at once -- MinimalDisplacementGoal, AvoidJointLimitsGoal, CenterJointsGoal and a RegularizationGoal made secondary
    (sums over the joint values, whose terms the lanes of the line search share: solve_body's secondary_shared) and, between them in the goals' order, a
    JointVariableGoal (evaluated lane by lane) -- on the seven-joint arm and on both arms with the torso: the oracle's trajectories bit for bit, pre-selection
    and memetic phase included, exact and linearised phenotypes, the quadratic and the linear line search"""
    from bio_ik_amd import AvoidJointLimitsGoal, CenterJointsGoal, JointVariableGoal, MinimalDisplacementGoal, RegularizationGoal
    model = templates["c2"].model
    reg = RegularizationGoal(weight=0.6)
    reg.secondary_ = True
    sec = [MinimalDisplacementGoal(weight=0.7), AvoidJointLimitsGoal(weight=0.3), JointVariableGoal("r_elbow_flex_joint", -1.0, weight=0.5, secondary=True),
           CenterJointsGoal(weight=0.2), reg]
    t = ProblemTemplate(model, "right_arm", [PoseGoal("r_wrist_roll_link")] + sec)
    h, o = HipSolver(t, lib=hostsim_lib), orc.Oracle(t)
    pc.trajectory(h, o, t, n=2, pop=16, steps_list=(3,))
    pc.trajectory(h, o, t, n=1, pop=128, steps_list=(2,))
    pc.trajectory(h, o, t, n=1, pop=40, steps_list=(2,), fk_mode=abi.FK_LINEAR)
    pc.trajectory(h, o, t, n=1, pop=16, steps_list=(3,), mode="bio2_memetic_l")
    t2 = ProblemTemplate(model, "all", [PoseGoal("r_wrist_roll_link"), PoseGoal("l_wrist_roll_link")] + sec)
    h2, o2 = HipSolver(t2, lib=hostsim_lib), orc.Oracle(t2)
    pc.trajectory(h2, o2, t2, n=2, pop=128, steps_list=(2,))


def test_goal_sets_beyond_one_goal_per_tip(hostsim_lib, templates):
    """parity_cases.goal_sets_beyond_one_goal_per_tip on the host simulator"""
    pc.goal_sets_beyond_one_goal_per_tip(templates["c2"].model, lambda t: HipSolver(t, lib=hostsim_lib))


def test_branching_hand(hostsim_lib):
    """parity_cases.branching_hand on the host simulator"""
    pc.branching_hand(lambda t: HipSolver(t, lib=hostsim_lib))


def test_exact_joint_program(hostsim_lib, templates, monkeypatch):
    """parity_cases.exact_joint_program on the host simulator"""
    monkeypatch.setenv("BIOIK_COMPILE_EXACT", "1")
    pc.exact_joint_program(lambda t: HipSolver(t, lib=hostsim_lib), templates)


def test_line_search_step_without_bound(hostsim_lib, monkeypatch):
    """parity_cases.line_search_step_without_bound (quirk Q7: the reference's candidate at +-DBL_MAX; no joint value of magnitude 1e300 leaves the product)"""
    monkeypatch.setenv("BIOIK_COMPILE_EXACT", "1")
    pc.line_search_step_without_bound(lambda t: HipSolver(t, lib=hostsim_lib))


def test_line_search_on_a_flat_model(hostsim_lib, monkeypatch):
    """parity_cases.line_search_on_a_flat_model (quirk Q5: the reference's NaN candidate)"""
    monkeypatch.setenv("BIOIK_COMPILE_EXACT", "1")
    pc.line_search_on_a_flat_model(lambda t: HipSolver(t, lib=hostsim_lib))


def test_random_trees_and_goal_lists(hostsim_lib):
    """A dozen cases each of the two CPU soaks (tools/robot_fuzz_hostsim.py: random kinematic trees, the unfolded joint program bit for bit and the folded one to
    rounding; tools/goal_fuzz_hostsim.py: random goal lists on the PR2-like robot, bit for bit) -- the soaks of record are under profiles/ (600 + 600 + 400 cases)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for tool, args in (("robot_fuzz_hostsim.py", ["12", "5"]), ("goal_fuzz_hostsim.py", ["8", "5"])):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", tool)] + args, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        assert "0 mismatches" in r.stdout


def test_mimic_joints(hostsim_lib):
    """a joint that follows a gene and a joint that follows a joint outside every goal chain: function level and whole solves"""
    from bio_ik_amd import MinimalDisplacementGoal, PoseGoal, PositionGoal
    from conftest import mimic_robot
    m = mimic_robot()
    sec = MinimalDisplacementGoal(weight=0.5)
    sec.secondary_ = True
    t = ProblemTemplate(m, "arm", [PoseGoal("tool"), PositionGoal("finger_r_tip", weight=0.3), sec])
    h, o = HipSolver(t, lib=hostsim_lib), orc.Oracle(t)
    assert h.D == o.D == 5  # s1 s2 e1 w1 w2: the mimic joints and the off-chain finger are not genes
    pc.function_level(h, o, m, np.random.default_rng(5), n=60)  # 1e-12: the folded prismatic finger rounds differently
    # whole solves, bit for bit, on the revolute part (the elbow that follows the shoulder)
    t2 = ProblemTemplate(m, "arm", [PoseGoal("tool"), sec])
    h2, o2 = HipSolver(t2, lib=hostsim_lib), orc.Oracle(t2)
    pc.function_level(h2, o2, m, np.random.default_rng(6), n=60, exact_bits=True)
    pc.trajectory(h2, o2, t2, n=2, pop=16, steps_list=(1, 3))
    pc.trajectory(h2, o2, t2, n=1, pop=70, steps_list=(2,), fk_mode=abi.FK_LINEAR)


def test_mimic_of_a_mimic(hostsim_lib):
    """a joint that follows a joint that itself follows a gene (the reference never meets one: MoveIt's RobotModel::buildMimic resolves such chains before
    bio_ik sees the model; this library does the same wherever a model is built or handed in): the oracle's trajectories, and the same bits as the robot
    with the resolution written out by hand"""
    from bio_ik_amd import PoseGoal
    from conftest import mimic_robot
    sols = []
    for chain in ("chain", "resolved"):
        m = mimic_robot(chain)
        t = ProblemTemplate(m, "arm", [PoseGoal("tool")])
        h, o = HipSolver(t, lib=hostsim_lib), orc.Oracle(t)
        assert h.D == o.D == 4  # s1 s2 e1 w2
        pc.function_level(h, o, m, np.random.default_rng(6), n=40, exact_bits=True)
        pc.trajectory(h, o, t, n=2, pop=16, steps_list=(3,))
        seeds, params, _ = make_queries(t, h.active_variables, h.fk_genes, 3, seed=8)
        sols.append(h.solve_batch(abi.default_solve_params(population=32, max_steps=4, random_seed=2), seeds, params))
    assert all(np.array_equal(x, y) for x, y in zip(*sols))


def test_no_active_variable(hostsim_lib, pr2):
    """every joint of the group fixed (BioIKKinematicsQueryOptions::fixed_joints, problem.cpp:104-114): D = 0, the solve runs its
    budget and returns the seed, as the oracle does"""
    from bio_ik_amd import PoseGoal
    t0 = ProblemTemplate(pr2, "right_arm", [PoseGoal("r_wrist_roll_link")])
    names = [pr2.variable_names[v] for v in HipSolver(t0, lib=hostsim_lib).active_variables]
    t = ProblemTemplate(pr2, "right_arm", [PoseGoal("r_wrist_roll_link")], fixed_joints=names)
    h, o = HipSolver(t, lib=hostsim_lib), orc.Oracle(t)
    assert h.D == o.D == 0
    seeds, params = np.tile(pr2.default_positions(), (2, 1)), np.tile(t.pack_params(), (2, 1))
    for pop, fk in ((16, abi.FK_EXACT), (128, abi.FK_EXACT), (16, abi.FK_LINEAR)):
        p = abi.default_solve_params(population=pop, max_steps=2, random_seed=1, fk_mode=fk)
        got, want = h.solve_batch(p, seeds, params), o.solve_batch(p, orc.RNG_COUNTER, seeds, params)
        assert all(np.array_equal(a, b) for a, b in zip(got, want))
        assert np.array_equal(got[0], seeds) and not got[2].any()


def test_streamed_fitness(sims, oracles, templates):
    """bioik_stream_fitness_device (genes [unit][D][pop] -> fitness [unit][pop]): odd
    tails, several blocks per unit, a tree with parked branch frames; in the host simulator device pointers are host pointers"""
    from bio_ik_amd.workload import make_queries
    for cfg, pop, units in (("c2", 128, 2), ("c2", 77, 2), ("c2", 600, 1), ("c3", 130, 2)):
        h, o, t = sims[cfg], oracles[cfg], templates[cfg]
        seeds, params, _ = make_queries(t, h.active_variables, h.fk_genes, units, seed=6)
        genes = np.ascontiguousarray(np.random.default_rng(pop).uniform(-1, 1, size=(units, h.D, pop)))
        got = np.zeros((units, pop))
        h.stream_fitness_device(units, pop, seeds.ctypes.data, params.ctypes.data, genes.ctypes.data, got.ctypes.data, 0)
        for u in range(units):
            want, _ = o.fitness(abi.FK_EXACT, seeds[u], params[u], genes[u].T)
            assert np.array_equal(got[u], want), (cfg, pop, u)


def test_more_than_32_joints(hostsim_lib):
    """48 moving joints on one chain (op masks, winner copy and the memetic lanes beyond 32 ops): function level and whole
    solves bit for bit; 64 active variables are refused (the memetic phase needs lane D of a 64-lane wavefront)."""
    from bio_ik_amd import AvoidJointLimitsGoal, PoseGoal, snake
    from bio_ik_amd.solver import BioIKError
    m = snake(48)
    t = ProblemTemplate(m, "snake", [PoseGoal("tip"), AvoidJointLimitsGoal()])
    h, o = HipSolver(t, lib=hostsim_lib), orc.Oracle(t)
    assert h.D == o.D == 48
    pc.function_level(h, o, m, np.random.default_rng(11), n=24, exact_bits=True)
    pc.trajectory(h, o, t, n=2, pop=20, steps_list=(1, 2))
    pc.trajectory(h, o, t, n=1, pop=70, steps_list=(1,), fk_mode=abi.FK_LINEAR)
    with pytest.raises(BioIKError):
        HipSolver(ProblemTemplate(snake(64), "snake", [PoseGoal("tip")]), lib=hostsim_lib)


@pytest.mark.parametrize("base", ["floating", "planar"])
def test_floating_and_planar_joints(hostsim_lib, base):
    """a free base in front of the arm: 7 (translation + quaternion) or 3 (x, y, theta) genes for one joint, Jacobian columns by
    forward difference, quaternion genes renormalised after reproduction (forward_kinematics.h:120-135, 695-726,
    ik_evolution_2.cpp:203-215, 320-324)"""
    from bio_ik_amd import PoseGoal, PositionGoal
    from conftest import mobile_robot
    m = mobile_robot(base)
    t = ProblemTemplate(m, "whole", [PoseGoal("tool"), PositionGoal("base", weight=0.2)])
    h, o = HipSolver(t, lib=hostsim_lib), orc.Oracle(t)
    assert h.D == o.D == (10 if base == "floating" else 6)
    pc.function_level(h, o, m, np.random.default_rng(8), n=60, exact_bits=True)
    pc.trajectory(h, o, t, n=2, pop=16, steps_list=(1, 3))
    pc.trajectory(h, o, t, n=1, pop=70, steps_list=(2,), fk_mode=abi.FK_LINEAR)
    monkey = {"BIOIK_SOLVE_THREADS": "128"}
    os.environ.update(monkey)
    try:
        pc.trajectory(h, o, t, n=1, pop=130, steps_list=(2,))
    finally:
        for k in monkey:
            os.environ.pop(k, None)


@pytest.mark.parametrize("mid,with_base", [("planar", False), ("floating", False), ("planar", True)])
def test_floating_and_planar_joints_anywhere(hostsim_lib, mid, with_base):
    """a planar stage / a floating coupling in the MIDDLE of the chain, and two multi-variable joints on one chain (round 5: forward_kinematics.h:120-135,
    331-354 take them wherever they are): (F_src o C) o J with the joint frame parked per individual, forward-difference Jacobian columns against the
    frame of the op in front; function level against the oracle, whole solves, the gradient family"""
    from bio_ik_amd import PoseGoal, PositionGoal
    from conftest import stage_robot
    m = stage_robot(mid, with_base)
    t = ProblemTemplate(m, "whole", [PoseGoal("tool"), PositionGoal("stage", weight=0.2)])
    h, o = HipSolver(t, lib=hostsim_lib), orc.Oracle(t)
    assert h.D == o.D == 4 + (7 if mid == "floating" else 3) + (3 if with_base else 0)
    pc.function_level(h, o, m, np.random.default_rng(9), n=60, exact_bits=True)
    pc.trajectory(h, o, t, n=2, pop=16, steps_list=(1, 3))
    pc.trajectory(h, o, t, n=1, pop=70, steps_list=(2,), fk_mode=abi.FK_LINEAR)
    pc.trajectory(h, o, t, n=2, pop=8, steps_list=(6,), mode="gd_c")
    pc.trajectory(h, o, t, n=2, pop=8, steps_list=(4,), mode="jac")


def test_wall_clock_timeout(sims, oracles, templates):
    """the caller's timeout (ik_parallel.h:160): every query runs at least one step, then stops when the launch's clock passes the
    budget; a generous timeout changes nothing"""
    h, o, t = sims["c2"], oracles["c2"], templates["c2"]
    from bio_ik_amd.workload import make_queries
    seeds, params, _ = make_queries(t, o.active_variables, o.fk_genes, 3, seed=9)
    free = h.solve_batch(abi.default_solve_params(population=16, max_steps=6, random_seed=2), seeds, params)
    slack = h.solve_batch(abi.default_solve_params(population=16, max_steps=6, random_seed=2, timeout=3600.0), seeds, params)
    assert all(np.array_equal(a, b) for a, b in zip(free, slack))
    tight = h.solve_batch(abi.default_solve_params(population=16, max_steps=1000, random_seed=2, timeout=1e-9), seeds, params)
    assert np.all(tight[3] == 1)  # one step each (`iteration != 0`), the state after that step is what comes back
    one = h.solve_batch(abi.default_solve_params(population=16, max_steps=1, random_seed=2), seeds, params)
    assert all(np.array_equal(a, b) for a, b in zip(tight, one))


def test_solve_batch_multi_equals_single_handle(hostsim_lib, templates, sims):
    """bioik_solve_batch_multi: three handles of one template (on a node: one per GPU), contiguous shards, one host thread each;
    identical to the single-handle solve, also with a query offset and with fewer queries than handles"""
    t = templates["c2"]
    from bio_ik_amd.workload import make_queries
    h0 = sims["c2"]
    others = [HipSolver(t, lib=hostsim_lib) for _ in range(2)]
    seeds, params, _ = make_queries(t, h0.active_variables, h0.fk_genes, 7, seed=31)
    p = abi.default_solve_params(population=16, max_steps=3, random_seed=8)
    for first in (0, 1000):
        h0.set_first_query(first)
        want = h0.solve_batch(p, seeds, params)
        got = h0.solve_batch_multi(others, p, seeds, params)
        assert all(np.array_equal(a, b) for a, b in zip(want, got))
        got2 = h0.solve_batch_multi(others, p, seeds[:2], params[:2])
        assert all(np.array_equal(a[:2], b) for a, b in zip(want, got2))
    h0.set_first_query(0)
    from bio_ik_amd.solver import BioIKError
    with pytest.raises(BioIKError):
        h0.solve_batch_multi([sims["c3"]], p, seeds, params)  # another template


def test_balance_goal(hostsim_lib):
    """BalanceGoal (goal_types.cpp:231-272): every link with a URDF mass is a tip; the device accumulates the centre of mass as the
    walk reaches the tips.  Function level against both oracle arithmetics (the device visits the tips in walk order, the reference in
    link order: sums agree to rounding), and a small FK -> IK -> FK round trip on pose + balance."""
    from bio_ik_amd import AvoidJointLimitsGoal, BalanceGoal, PoseGoal
    from conftest import balance_robot
    m = balance_robot()
    for goals in ([PoseGoal("a_tool"), BalanceGoal((0.02, -0.01, 0.0), weight=0.8)], [BalanceGoal((0.0, 0.0, 0.0))],
                  [BalanceGoal((0.01, 0.0, 0.0)), PoseGoal("b_tool"), AvoidJointLimitsGoal(weight=0.2)]):
        t = ProblemTemplate(m, "body", goals)
        h, o = HipSolver(t, lib=hostsim_lib), orc.Oracle(t)
        assert h.T == o.T >= 10
        for mode in (0, 1):
            with pc.oracle_arithmetic(mode):
                pc.function_level(h, o, m, np.random.default_rng(15), n=40, frame_tol=1e-12, fit_rtol=1e-10)
    t = ProblemTemplate(m, "body", [PoseGoal("a_tool"), BalanceGoal(weight=1.0)])
    h, o = HipSolver(t, lib=hostsim_lib), orc.Oracle(t)
    with pc.oracle_arithmetic(0):
        seeds, params, off = pc.balance_queries(t, o, 3, seed=8)
    sol, fit, suc, steps = h.solve_batch(abi.default_solve_params(population=24, max_steps=80, random_seed=4), seeds, params)
    assert suc.sum() >= 2
    with pc.oracle_arithmetic(0):
        perr, rerr = pc.pose_errors(o, sol, params)
        berr = pc.balance_errors(t, o, sol, params, off)
    assert perr[suc == 1].max() < 1e-4 and rerr[suc == 1].max() < 1e-3 and berr[suc == 1].max() < 1e-4


@pytest.mark.parametrize("cfg", ["c2", "c3", "c4"])
def test_gradient_descent_and_jacobian_solvers(sims, oracles, templates, cfg):
    """modes gd_c and jac (reference src/ik_gradient.cpp:136-251, 42-133): the kernel bodies against the oracle, bit for bit"""
    out = pc.point_solvers(sims[cfg], oracles[cfg], templates[cfg], n=3)
    if cfg == "c2":
        assert out[2].all()  # jac reaches a pose goal near its seed within ten steps


def test_submit_wait_pipelining(sims, templates):
    """bioik_solve_batch_submit / _wait (host-simulated kernels): batches kept in flight on the handle's six slots return what the
    synchronous call returns, in any order of waiting, and a seventh submit completes the oldest ticket by itself"""
    from bio_ik_amd.workload import make_queries
    h, t = sims["c2"], templates["c2"]
    p = abi.default_solve_params(population=16, max_steps=2, random_seed=5)
    batches = []
    for k in range(8):
        seeds, params, _ = make_queries(t, h.active_variables, h.fk_genes, 1 + k % 2, seed=100 + k)
        batches.append((seeds, params, h.solve_batch(p, seeds, params)))
    tickets = [h.submit_batch(p, b[0], b[1]) for b in batches]   # eight submits on six slots: the first two complete on the way
    for k in (4, 0, 7, 2, 1, 6, 3, 5):
        sol, fit, suc, steps = h.wait_batch(tickets[k])
        ref = batches[k][2]
        assert np.array_equal(sol, ref[0]) and np.array_equal(fit, ref[1]) and np.array_equal(suc, ref[2]) and np.array_equal(steps, ref[3])
    h.wait_batch(tickets[0])  # waiting twice is harmless
    with pytest.raises(Exception):
        h.wait_batch((10 ** 6, None, None))


def test_throughput_schedule_changes_no_result(sims, oracles, templates):
    """bioik_solve_params::schedule = BIOIK_SCHEDULE_THROUGHPUT (the whole solve with both species of a query on one wavefront and computed
    children): the oracle's trajectories, like every other lane mapping"""
    pc.trajectory(sims["c2"], oracles["c2"], templates["c2"], n=2, pop=128, steps_list=(1, 4), schedule=abi.SCHEDULE_THROUGHPUT)
    pc.trajectory(sims["c2"], oracles["c2"], templates["c2"], n=1, pop=200, steps_list=(2,), schedule="throughput", islands=2)
    pc.trajectory(sims["c3"], oracles["c3"], templates["c3"], n=1, pop=128, steps_list=(2,), schedule=abi.SCHEDULE_THROUGHPUT)  # (no such mapping: as under LATENCY)
    with pytest.raises(Exception):
        sims["c2"].solve_batch(abi.default_solve_params(schedule=7), np.zeros((1, sims["c2"].V)), np.zeros((1, sims["c2"].P)))
    # BIOIK_SCHEDULE_AUTO: the asynchronous entry turns to the dense mapping once two solves of the handle are in flight
    from bio_ik_amd.workload import make_queries
    h, t = sims["c2"], templates["c2"]
    p = abi.default_solve_params(population=128, max_steps=2, random_seed=5, schedule="auto")
    seeds, params, _ = make_queries(t, h.active_variables, h.fk_genes, 1, seed=300)
    want = h.solve_batch(p, seeds, params)
    tickets = [h.submit_batch(p, seeds, params) for _ in range(5)]
    for tk in tickets:
        got = h.wait_batch(tk)
        assert all(np.array_equal(a, b) for a, b in zip(want, got))


def test_a_helper_that_never_answers_is_an_error_not_a_result(sims, templates, monkeypatch):
    """The helped kernel's wavefronts meet at words in LDS, and a wait gives up after a bounded number of polls (bioik_platform.h: p_flag_wait_ge).  A main wavefront
    whose helper never answers (BIOIK_SOLVE_DEBUG_FLAGS=1: the helper of species 0 leaves at once) must not go on as if nothing had happened: it sets the call's
    error word (SolveArgs::error), and the host-pointer entries report BIOIK_ERR_HIP instead of handing out the arrays -- the synchronous call, the ticket's wait,
    and the device-pointer entry at the handle's next call.  The reference's boost::barrier cannot time out (ik_parallel.h:64-67)."""
    from bio_ik_amd.solver import BioIKError
    from bio_ik_amd.workload import make_queries
    h, t = sims["c2"], templates["c2"]
    seeds, params, _ = make_queries(t, h.active_variables, h.fk_genes, 1, seed=21)
    p = abi.default_solve_params(population=128, max_steps=2, random_seed=3)
    good = h.solve_batch(p, seeds, params)
    monkeypatch.setenv("BIOIK_SOLVE_DEBUG_FLAGS", "1")
    with pytest.raises(BioIKError) as e:
        h.solve_batch(p, seeds, params)
    assert e.value.code == abi.ERR_HIP and "rendezvous" in str(e.value)
    tk = h.submit_batch(p, seeds, params)
    with pytest.raises(BioIKError) as e:
        h.wait_batch(tk)
    assert e.value.code == abi.ERR_HIP
    monkeypatch.delenv("BIOIK_SOLVE_DEBUG_FLAGS")
    again = h.solve_batch(p, seeds, params)  # the handle is as good as before
    assert all(np.array_equal(a, b) for a, b in zip(good, again))


def test_kernels_compiled_for_one_mapping(sims, oracles, templates, monkeypatch):
    """The two builds of the computed-children kernel for the 128-register budget know their lane mapping at compile time (solve_body<.., FIXED>):
    k_solve_lean_cl4 = 128 lanes, a wavefront per species, children in pairs -- what the launcher picks for the 31-joint chain at 512 children, and,
    forced (BIOIK_SOLVE_FOUR_WAVES), for any problem under that mapping, with and without a secondary goal, odd populations included;
    k_solve_lean_cl64w4 = the throughput schedule's kernel (its own test above).  The oracle's trajectories bit for bit."""
    pc.trajectory(sims["c4"], oracles["c4"], templates["c4"], n=1, pop=512, steps_list=(1, 2))  # (the launcher's own choice)
    for k, v in {"BIOIK_SOLVE_THREADS": "128", "BIOIK_SOLVE_COLUMNLESS": "2", "BIOIK_SOLVE_FOUR_WAVES": "1"}.items():
        monkeypatch.setenv(k, v)
    pc.trajectory(sims["c2"], oracles["c2"], templates["c2"], n=2, pop=128, steps_list=(1, 3))
    pc.trajectory(sims["c2"], oracles["c2"], templates["c2"], n=1, pop=70, steps_list=(2,))
    pc.trajectory(sims["c3"], oracles["c3"], templates["c3"], n=1, pop=200, steps_list=(2,))
    pc.trajectory(sims["c4"], oracles["c4"], templates["c4"], n=1, pop=33, steps_list=(2,), islands=2)
    monkeypatch.setenv("BIOIK_SOLVE_TWO_PHASE", "1")  # ... and resumed from a hand-over
    pc.trajectory(sims["c3"], oracles["c3"], templates["c3"], n=2, pop=128, steps_list=(3,))


def test_helper_wavefronts(sims, oracles, templates, monkeypatch):
    """k_solve_lean_cl4h (solve_body<.., FIXED = 5>): k_solve_lean_cl4's two wavefronts plus two HELPERS that walk half of every generation's children; the
    wavefronts hand over through words in LDS and the helpers reach no barrier.  What the launcher picks for launches of up to 1024 units of a PoseGoal-class
    problem: populations of one and two trips per wavefront, an odd one, islands that stop each other, a solve resumed from a hand-over -- the oracle's
    trajectories bit for bit; with BIOIK_SOLVE_HELPED=0 the same launches run k_solve_lean_cl4 itself."""
    for helped in ("1024", "0"):
        monkeypatch.setenv("BIOIK_SOLVE_HELPED", helped)
        pc.trajectory(sims["c2"], oracles["c2"], templates["c2"], n=2, pop=128, steps_list=(1, 3))
        pc.trajectory(sims["c2"], oracles["c2"], templates["c2"], n=1, pop=200, steps_list=(2,), islands=2, island_sync=1)
        pc.trajectory(sims["c2"], oracles["c2"], templates["c2"], n=1, pop=131, steps_list=(2,), no_wipeout=1)
    monkeypatch.setenv("BIOIK_SOLVE_HELPED", "1024")
    # ... with secondary goals (the main wavefront pre-selects, the helper takes the upper half of the survivors' walks): the 31-joint chain at 512 and at an odd
    # number of children, a 7-joint arm with a MinimalDisplacementGoal at one and two trips per wavefront
    from bio_ik_amd import MinimalDisplacementGoal, PoseGoal
    pc.trajectory(sims["c4"], oracles["c4"], templates["c4"], n=1, pop=512, steps_list=(2,))
    pc.trajectory(sims["c4"], oracles["c4"], templates["c4"], n=1, pop=150, steps_list=(2,), islands=2)
    ts = ProblemTemplate(templates["c2"].model, "right_arm", [PoseGoal("r_wrist_roll_link"), MinimalDisplacementGoal()])
    hs, os_ = HipSolver(ts, lib=sims["c2"].L), orc.Oracle(ts)
    pc.trajectory(hs, os_, ts, n=2, pop=128, steps_list=(3,))
    pc.trajectory(hs, os_, ts, n=1, pop=200, steps_list=(2,))
    monkeypatch.setenv("BIOIK_SOLVE_TWO_PHASE", "1,3")  # (the helped kernel as the second and third launch of a solve)
    pc.trajectory(sims["c2"], oracles["c2"], templates["c2"], n=2, pop=128, steps_list=(5,))
    monkeypatch.delenv("BIOIK_SOLVE_TWO_PHASE")
    monkeypatch.setenv("BIOIK_SOLVE_DRAIN_TEST", "4")   # (... and as the consumer of units that leave their first launch at their own step)
    pc.trajectory(sims["c2"], oracles["c2"], templates["c2"], n=3, pop=128, steps_list=(6,))


def test_joint_walk_under_the_small_register_budget(sims, oracles, templates, monkeypatch):
    """k_solve_lean_clj4 (solve_body<.., JOINT, SLIM, FIXED = 4>): both species of a query on the halves of one wavefront, secondary goals, the
    pre-selected children of both species walked as ONE list whose fitness values are parked per species in LDS -- what the launcher picks for
    the two-armed problem wherever a CU's LDS holds more than twelve queries.  Populations around the lane counts (a list shorter than a half, a
    list over several trips, odd tails), islands, a resumed launch; and the three-wavefront kernel of the same mapping gives the same trajectories."""
    for k, v in {"BIOIK_SOLVE_THREADS": "64", "BIOIK_SOLVE_SPECIES_PARALLEL": "1", "BIOIK_SOLVE_COLUMNLESS": "2"}.items():
        monkeypatch.setenv(k, v)
    for pop in (9, 33, 70):
        pc.trajectory(sims["c3"], oracles["c3"], templates["c3"], n=2, pop=pop, steps_list=(2,))
    pc.trajectory(sims["c3"], oracles["c3"], templates["c3"], n=1, pop=200, steps_list=(1, 2), islands=2)
    monkeypatch.setenv("BIOIK_SOLVE_THREE_WAVES", "1")
    pc.trajectory(sims["c3"], oracles["c3"], templates["c3"], n=2, pop=33, steps_list=(2,))
    monkeypatch.delenv("BIOIK_SOLVE_THREE_WAVES")
    monkeypatch.setenv("BIOIK_SOLVE_TWO_PHASE", "1")
    pc.trajectory(sims["c3"], oracles["c3"], templates["c3"], n=2, pop=64, steps_list=(3,))


def test_preselection_sort_keys_and_the_exact_path_behind_them(sims, oracles, templates, monkeypatch):
    """The pre-selection orders the children of a species by 64-bit keys -- the upper 54 bits of a secondary fitness and the child index -- and sorts
    exactly whenever sorted neighbours differ in the dropped bits only (about once in a billion generations).  With keys that give up 36 or 50 bits that
    happens in most generations: the oracle's trajectories bit for bit either way, for four and for eight children per lane (C3's and C4's kernels),
    for all-zero costs (C4: every child inside AvoidJointLimitsGoal's free zone ties with every other) and for the generic computed-children kernel.  The
    half-wavefront kernels pick the two best children of a generation from keys of the same kind (the primary fitness's upper bits and the position),
    with the exact reduction behind a count of the candidates that share the runner-up's upper bits."""
    for drop in ("36", "50", "10"):
        monkeypatch.setenv("BIOIK_SOLVE_SORT_KEY_DROP", drop)
        pc.trajectory(sims["c4"], oracles["c4"], templates["c4"], n=1, pop=512, steps_list=(2,))
        # (the same keys pick the two best children of a generation in the half-wavefront kernels: the dense kernel here, the joint walk below)
        pc.trajectory(sims["c2"], oracles["c2"], templates["c2"], n=4, pop=128, steps_list=(4,), schedule=abi.SCHEDULE_THROUGHPUT)
        for k, v in {"BIOIK_SOLVE_THREADS": "64", "BIOIK_SOLVE_SPECIES_PARALLEL": "1", "BIOIK_SOLVE_COLUMNLESS": "2"}.items():
            monkeypatch.setenv(k, v)
        pc.trajectory(sims["c3"], oracles["c3"], templates["c3"], n=2, pop=128, steps_list=(3,))
        monkeypatch.setenv("BIOIK_SOLVE_THREE_WAVES", "1")
        pc.trajectory(sims["c3"], oracles["c3"], templates["c3"], n=1, pop=100, steps_list=(2,))
        for k in ("BIOIK_SOLVE_THREADS", "BIOIK_SOLVE_SPECIES_PARALLEL", "BIOIK_SOLVE_COLUMNLESS", "BIOIK_SOLVE_THREE_WAVES"):
            monkeypatch.delenv(k)


def test_preselection_by_selection_and_ties_of_the_whole_fitness(sims, oracles, templates, monkeypatch):
    """The kernels of the 128-register budget find the pre-selection's survivors by SELECTION (select_threshold: a bisection for the k-th least key, no sort)
    and write them in lane order; the reference's stable order among them matters only where two children tie on their whole fitness, and is restored there
    from the candidates' secondary fitness.  (i) Both ways give the oracle's trajectories bit for bit -- 512 and 300 children on a wavefront per species
    (eight per lane, padding), the helped kernel, keys that give up 36 bits (classes of values on both sides of the threshold); the joint walk of the two-armed
    problem (four keys per lane on half a wavefront) keeps the sort, which is the cheaper of the two there.  (ii) With the children's fitness made coarse (BIOIK_SOLVE_TIE_TEST_BITS: a test switch; most generations then have several best
    children) the selection gives what the sort gives, bit for bit: the sort's positions ARE the stable order."""
    c3map = {"BIOIK_SOLVE_THREADS": "64", "BIOIK_SOLVE_SPECIES_PARALLEL": "1", "BIOIK_SOLVE_COLUMNLESS": "2"}

    def runs(check):
        check("c4", 1, 512, 2, {})
        check("c4", 1, 512, 2, {"BIOIK_SOLVE_HELPED": "0"})
        check("c4", 1, 300, 2, {"BIOIK_SOLVE_HELPED": "0", "BIOIK_SOLVE_FOUR_WAVES": "1"})  # (the 128-register build for fewer children than the launcher gives it)
        check("c3", 2, 128, 2, c3map)
        check("c3", 1, 100, 2, c3map)

    def against_oracle(name, n, pop, steps, env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        pc.trajectory(sims[name], oracles[name], templates[name], n=n, pop=pop, steps_list=(steps,))
        for k in env:
            monkeypatch.delenv(k)
    for mode, drop in (("0", "10"), ("1", "10"), ("1", "36")):
        monkeypatch.setenv("BIOIK_SOLVE_PRESELECT", mode)
        monkeypatch.setenv("BIOIK_SOLVE_SORT_KEY_DROP", drop)
        runs(against_oracle)
    monkeypatch.delenv("BIOIK_SOLVE_SORT_KEY_DROP")
    changed = []

    def sort_against_selection(name, n, pop, steps, env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        h, t = sims[name], templates[name]
        seeds, params, _ = make_queries(t, h.active_variables, h.fk_genes, n, seed=7)
        p = abi.default_solve_params(population=pop, max_steps=steps, random_seed=11, mode="bio2")  # (no memetic phase: it is the children that improve an elite)
        out = {}
        for bits in ("0", "46", "50"):
            monkeypatch.setenv("BIOIK_SOLVE_TIE_TEST_BITS", bits)
            for mode in ("0", "1"):
                monkeypatch.setenv("BIOIK_SOLVE_PRESELECT", mode)
                out[bits, mode] = h.solve_batch(p, seeds, params)
            for a, b in zip(out[bits, "0"], out[bits, "1"]):
                assert np.array_equal(a, b), (name, pop, bits)
        changed.append(not np.array_equal(out["0", "1"][0], out["50", "1"][0]))  # (the coarse values took the search elsewhere: ties were there to be settled)
        monkeypatch.delenv("BIOIK_SOLVE_TIE_TEST_BITS")
        for k in env:
            monkeypatch.delenv(k)
    runs(sort_against_selection)
    assert all(changed)
    # (a runner-up that ties with parent 0: the reference's swap has put that parent at the first winner's position, so the two winners' places in the list decide --
    # six queries over five steps meet the case; without the selection's word on it five of thirty such runs differ)
    monkeypatch.setenv("BIOIK_SOLVE_HELPED", "0")
    h, t = sims["c4"], templates["c4"]
    for bits, mode in (("48", "bio2"), ("50", "bio2_memetic")):
        seeds, params, _ = make_queries(t, h.active_variables, h.fk_genes, 6, seed=int(bits) + 512)
        p = abi.default_solve_params(population=512, max_steps=5, random_seed=int(bits), mode=mode)
        monkeypatch.setenv("BIOIK_SOLVE_TIE_TEST_BITS", bits)
        out = []
        for m in ("0", "1"):
            monkeypatch.setenv("BIOIK_SOLVE_PRESELECT", m)
            out.append(h.solve_batch(p, seeds, params))
        assert all(np.array_equal(a, b) for a, b in zip(*out)), (bits, mode)


def test_units_that_change_launch_at_their_own_step(sims, oracles, templates, monkeypatch):
    """SolveArgs::resident: the throughput schedule's stragglers leave for the launch with the faster lone step when the chip runs empty -- every
    unit from whatever step it is at, its step count travelling with its state.  Which unit leaves when is a matter of timing on the device; here
    the test pattern (BIOIK_SOLVE_DRAIN_TEST=n: unit u after 1 + hash(u) % n steps) stands in for it.  The oracle's trajectories bit for bit, for the
    dense kernel and its successor, for islands (whose step counts decide the selection) and for problems with secondary goals."""
    monkeypatch.setenv("BIOIK_SOLVE_DRAIN_TEST", "5")
    pc.trajectory(sims["c2"], oracles["c2"], templates["c2"], n=12, pop=128, steps_list=(9,), schedule=abi.SCHEDULE_THROUGHPUT)
    pc.trajectory(sims["c2"], oracles["c2"], templates["c2"], n=6, pop=128, steps_list=(7,), islands=2, island_sync=1, schedule=abi.SCHEDULE_THROUGHPUT)
    pc.trajectory(sims["c2"], oracles["c2"], templates["c2"], n=6, pop=32, steps_list=(8,))
    pc.trajectory(sims["c3"], oracles["c3"], templates["c3"], n=4, pop=128, steps_list=(7,))
    pc.trajectory(sims["c2"], oracles["c2"], templates["c2"], n=4, pop=16, steps_list=(9,), fk_mode=abi.FK_LINEAR)


def test_islands_that_stop_each_other(sims, oracles, templates, monkeypatch):
    """bioik_solve_params::island_sync = 1: "any island succeeds => all stop" (ik_parallel.h:102, 160-178) in lock step -- the answer is the best of the
    islands that passed after the LEAST number of steps, whatever the other islands went on to find; bit for bit the oracle's, under one launch and under
    hand-overs, for the evolution and for the gradient family (the reference's gd_4: four solver threads)"""
    pc.trajectory(sims["c2"], oracles["c2"], templates["c2"], n=6, pop=32, steps_list=(12,), islands=3, island_sync=1, seed=3)
    pc.trajectory(sims["c2"], oracles["c2"], templates["c2"], n=3, pop=16, steps_list=(30,), islands=4, island_sync=1, fk_mode=abi.FK_LINEAR, seed=4)
    pc.trajectory(sims["c3"], oracles["c3"], templates["c3"], n=2, pop=24, steps_list=(6,), islands=2, island_sync=1)
    pc.trajectory(sims["c2"], oracles["c2"], templates["c2"], n=3, pop=8, steps_list=(25,), islands=4, island_sync=1, mode="gd_c")
    monkeypatch.setenv("BIOIK_SOLVE_TWO_PHASE", "2,5")
    pc.trajectory(sims["c2"], oracles["c2"], templates["c2"], n=6, pop=32, steps_list=(12,), islands=3, island_sync=1, seed=3)
    # the two selection rules do differ: somewhere an island that passes later has the better fitness
    from bio_ik_amd.workload import make_queries
    h, t = sims["c2"], templates["c2"]
    seeds, params, _ = make_queries(t, h.active_variables, h.fk_genes, 24, seed=3)
    a = h.solve_batch(abi.default_solve_params(population=32, max_steps=16, random_seed=5, islands=3, island_sync=0), seeds, params)
    b = h.solve_batch(abi.default_solve_params(population=32, max_steps=16, random_seed=5, islands=3, island_sync=1), seeds, params)
    assert np.array_equal(a[2], b[2]) and np.all(b[3] <= a[3]) and np.any(b[3] < a[3])
    with pytest.raises(Exception):
        h.solve_batch(abi.default_solve_params(islands=2, island_sync=2), seeds[:1], params[:1])


def test_islands_sized_to_the_idle_chip(sims, templates):
    """bioik_solve_params::islands = BIOIK_ISLANDS_AUTO (0): min(16, 2048 / n) islands per query (at least four up to 1024 queries, one beyond; 64 up to eight queries and 32 up to sixteen since round 6) that stop each other -- the same solve as that count
    given explicitly with island_sync = 1; the gradient family keeps one island (an island count there names another solver)"""
    h, t = sims["c2"], templates["c2"]
    seeds, params, _ = make_queries(t, h.active_variables, h.fk_genes, 3, seed=21)
    a = h.solve_batch(abi.default_solve_params(population=16, max_steps=6, random_seed=3, islands=abi.ISLANDS_AUTO, fk_mode=abi.FK_LINEAR), seeds, params)
    b = h.solve_batch(abi.default_solve_params(population=16, max_steps=6, random_seed=3, islands=64, island_sync=1, fk_mode=abi.FK_LINEAR), seeds, params)
    c = h.solve_batch(abi.default_solve_params(population=16, max_steps=6, random_seed=3, islands=1, fk_mode=abi.FK_LINEAR), seeds, params)
    assert all(np.array_equal(x, y) for x, y in zip(a, b)) and not np.array_equal(a[0], c[0])
    # bioik_resolve_islands: what a caller that shards a request itself asks once per request (the plugin over several devices)
    import ctypes
    isl, sync = ctypes.c_int32(), ctypes.c_int32()
    for n, want in ((1, 64), (8, 64), (16, 32), (32, 16), (200, 10), (700, 4), (1024, 4), (1025, 1)):
        p = abi.default_solve_params(islands=abi.ISLANDS_AUTO)
        assert h.L.bioik_resolve_islands(h.problem, ctypes.byref(p), ctypes.c_size_t(n), ctypes.byref(isl), ctypes.byref(sync)) == 0
        assert (isl.value, sync.value) == (want, 1 if want > 1 else 0), (n, isl.value, sync.value)
    p = abi.default_solve_params(islands=3)
    assert h.L.bioik_resolve_islands(h.problem, ctypes.byref(p), ctypes.c_size_t(5), ctypes.byref(isl), ctypes.byref(sync)) == 0 and (isl.value, sync.value) == (3, 0)
    g0 = h.solve_batch(abi.default_solve_params(population=16, max_steps=6, random_seed=3, islands=abi.ISLANDS_AUTO, mode="gd_c"), seeds, params)
    g1 = h.solve_batch(abi.default_solve_params(population=16, max_steps=6, random_seed=3, islands=1, mode="gd_c"), seeds, params)
    assert all(np.array_equal(x, y) for x, y in zip(g0, g1))


def test_best_island_three_ways(sims, templates, monkeypatch):
    """round 6: the islands' reduction by a wavefront (inside the solve's launch, or a launch of its own) against the lane that walks the islands"""
    pc.island_selection_three_ways(sims["c2"], templates["c2"], monkeypatch, n=2, pop=8, steps=4, fk_mode=abi.FK_LINEAR, kind="tracking", noise=0.02, configs=((9, 1, False), (70, 1, True), (66, 0, False)))


def test_selection_ties_are_decided_by_position(hostsim_lib, monkeypatch):
    """joints without any range: every child of a generation is the same genotype, so every fitness of a generation is the same number and
    the elitist selection is decided by position alone (ik_evolution_2.cpp:410-431) -- the tie path of the wavefront-minimum top-2
    (whole wavefronts) and of the merging butterfly (half-wavefront groups) against the oracle"""
    from bio_ik_amd import snake
    m = snake(4, limit=0.0)
    t = ProblemTemplate(m, "snake", [PoseGoal("tip")])
    o = orc.Oracle(t)
    for env in ({"BIOIK_SOLVE_THREADS": "128"}, {"BIOIK_SOLVE_THREADS": "64", "BIOIK_SOLVE_SPECIES_PARALLEL": "1"}, {}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        h = HipSolver(t, lib=hostsim_lib)
        pc.trajectory(h, o, t, n=2, pop=128 if env else 16, steps_list=(2,))
        for k in env:
            monkeypatch.delenv(k)
