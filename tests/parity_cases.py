"""Parity cases shared by the GPU suite (tests/test_gpu_parity.py, through the C-ABI of libbioik_hip.so) and by the
host-simulator suite (tests/test_hostsim_parity.py, same kernel bodies built for the host).

`h` is a bio_ik_amd.solver.HipSolver, `o` an oracle.orc.Oracle of the same problem template.  The oracle must be in
trig mode 1 (bioik_sincos shared with the device) wherever bit-exactness is asserted."""
import contextlib

import numpy as np

from bio_ik_amd import abi
from bio_ik_amd.workload import make_queries
from conftest import random_configuration
from oracle import orc


@contextlib.contextmanager
def oracle_arithmetic(mode):
    """Run a block with the oracle in arithmetic mode 0 (the reference's own expressions + libm: the mode in which the oracle is
    pinned bit for bit against the reference's code, tests/test_oracle_vs_reference.py) or 1 (bioik_sincos + the fused forms the
    device evaluates); restores the previous mode."""
    prev = int(orc.lib().orc_get_trig_mode())
    orc.set_trig_mode(mode)
    try:
        yield
    finally:
        orc.set_trig_mode(prev)


def assert_same_structure(h, o):
    assert (h.D, h.T, h.P, h.V) == (o.D, o.T, o.P, o.V)
    assert np.array_equal(h.active_variables, o.active_variables)
    assert np.array_equal(h.tip_links, o.tip_links)


def function_level(h, o, model, rng, n=200, frame_tol=1e-12, fit_rtol=1e-10, exact_bits=False):
    """K7 exact FK, K4 goal fitness (exact and linear), K8 approximator tables, K1 reproduce, K9 success check."""
    assert_same_structure(h, o)
    seed = random_configuration(model, rng)
    genes = random_configuration(model, rng, n)[:, o.active_variables]
    par = rng.normal(size=o.P)
    # quaternion parameters must be unit quaternions for the twist-based success test to be meaningful
    a, b = o.fk_genes(seed, genes), h.fk_genes(seed, genes)
    assert np.abs(a - b).max() <= (0.0 if exact_bits else frame_tol)
    pa, sa = o.fitness(abi.FK_EXACT, seed, par, genes)
    pb, sb = h.fitness(abi.FK_EXACT, seed, par, genes)
    tol = 0.0 if exact_bits else fit_rtol
    assert np.all(np.abs(pa - pb) <= tol * np.abs(pa))
    assert np.all(np.abs(sa - sb) <= tol * np.abs(sa) + (0.0 if exact_bits else 1e-300))
    base = genes[0]
    near = base + 0.01 * rng.normal(size=(n, o.D))
    pa, sa = o.fitness(abi.FK_LINEAR, seed, par, near, base)
    pb, sb = h.fitness(abi.FK_LINEAR, seed, par, near, base)
    assert np.all(np.abs(pa - pb) <= tol * np.abs(pa))
    ta, da, _ = o.approximator(seed, base)
    tb, db = h.approximator(seed, base)
    assert np.abs(ta - tb).max() <= (0.0 if exact_bits else frame_tol)
    assert np.abs(da - db).max() <= (0.0 if exact_bits else frame_tol)
    # reproduce: bit-exact always (counter RNG + strict arithmetic on both sides)
    parents = rng.normal(size=(2, 2, o.D)) * 0.1
    parents[:, 0, :] = random_configuration(model, rng, 2)[:, o.active_variables]
    for lam, key, sp, gen in ((16, 12345, 1, 37), (130, 0xDEADBEEF, 0, 1601)):
        ga, gra = o.reproduce_counter(lam, key, sp, gen, parents)
        gb, grb = h.reproduce(lam, key, sp, gen, parents)
        assert np.array_equal(ga, gb) and np.array_equal(gra, grb)
    for kw in ({}, {"dpos": 0.05, "drot": 5.0, "dtwist": -1.0}, {"dpos": 0.3, "drot": -1.0, "dtwist": 0.2}):
        p = abi.default_solve_params(**kw)
        assert np.array_equal(o.check(p, seed, par, genes), h.check(p, seed, par, genes))


def success_check_near_goal(h, o, template, rng, n=64):
    """K9 around the decision threshold: goals taken from FK of the genes themselves, perturbed by ~dtwist."""
    seeds, params, targets = make_queries(template, o.active_variables, o.fk_genes, n, seed=int(rng.integers(1 << 30)))
    p = abi.default_solve_params()
    for k in range(0, n, 8):
        g = targets[k] + rng.normal(size=(16, o.D)) * rng.choice([0.0, 1e-7, 3e-6, 1e-5, 1e-3], size=(16, 1))
        a = o.check(p, seeds[k], params[k], g)
        b = h.check(p, seeds[k], params[k], g)
        assert np.array_equal(a, b)
        assert a[np.all(g == targets[k], axis=1)].all() if np.any(np.all(g == targets[k], axis=1)) else True


def trajectory(h, o, template, n, pop, steps_list, seed=7, exact_bits=True, **kw):
    """Whole solves with identical RNG streams: the device result must equal the oracle's (bit for bit when both sides
    use bioik_sincos and the explicitly fused forms of bioik_fused.h, compiler contraction off)."""
    seeds, params, _ = make_queries(template, o.active_variables, o.fk_genes, n, seed=seed)
    for steps in steps_list:
        p = abi.default_solve_params(population=pop, max_steps=steps, random_seed=11, **kw)
        sa = o.solve_batch(p, orc.RNG_COUNTER, seeds, params, n_threads=4)
        sb = h.solve_batch(p, seeds, params)
        if exact_bits:
            assert np.array_equal(sa[0], sb[0]), "solutions differ after %d steps: %g" % (steps, np.abs(sa[0] - sb[0]).max())
            assert np.array_equal(sa[1], sb[1])
        else:
            assert np.abs(sa[0] - sb[0]).max() < 1e-9
        assert np.array_equal(sa[2], sb[2]) and np.array_equal(sa[3], sb[3])


def pose_errors(o, sol, params, tip=0, off=0):
    """position [m] and rotation [rad] error of the returned solutions under the ORACLE's exact FK"""
    tips = o.fk(sol)
    perr = np.linalg.norm(tips[:, tip, :3] - params[:, off:off + 3], axis=1)
    dots = np.abs(np.einsum("ij,ij->i", tips[:, tip, 3:], params[:, off + 3:off + 7]))
    rerr = 2 * np.arccos(np.minimum(1.0, dots))
    return perr, rerr


def balance_queries(template, o, n, seed=5):
    """FK -> IK -> FK round trip with a BalanceGoal: goals [PoseGoal(tip), BalanceGoal]; the pose AND the centre-of-mass target are those of
    a random reachable configuration (oracle FK), the seed is another random configuration.  Returns seeds, params, (pose offset, balance offset)."""
    model = template.model
    seeds, params, targets = make_queries(template, o.active_variables, o.fk_genes, n, seed=seed)
    mass = np.asarray(model.link_mass)
    share = mass / mass.sum()
    centers = np.asarray(model.link_center)
    from bio_ik_amd.robot import quat_rotate
    off = [off for g, off in zip(template.goals, template.param_offsets) if g.opcode == abi.GOAL_BALANCE][0]
    full = np.tile(model.default_positions(), (n, 1))
    full[:, o.active_variables] = targets
    tips = o.fk(full)
    tip_of_link = {int(l): i for i, l in enumerate(o.tip_links)}
    for k in range(n):
        com = np.zeros(3)
        for l in np.nonzero(mass > 0)[0]:
            f = tips[k, tip_of_link[int(l)]]
            com += (f[:3] + quat_rotate(f[3:], centers[l])) * share[l]
        params[k, off:off + 3] = com
    return seeds, params, off


def balance_errors(template, o, sol, params, off):
    """horizontal distance [m] of the centre of mass of the returned configurations from the balance target, under the ORACLE's FK"""
    model = template.model
    mass = np.asarray(model.link_mass)
    share = mass / mass.sum()
    centers = np.asarray(model.link_center)
    from bio_ik_amd.robot import quat_rotate
    tips = o.fk(sol)
    tip_of_link = {int(l): i for i, l in enumerate(o.tip_links)}
    err = np.zeros(sol.shape[0])
    for k in range(sol.shape[0]):
        com = np.zeros(3)
        for l in np.nonzero(mass > 0)[0]:
            f = tips[k, tip_of_link[int(l)]]
            com += (f[:3] + quat_rotate(f[3:], centers[l])) * share[l]
        d = com - params[k, off:off + 3]
        ax = params[k, off + 3:off + 6]
        d = d - ax * (ax @ d)
        err[k] = np.linalg.norm(d)
    return err


def point_solvers(h, o, template, n=8, exact_jac=True):
    """gd_c and jac (src/ik_gradient.cpp) through the batched entry point against the oracle's restatement (itself pinned bit for bit
    against the reference's source, tests/test_oracle_vs_reference.py): tracking seeds, several step budgets.  gd_c touches only the
    shared exact-FK arithmetic: identical bits.  jac goes through acos (twist of the pose error): the shared implementation of bioik_acos.h since round 6 --
    identical bits on the host simulator and on the device (exact_jac=False: the tolerance of rounds 2 - 5, when the device used its own library's)."""
    seeds, params, _ = make_queries(template, o.active_variables, o.fk_genes, n, seed=41, kind="tracking")
    # islands = N: the reference's gd_N / gd_r_N / gd_c_N / jac_N (solver threads 1 ... N - 1 start at random configurations, the best
    # island wins); gd_r: random restarts after a step that did not improve (the draws of both are the counter generator's)
    for mode, islands, budgets in (("gd_c", 1, (1, 6, 20)), ("gd", 1, (1, 6, 20)), ("jac", 1, (1, 3, 10)), ("gd_r", 1, (1, 6, 40)), ("gd_r", 2, (8,)),
                                   ("gd", 4, (1, 6)), ("gd_c", 2, (6,)), ("jac", 4, (1, 5))):
        for st in budgets:
            p = abi.default_solve_params(mode=mode, max_steps=st, islands=islands, random_seed=7)
            a = o.solve_batch(p, orc.RNG_COUNTER, seeds, params, n_threads=4)
            b = h.solve_batch(p, seeds, params)
            if mode != "jac" or exact_jac:
                assert all(np.array_equal(x, y) for x, y in zip(a, b)), (mode, islands, st, np.abs(a[0] - b[0]).max())
            else:
                assert np.abs(a[0] - b[0]).max() < 1e-9 and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]), (mode, islands, st)
    # far goals (uniformly drawn seeds): steps that fail to improve, so gd_r restarts, and islands whose random starting points win
    seeds, params, _ = make_queries(template, o.active_variables, o.fk_genes, n, seed=43)
    for mode, islands, st in (("gd_r", 1, 40), ("gd_r", 4, 24), ("gd", 8, 12)):
        p = abi.default_solve_params(mode=mode, max_steps=st, islands=islands, random_seed=11)
        a = o.solve_batch(p, orc.RNG_COUNTER, seeds, params, n_threads=4)
        c = h.solve_batch(p, seeds, params)
        assert all(np.array_equal(x, y) for x, y in zip(a, c)), (mode, islands, st, np.abs(a[0] - c[0]).max())
    return b


def goal_sets_beyond_one_goal_per_tip(model, make_solver, whole_solves=True):
    """Several goals on one tip, goals on a link in the middle of the chain, several gene-only primary goals, goals that name variables of their own (a
    JointVariableGoal puts its variable in FRONT of the chains' among the active variables, problem.cpp:103-125: the genes then do not follow the ops) -- the
    reference adds goal after goal to one running sum (problem.cpp:244-257) and its sums over the joint values run over the genes in their order.  The device
    evaluates a tip's goals when the walk completes the tip: listed in that order (the links from the root outwards, gene-only goals behind them) the goals give
    the oracle's fitness and trajectories bit for bit; listed otherwise the same terms are added in another order -- the last bit of a fitness may differ."""
    from bio_ik_amd import (AvoidJointLimitsGoal, CenterJointsGoal, JointVariableGoal, MinimalDisplacementGoal, PoseGoal, PositionGoal, ProblemTemplate,
                            RegularizationGoal)
    elbow = [PositionGoal("r_elbow_flex_link", weight=0.3), PositionGoal("r_elbow_flex_link", (0.1, 0.2, 0.3), weight=0.2)]
    tail = [RegularizationGoal(weight=0.2), CenterJointsGoal(weight=0.1, secondary=False), MinimalDisplacementGoal(weight=0.3)]
    in_order = {
        "two goals on a tip, gene-only goals": elbow + [PoseGoal("r_wrist_roll_link")] + tail,
        "variables of their own": [PoseGoal("r_wrist_roll_link"), JointVariableGoal("r_forearm_roll_joint", 0.3, weight=0.4), RegularizationGoal(weight=0.2),
                                   MinimalDisplacementGoal(weight=0.3, secondary=False), AvoidJointLimitsGoal(weight=0.5, secondary=False),
                                   CenterJointsGoal(weight=0.1, secondary=False)],
        "two of them": [PositionGoal("r_wrist_roll_link"), JointVariableGoal("r_wrist_flex_joint", -0.3, weight=0.4),
                        JointVariableGoal("r_shoulder_lift_joint", 0.2, weight=0.6, secondary=True), CenterJointsGoal(weight=0.3)],
    }
    for name, goals in in_order.items():
        for group in ("right_arm", "all"):
            t = ProblemTemplate(model, group, goals)
            h, o = make_solver(t), orc.Oracle(t)
            function_level(h, o, model, np.random.default_rng(3), n=40, exact_bits=True)
            if whole_solves:
                for mode in ("bio2", "bio2_memetic", "bio2_memetic_l"):
                    trajectory(h, o, t, n=2, pop=16, steps_list=(3,), mode=mode)
                trajectory(h, o, t, n=1, pop=24, steps_list=(3,), fk_mode=abi.FK_LINEAR)
    t = ProblemTemplate(model, "right_arm", [PoseGoal("r_wrist_roll_link")] + elbow + tail)  # (the wrist's goal first: the walk completes the elbow first)
    function_level(make_solver(t), orc.Oracle(t), model, np.random.default_rng(3), n=40, fit_rtol=1e-15)


def branching_hand(make_solver):
    """conftest.hand_robot: three tips behind a common chain (parked frames), goals on the palm in the middle of the chain, on every finger tip, over the joint
    values, a JointVariableGoal -- listed in walk order: fitness, tables and whole solves bit for bit, every mode, both phenotype models, islands.  With the
    arm's third joint PRISMATIC the tip positions agree to the last bit or two instead (the joint program applies origin and slide as one folded constant
    where the reference concatenates frame by frame)."""
    from conftest import hand_robot
    from bio_ik_amd import (AvoidJointLimitsGoal, CenterJointsGoal, JointVariableGoal, LookAtGoal, MinimalDisplacementGoal, OrientationGoal, PoseGoal, PositionGoal,
                            ProblemTemplate)
    model = hand_robot()
    cases = {
        "three tips": [PositionGoal("f0_tip"), PositionGoal("f1_tip", weight=0.8), PoseGoal("f2_tip", weight=0.5)],
        "palm, tips, secondary goals": [OrientationGoal("palm", weight=0.4), PositionGoal("f0_tip"), LookAtGoal("f0_tip", (1, 0, 0), (0.5, 0.1, 0.3), weight=0.3),
                                        PositionGoal("f1_tip", weight=0.8), PositionGoal("f2_tip", weight=0.5), MinimalDisplacementGoal(weight=0.4),
                                        AvoidJointLimitsGoal(weight=0.6)],
        "a link in the middle, a variable of its own": [PositionGoal("a2", weight=0.3), PositionGoal("f1_tip"), PositionGoal("f2_tip"),
                                                        JointVariableGoal("f0a_joint", 0.3, weight=0.5), CenterJointsGoal(weight=0.2)],
    }
    for name, goals in cases.items():
        t = ProblemTemplate(model, "hand", goals)
        h, o = make_solver(t), orc.Oracle(t)
        function_level(h, o, model, np.random.default_rng(3), n=40, exact_bits=True)
        for mode in ("bio2", "bio2_memetic", "bio2_memetic_l"):
            trajectory(h, o, t, n=2, pop=16, steps_list=(3,), mode=mode)
        trajectory(h, o, t, n=2, pop=70, steps_list=(2,))
        trajectory(h, o, t, n=2, pop=24, steps_list=(3,), fk_mode=abi.FK_LINEAR)
        trajectory(h, o, t, n=2, pop=128, steps_list=(2,), islands=2)
    model_p = hand_robot("prismatic")
    t = ProblemTemplate(model_p, "hand", cases["three tips"])
    function_level(make_solver(t), orc.Oracle(t), model_p, np.random.default_rng(3), n=40, frame_tol=5e-16, fit_rtol=1e-14)


def exact_joint_program(make_solver, templates):
    """BIOIK_COMPILE_EXACT=1 (bioik_compile.cpp): the joint program without folding -- an op per origin that is not the identity, the bare joint behind it, the
    reference's frame-by-frame association on ANY robot.  The caller has set the switch.  `gnarly` (rotated origins, oblique axes, a prismatic joint inside a chain,
    fixed links with offsets, three branches, a tip off the root) with a goal of every opcode listed in walk order, and the branching hand with a prismatic joint
    in the middle of its arm: FK, fitness, tables, success test and whole solves bit for bit -- what the folded program reaches to 1e-12 there.  On a benchmark
    fixture, where folding is exact, the two programs give the same bits (through other kernels: the unfolded program is not a serial chain).
    (Until round 5 the GPU suite left the whole solves with LookAtGoal / ConeGoal out: their acos was the device library's.  Since round 6 it is bioik_acos.h's on
    both sides.)"""
    from conftest import gnarly_goals, gnarly_robot, hand_robot
    from bio_ik_amd import PoseGoal, PositionGoal, ProblemTemplate
    g = gnarly_robot()
    order = {n: i for i, n in enumerate(g.link_names)}
    goals = gnarly_goals()
    in_walk_order = sorted([x for x in goals if x.link_name()], key=lambda x: order[x.link_name()]) + [x for x in goals if not x.link_name()]
    hp = hand_robot("prismatic")
    for model, group, gl in ((g, "body", in_walk_order), (g, "body", [PoseGoal("a_tool"), PoseGoal("b_tool"), PositionGoal("head")]),
                             (hp, "hand", [PositionGoal("f0_tip"), PositionGoal("f1_tip", weight=0.8), PoseGoal("f2_tip", weight=0.5)])):
        t = ProblemTemplate(model, group, gl)
        h, o = make_solver(t), orc.Oracle(t)
        function_level(h, o, model, np.random.default_rng(3), n=40, exact_bits=True)
        for mode in ("bio2", "bio2_memetic", "bio2_memetic_l"):
            trajectory(h, o, t, n=2, pop=16, steps_list=(3,), mode=mode)
        trajectory(h, o, t, n=2, pop=24, steps_list=(3,), fk_mode=abi.FK_LINEAR)
    t = templates["c3"]
    trajectory(make_solver(t), orc.Oracle(t), t, n=2, pop=32, steps_list=(3,))


def line_search_on_a_flat_model(make_solver):
    """Quirk Q5 (oracle/orc_evolution.h): when the three support values of the memetic line search are equal the quadratic step is 0 / 0 (ik_evolution_2.cpp:503-507);
    the candidate's genes are NaN, RobotInfo::clip lets a NaN through (utils.h:328-333) and a goal that takes max(0, .) of its error hides it, so the literal algorithm
    can ACCEPT the NaN candidate and return it.  The device takes a candidate with a NaN gene for no candidate and stops the search, as the reference does whenever the
    NaN is not hidden; it never returns one.  The robot and
    the goals are the case tools/robot_fuzz_hostsim.py met (three links, the last joint turns its link about the link's own origin, so a goal on that link's POSITION
    is flat in it): the literal oracle (quirk mode 1) returns NaN genes with a finite fitness, the default oracle (mode 0) and the device the same finite solves,
    bit for bit.  The caller has set BIOIK_COMPILE_EXACT (the last origin is rotated)."""
    from bio_ik_amd import LineGoal, MaxDistanceGoal, PositionGoal, ProblemTemplate, RobotModel
    m = RobotModel("flat")
    m.add_link("l0")
    m.add_link("l1", "l0", "j1", "fixed", xyz=(0.2893761985906587, 0.03248937499002066, -0.014841302929549098))
    m.add_link("l2", "l1", "j2", "revolute", xyz=(0.10755360117858924, 0.02493781640192608, -0.04886061998123816), axis=(0, 1, 0), lower=-0.2803688282418388,
               upper=3.1855786120683693, velocity=2.9325216061470356)
    m.add_link("l3", "l2", "j3", "revolute", xyz=(-0.017080529114468283, 0.2551355563784019, -0.04863453264413984),
               quat=(0.045195500178852516, 0.1273747337154112, 0.2667122008665194, 0.9542524015602207), axis=(0, 0, 1), lower=0.45554072748806607,
               upper=1.4138397555794018, velocity=0.6399125293343063)
    m.add_group("g", joints=["j2", "j3"], tips=["l1", "l2", "l3"])
    goals = [PositionGoal("l1", (-0.5833139686114326, 0.05236954534023761, -0.48273822641048036), weight=0.3),
             PositionGoal("l1", (0.2359118231308469, 0.3688053123007919, 0.10831832574337653), weight=1.6),
             LineGoal("l2", (-0.162322900664759, -0.27458126148279965, -0.04181421604763092), (0.774617869644533, -0.08365929393599635, -0.6268718198846522), weight=1.6),
             MaxDistanceGoal("l3", (-0.15454564414528751, -0.10852275777981442, -0.02007615363449216), 0.3)]
    t = ProblemTemplate(m, "g", goals)
    o, h = orc.Oracle(t), make_solver(t)
    seeds, params, _ = make_queries(t, o.active_variables, o.fk_genes, 2, seed=570)
    p = abi.default_solve_params(population=8, max_steps=3, random_seed=11, mode="bio2_memetic")
    orc.set_quirk_mode(1)
    try:
        literal = o.solve_batch(p, orc.RNG_COUNTER, seeds, params, n_threads=1)
    finally:
        orc.set_quirk_mode(0)
    assert np.isnan(literal[0]).any() and np.isfinite(literal[1]).all()
    want, got = o.solve_batch(p, orc.RNG_COUNTER, seeds, params, n_threads=1), h.solve_batch(p, seeds, params)
    assert np.isfinite(want[0]).all()
    assert all(np.array_equal(a, b) for a, b in zip(want, got))


def line_search_step_without_bound(make_solver):
    """Quirk Q7 (oracle/orc_evolution.h): a line search whose three support values have a slope but no curvature steps by v / 0 = infinity (ik_evolution_2.cpp:498-507);
    RobotInfo::clip puts a joint WITHOUT limits at +-DBL_MAX (utils.h:328-333, robot_info.h:109-113), the linear model is evaluated at 1.8e308, and where a goal
    hides the overflow the candidate is ACCEPTED: the literal algorithm returns a joint value of 1.8e308.  Here: the case tools/robot_fuzz_hostsim.py met (its robot
    7252 of seed 1001): a link on a continuous joint under a ConeGoal without a position term -- acos of a NaN is NaN, max(0, NaN - angle) = 0.  The literal oracle
    (quirk mode 1) returns +-DBL_MAX; the default oracle and the device take such a candidate for none, return the same finite solves bit for bit, and no joint
    value of magnitude 1e300 or more ever leaves the product.  The caller has set BIOIK_COMPILE_EXACT (rotated origins)."""
    from bio_ik_amd import ConeGoal, ProblemTemplate, RobotModel
    m = RobotModel("unbounded")
    m.add_link("l0")
    m.add_link("l1", "l0", "j1", "continuous", xyz=(0.0, 0.0, 0.0), rpy=(0.0, 0.0, 0.0), axis=(1.0, 0.0, 0.0), velocity=0.5426426563843417)
    m.add_link("l2", "l1", "j2", "revolute", xyz=(0.04572965481048473, 0.05277106152021511, 0.05050941592637754), rpy=(-0.3608740952450565, 0.980351570493485, -1.22862373643628),
               axis=(0.9541379120674467, 0.013104401555596383, -0.2990804564250949), lower=-2.2247343186350625, upper=3.2019406218963327, velocity=1.182617098045548)
    m.add_link("l3", "l2", "j3", "revolute", xyz=(0.005954400579178779, -0.08672517736579491, -0.04541483446181143), rpy=(0.0, 0.0, 0.0),
               axis=(-0.24200882816021646, -0.9626430338070318, 0.12145005786459118), lower=-0.7251913054947542, upper=0.053860341485142835, velocity=1.7665671554542268)
    m.add_link("l4", "l3", "j4", "fixed", xyz=(0.24972201631502833, -0.25755953675525123, 0.26952303216126744), rpy=(-0.042157648764034404, -1.0784397959920873, 0.08927858511789348))
    m.add_link("l5", "l1", "j5", "revolute", xyz=(0.17325897707111526, 0.1524531895817025, -0.11564002553074237), rpy=(-0.7542471072117699, 0.280108269925566, 0.32269292633260394),
               axis=(0.0, 0.0, 1.0), lower=1.5690686444480715, upper=1.8057083350514227, velocity=1.2485997210011077)
    m.add_link("l6", "l5", "j6", "continuous", xyz=(0.059373342160621435, 0.033905455876769394, 0.22673021365372945), rpy=(0.8700779369263824, 0.647723143901797, 0.9600318977820447),
               axis=(0.0, 0.0, 1.0), velocity=1.0493670281090843)
    m.add_link("l7", "l6", "j7", "prismatic", xyz=(0.0, 0.0, 0.0), rpy=(0.0, 0.0, 0.0), axis=(0.0, 0.0, 1.0), lower=-0.24420766820072295, upper=0.31795360850378185, velocity=2.567223066835506)
    m.add_link("l8", "l6", "j8", "revolute", xyz=(0.3110930684668922, 0.21261168225543117, -0.07030637061651651), rpy=(0.0, 0.0, 0.0),
               axis=(-0.5912632433391664, -0.7366574365173285, 0.3282431999292108), lower=-0.26411770069874596, upper=1.3269122675011662, velocity=2.9448390157031685)
    m.add_link("l9", "l8", "j9", "revolute", xyz=(0.00858138555622114, -0.10387504668282581, 0.1110967330723456), rpy=(0.0, 0.0, 0.0),
               axis=(0.8810341758029753, 0.07523516021682576, 0.46703153184160984), lower=-2.9528953897285146, upper=1.794084985679735, velocity=2.1667318473448707)
    m.add_link("l10", "l3", "j10", "fixed", xyz=(0.03843744082044307, 0.09418822225340512, -0.03241751247524506), rpy=(-0.28941665223466384, -0.8663386046162123, -1.0747080865501497))
    m.add_group("g", joints=["j1", "j2", "j3", "j5", "j6", "j7", "j8", "j9"], tips=["l1"])
    goals = [ConeGoal("l1", axis=(0.34061279580669074, 0.10192115422766176, -0.9346630417715525), direction=(-0.24861241457165142, 0.9594718696304163, 0.13268609086398958),
                      angle=0.4, weight=1.6)]
    t = ProblemTemplate(m, "g", goals)
    o, h = orc.Oracle(t), make_solver(t)
    seeds, params, _ = make_queries(t, o.active_variables, o.fk_genes, 2, seed=7252)
    p = abi.default_solve_params(population=8, max_steps=2, random_seed=11, mode="bio2_memetic")
    orc.set_quirk_mode(1)
    try:
        literal = o.solve_batch(p, orc.RNG_COUNTER, seeds, params, n_threads=1)
    finally:
        orc.set_quirk_mode(0)
    assert (np.abs(literal[0]) >= 1e300).any()  # the reference's behaviour: a joint value of +-DBL_MAX as the solution
    met = orc.unbounded_candidates()
    want, got = o.solve_batch(p, orc.RNG_COUNTER, seeds, params, n_threads=1), h.solve_batch(p, seeds, params)
    assert orc.unbounded_candidates() > met  # (the default mode met such candidates too -- and dropped them)
    assert np.isfinite(want[0]).all() and np.abs(want[0]).max() < 1e300
    assert np.isfinite(got[0]).all() and np.abs(got[0]).max() < 1e300
    assert all(np.array_equal(a, b) for a, b in zip(want, got))
    # ... and over a longer search with more streams: nothing of magnitude 1e300 ever comes back from the device
    seeds, params, _ = make_queries(t, o.active_variables, o.fk_genes, 32, seed=99)
    for mode in ("bio2_memetic", "bio2_memetic_l"):
        sol = h.solve_batch(abi.default_solve_params(population=16, max_steps=12, random_seed=5, mode=mode), seeds, params)[0]
        assert np.isfinite(sol).all() and np.abs(sol).max() < 1e300


def island_selection_three_ways(h, template, monkeypatch, n=3, pop=16, steps=5, fk_mode=None, kind="global", noise=0.1, configs=((9, 1, False), (64, 1, False), (100, 0, False), (100, 1, True), (70, 1, False))):
    """The best island of every query (ik_parallel.h:220-269) three ways -- by the query's last island inside the solve's launch (a wavefront: select_coop), by a
    wavefront per query in a launch of its own (k_select_wave), by a lane per query walking the islands (k_select: the loop that restates the reference) -- on solves
    whose islands pass at different steps, pass not at all (the fallback: least fitness of all), and number more than a wavefront has lanes: the same answer."""
    from bio_ik_amd import abi
    from bio_ik_amd.workload import make_queries
    import numpy as np
    seeds, params, _ = make_queries(template, h.active_variables, h.fk_genes, n, seed=77, kind=kind, noise=noise)
    far = params.copy()
    far[:, 0:3] += 10.0  # out of reach: no island ever passes
    kw = {} if fk_mode is None else {"fk_mode": fk_mode}
    for islands, sync, unreachable in configs:
        par = far if unreachable else params
        res = []
        for mode in ("1", "0", "-1"):
            monkeypatch.setenv("BIOIK_SOLVE_FUSED_SELECT", mode)
            res.append(h.solve_batch(abi.default_solve_params(population=pop, max_steps=steps, random_seed=11, islands=islands, island_sync=sync, **kw), seeds, par))
        monkeypatch.delenv("BIOIK_SOLVE_FUSED_SELECT")
        for r in res[1:]:
            assert all(np.array_equal(x, y) for x, y in zip(res[0], r)), (islands, sync)
        assert not res[0][2].any() if unreachable else res[0][2].any(), (islands, sync, res[0][2])  # (the rule for passing islands / the fallback is what ran)
