"""The Python face of the plugin (bio_ik_amd/plugin.py over bio_ik/plugin_core.h) against the reference's searchPositionIK contract
(src/kinematics_plugin.cpp:437-655), driven on CPU through the host simulator of the kernels."""
import numpy as np
import pytest

from bio_ik_amd import (BioIKKinematicsPlugin, BioIKKinematicsQueryOptions, KinematicsQueryOptions, MoveItErrorCodes, PositionGoal, abi)
from bio_ik_amd.robot import frame_concat, link_transform
from conftest import random_configuration
from oracle import orc


@pytest.fixture(scope="module")
def plugin(hostsim_shim, pr2):
    p = BioIKKinematicsPlugin(lib=hostsim_shim)
    assert p.initialize(pr2, "right_arm", "torso_lift_link", ["r_wrist_roll_link"], 0.0,
                        params={"gpu_population": 16, "gpu_fk": "linear", "gpu_max_steps": 40, "random_seed": 3, "gpu_reproducible_calls": True})
    return p


def goal_in_base_frame(pr2, target_state):
    """pose of the tip expressed in the group's base frame, as a MoveIt caller would pass it"""
    tip = link_transform(pr2, pr2.link_index("r_wrist_roll_link"), target_state)
    base = link_transform(pr2, pr2.link_index("torso_lift_link"), pr2.default_positions())
    inv_q = np.array([-base[3], -base[4], -base[5], base[6]])
    from bio_ik_amd.robot import quat_multiply, quat_rotate
    return np.concatenate([quat_rotate(inv_q, tip[:3] - base[:3]), quat_multiply(inv_q, tip[3:])])


def test_interface_shape(plugin):
    assert plugin.getJointNames()[0] == "r_shoulder_pan_joint" and len(plugin.getJointNames()) == 7
    assert plugin.getLinkNames() == ["r_wrist_roll_link"]
    assert plugin.getPositionFK([], [], []) is False and plugin.getPositionIK(None, [], [], MoveItErrorCodes()) is False
    assert plugin.supportsGroup(None)
    with pytest.raises(RuntimeError):
        BioIKKinematicsPlugin(lib=plugin._lib_path).initialize(plugin.robot_model, "right_arm", "torso_lift_link", ["r_wrist_roll_link"], params={"mode": "nonsense"})


def test_search_position_ik_round_trip(plugin, pr2):
    rng = np.random.default_rng(5)
    target = pr2.default_positions()
    gv = plugin._group_vars
    target[gv] = random_configuration(pr2, rng)[gv]
    pose = goal_in_base_frame(pr2, target)
    seed = np.clip(target[gv] + 0.2 * rng.normal(size=len(gv)), np.asarray(pr2.var_min)[gv], np.asarray(pr2.var_max)[gv])
    solution, code = [], MoveItErrorCodes()
    assert plugin.searchPositionIK([pose], list(seed), 60.0, solution, code) is True
    assert code.val == MoveItErrorCodes.SUCCESS and len(solution) == 7
    state = pr2.default_positions()
    state[gv] = solution
    got = link_transform(pr2, pr2.link_index("r_wrist_roll_link"), state)
    want = frame_concat(link_transform(pr2, pr2.link_index("torso_lift_link"), pr2.default_positions()), pose)
    assert np.linalg.norm(got[:3] - want[:3]) < 1e-4 and 2 * np.arccos(min(1.0, abs(got[3:] @ want[3:]))) < 1e-3
    lo, hi = np.asarray(pr2.var_min)[gv], np.asarray(pr2.var_max)[gv]
    assert np.all(np.asarray(solution) >= lo - 1e-12) and np.all(np.asarray(solution) <= hi + 1e-12)
    # callback semantics (:644-649): the callback's verdict is the return value
    def reject(p, s, e):
        e.val = MoveItErrorCodes.NO_IK_SOLUTION
    assert plugin.searchPositionIK([pose], list(seed), 60.0, [], MoveItErrorCodes(), solution_callback=reject) is False


def test_pose_quaternion_is_normalised_like_set_orientation(plugin, pr2):
    """geometry_msgs orientations are rarely exactly unit: PoseGoal::setOrientation normalises (goal_types.h:146, called at
    kinematics_plugin.cpp:543-544), so a scaled quaternion must give the same solve as the unit one"""
    rng = np.random.default_rng(12)
    target = pr2.default_positions()
    gv = plugin._group_vars
    target[gv] = random_configuration(pr2, rng)[gv]
    pose = goal_in_base_frame(pr2, target)
    seed = list(np.clip(target[gv] + 0.2 * rng.normal(size=len(gv)), np.asarray(pr2.var_min)[gv], np.asarray(pr2.var_max)[gv]))
    scaled = pose.copy()
    scaled[3:] *= 2.0
    a, b = [], []
    assert plugin.searchPositionIK([pose], seed, 60.0, a, MoveItErrorCodes()) is True
    assert plugin.searchPositionIK([scaled], seed, 60.0, b, MoveItErrorCodes()) is True
    assert np.allclose(a, b, atol=1e-9)


def test_unreachable_goal_error_codes(plugin, pr2):
    far = np.array([5.0, 5.0, 5.0, 0, 0, 0, 1.0])
    seed = list(pr2.default_positions()[plugin._group_vars])
    plugin.params["gpu_max_steps"] = 2
    try:
        sol, code = [], MoveItErrorCodes()
        assert plugin.searchPositionIK([far], seed, 60.0, sol, code) is False and code.val == MoveItErrorCodes.NO_IK_SOLUTION  # :638-641
        sol, code = [], MoveItErrorCodes()
        assert plugin.searchPositionIK([far], seed, 60.0, sol, code, options=KinematicsQueryOptions(return_approximate_solution=True)) is True
        assert len(sol) == 7
        opts = BioIKKinematicsQueryOptions()  # replace: only the caller's goals (:540-556)
        opts.replace = True
        opts.return_approximate_solution = True
        opts.goals.append(PositionGoal("r_wrist_roll_link", (0.5, -0.2, 0.9)))
        assert plugin.searchPositionIK([], seed, 60.0, sol, MoveItErrorCodes(), options=opts) is True
        assert opts.solution_fitness >= 0.0
    finally:
        plugin.params["gpu_max_steps"] = 40


def test_angle_wrapping_matches_oracle(plugin, oracles, pr2):
    """kinematics_plugin.cpp:580-616: the plugin core's post-processing (bio_ik/plugin_core.h, through the shim) against the oracle's restatement"""
    o = oracles["c2"]
    rng = np.random.default_rng(9)
    seed = random_configuration(pr2, rng, 64)
    state = seed + rng.normal(size=seed.shape) * 7.0
    want = np.stack([o.wrap_angles(seed[k], state[k]) for k in range(64)])
    got = plugin._wrap_angles(state, seed, o.active_variables)
    act = o.active_variables
    assert np.abs(got[:, act] - want[:, act]).max() < 1e-12


def test_factory_names_of_the_gradient_solvers(hostsim_shim, pr2):
    """IKFactory names of src/ik_gradient.cpp:254-292 as the yaml `mode`: gd / gd_r / gd_c / jac and their _2 / _4 / _8 forms (N solver
    threads, the further ones started at random configurations = N islands); bio1 and the optimiser-library names are configuration errors"""
    rng = np.random.default_rng(3)
    target = pr2.default_positions()
    for mode, want_success in (("jac", True), ("jac_4", True), ("gd_r_2", False), ("gd_c_8", False), ("gd", False)):
        p = BioIKKinematicsPlugin(lib=hostsim_shim)
        assert p.initialize(pr2, "right_arm", "torso_lift_link", ["r_wrist_roll_link"], 0.0, params={"mode": mode, "gpu_max_steps": 12, "gpu_reproducible_calls": True})
        gv = p._group_vars
        target[gv] = random_configuration(pr2, rng)[gv]
        pose = goal_in_base_frame(pr2, target)
        seed = list(np.clip(target[gv] + 0.05 * rng.normal(size=len(gv)), np.asarray(pr2.var_min)[gv], np.asarray(pr2.var_max)[gv]))
        sol, code = [], MoveItErrorCodes()
        ok = p.searchPositionIK([pose], seed, 0.0, sol, code, options=KinematicsQueryOptions(return_approximate_solution=not want_success))
        assert ok and len(sol) == 7, mode
        p.close()
    for mode in ("bio1", "gd_3", "jac_16", "optlib_bfgs"):
        with pytest.raises(RuntimeError):
            BioIKKinematicsPlugin(lib=hostsim_shim).initialize(pr2, "right_arm", "torso_lift_link", ["r_wrist_roll_link"], params={"mode": mode})


def test_gpu_schedule_key(hostsim_shim, pr2):
    """gpu_schedule: "auto" (default) | "latency" | "throughput" (bioik_solve_params::schedule); the answer does not depend on it, anything else
    is a configuration error"""
    rng = np.random.default_rng(4)
    target = pr2.default_positions()
    sols = []
    for schedule in ("latency", "throughput", "auto"):
        p = BioIKKinematicsPlugin(lib=hostsim_shim)
        assert p.initialize(pr2, "right_arm", "torso_lift_link", ["r_wrist_roll_link"], 0.0,
                            params={"gpu_schedule": schedule, "gpu_population": 128, "gpu_max_steps": 6, "random_seed": 2, "gpu_reproducible_calls": True})
        if not sols:
            gv = p._group_vars
            target[gv] = random_configuration(pr2, rng)[gv]
            pose = goal_in_base_frame(pr2, target)
            seed = list(np.clip(target[gv] + 0.3 * rng.normal(size=len(gv)), np.asarray(pr2.var_min)[gv], np.asarray(pr2.var_max)[gv]))
        sol = []
        p.searchPositionIK([pose], seed, 0.0, sol, MoveItErrorCodes(), options=KinematicsQueryOptions(return_approximate_solution=True))
        sols.append(sol)
        p.close()
    assert sols[0] == sols[1] == sols[2]
    with pytest.raises(RuntimeError):
        BioIKKinematicsPlugin(lib=hostsim_shim).initialize(pr2, "right_arm", "torso_lift_link", ["r_wrist_roll_link"], params={"gpu_schedule": "fast"})


def test_product_shim_exports():
    """libbio_ik_shim.so (linked against libbioik_hip.so) loads without a GPU and exports what plugin.py binds"""
    import subprocess, os
    from bio_ik_amd import plugin as pl
    subprocess.run(["make", "-C", os.path.join(os.path.dirname(pl.SHIM_PATH)), "-s", "shim"], check=True)
    L = pl.load_shim()
    for name in ("bioik_plugin_create", "bioik_plugin_destroy", "bioik_plugin_update", "bioik_plugin_submit", "bioik_plugin_wait", "bioik_plugin_postprocess",
                 "bioik_plugin_group_variables", "bioik_plugin_group_variable_count", "bioik_plugin_last_error"):
        assert hasattr(L, name)


@pytest.mark.gpu
def test_python_plugin_on_the_device(pr2):
    """the Python face over the product libraries: a batch through searchPositionIKBatch, two batches in flight through the submit / wait
    form, every reported success reproducing its pose; the async form returns what the synchronous one does"""
    p = BioIKKinematicsPlugin()
    assert p.initialize(pr2, "right_arm", "torso_lift_link", ["r_wrist_roll_link"], 0.0, params={"gpu_max_steps": 100, "random_seed": 5, "gpu_reproducible_calls": True})
    rng = np.random.default_rng(21)
    gv = p._group_vars
    n = 256
    targets = np.tile(pr2.default_positions(), (n, 1))
    targets[:, gv] = random_configuration(pr2, rng, n)[:, gv]
    poses = np.stack([goal_in_base_frame(pr2, t) for t in targets]).reshape(n, 1, 7)
    seeds = random_configuration(pr2, rng, n)[:, gv]
    sols, ok, fit, codes = p.searchPositionIKBatch(poses, seeds)
    assert ok.mean() > 0.9 and np.all(codes[ok] == MoveItErrorCodes.SUCCESS) and np.all(codes[~ok] == MoveItErrorCodes.NO_IK_SOLUTION)
    base = link_transform(pr2, pr2.link_index("torso_lift_link"), pr2.default_positions())
    for k in np.nonzero(ok)[0][:64]:
        state = pr2.default_positions()
        state[gv] = sols[k]
        got = link_transform(pr2, pr2.link_index("r_wrist_roll_link"), state)
        want = frame_concat(base, poses[k, 0])
        assert np.linalg.norm(got[:3] - want[:3]) < 1e-4 and 2 * np.arccos(min(1.0, abs(got[3:] @ want[3:]))) < 1e-3
    # (the plugin's default gpu_islands = 0 sizes the islands to the call -- sixteen per query for 128 queries, eight for 256 --, so a call is compared
    # with a call of its own size)
    sols128, ok128, _, _ = p.searchPositionIKBatch(poses[:128], seeds[:128])
    a = p.searchPositionIKBatchAsync(poses[:128], seeds[:128])
    b = p.searchPositionIKBatchAsync(poses[:128], seeds[:128])
    ra, rb = p.searchPositionIKBatchWait(a), p.searchPositionIKBatchWait(b)
    assert np.array_equal(ra[0], sols128) and np.array_equal(rb[0], sols128) and np.array_equal(ra[1], ok128)
    p.close()
