"""Plugin-shaped front end: a mirror of `bio_ik_kinematics_plugin::BioIKKinematicsPlugin`
(reference src/kinematics_plugin.cpp:117-671) over the HIP solver, plus the batched `searchPositionIKBatch`.

Same method names, argument meaning and error behaviour as the reference's `kinematics::KinematicsBase` implementation,
with Python stand-ins for the ROS message types: a pose is 7 numbers (position xyz, orientation xyzw, like
geometry_msgs::Pose), `MoveItErrorCodes.val` carries SUCCESS / NO_IK_SOLUTION, `KinematicsQueryOptions` /
`BioIKKinematicsQueryOptions` (goals.py) carry `return_approximate_solution`, `goals`, `fixed_joints`, `replace`.
Parameters are the reference's kinematics.yaml keys (`kinematics_plugin.cpp:243-328`) plus the additive `gpu_*` keys.
"""
import threading

import numpy as np

from . import abi
from .goals import AvoidJointLimitsGoal, BioIKKinematicsQueryOptions, CenterJointsGoal, MinimalDisplacementGoal, PoseGoal
from .problem import ProblemTemplate
from .robot import frame_concat, link_transform
from .solver import HipSolver, device_count


class MoveItErrorCodes:
    SUCCESS = 1
    NO_IK_SOLUTION = -31

    def __init__(self):
        self.val = 0


class KinematicsQueryOptions:
    def __init__(self, return_approximate_solution=False):
        self.return_approximate_solution = return_approximate_solution


DEFAULT_PARAMS = {
    "mode": "bio2_memetic", "random_seed": 0, "dpos": -1.0, "drot": -1.0, "dtwist": 1e-5, "no_wipeout": False,
    "rotation_scale": 0.5, "position_only_ik": False, "center_joints_weight": 0.0, "avoid_joint_limits_weight": 0.0,
    "minimal_displacement_weight": 0.0,
    # additive keys of the GPU build
    "gpu_population": 128, "gpu_fk": "exact", "gpu_islands": 1, "gpu_max_steps": 64, "gpu_devices": None,
}


class BioIKKinematicsPlugin:
    def __init__(self, solver_factory=None):
        self._solver_factory = solver_factory or (lambda template, device: HipSolver(template, device=device))
        self._solvers = {}
        self.robot_model = None

    # ---- kinematics::KinematicsBase ---------------------------------------------------------------------------
    def initialize(self, robot_model, group_name, base_frame, tip_frames, search_discretization=0.0, params=None):
        """kinematics_plugin.cpp:337-374 (the RobotModel overload); returns True like the reference."""
        self.robot_model = robot_model
        self.group_name = group_name
        self.base_frame = base_frame
        self.tip_frames = [tip_frames] if isinstance(tip_frames, str) else list(tip_frames)
        self.params = dict(DEFAULT_PARAMS)
        self.params.update(params or {})
        if self.params["mode"] not in abi.MODE_BY_NAME:
            raise RuntimeError("unknown solver mode %r" % self.params["mode"])  # IKFactory::create -> ERROR (utils.h:436)
        group = robot_model.groups[group_name]
        self.joint_names = [robot_model.joint_names[j] for j in group.active_joints]  # kinematics_plugin.cpp:226-231
        self.link_names = list(self.tip_frames)
        self._group_vars = []
        for j in group.active_joints:
            fv = robot_model.joint_first_variable[j]
            self._group_vars.extend(range(fv, fv + abi.JOINT_VAR_COUNT[robot_model.joint_type[j]]))
        # default goals, kinematics_plugin.cpp:279-329
        rs = 0.0 if self.params["position_only_ik"] else self.params["rotation_scale"]
        self.default_goals = []
        for tip in self.tip_frames:
            g = PoseGoal(tip)
            g.setRotationScale(rs)
            self.default_goals.append(g)
        for key, cls in (("center_joints_weight", CenterJointsGoal), ("avoid_joint_limits_weight", AvoidJointLimitsGoal),
                         ("minimal_displacement_weight", MinimalDisplacementGoal)):
            if self.params[key] > 0.0:
                g = cls()
                g.setWeight(self.params[key])
                self.default_goals.append(g)
        self._base_link = robot_model.link_index(base_frame)
        self._base_default = link_transform(robot_model, self._base_link, robot_model.default_positions())
        info = np.zeros((robot_model.n_variables, 3))
        self._lo = np.asarray(robot_model.var_min, dtype=np.float64)
        self._hi = np.asarray(robot_model.var_max, dtype=np.float64)
        self._revolute = np.array([robot_model.joint_type[self._joint_of_var(v)] == abi.JOINT_REVOLUTE for v in range(robot_model.n_variables)])
        # MoveIt decides clamp-or-wrap by the variable's position_bounded flag (continuous joints are unbounded), not by the width of its
        # limits (that rule belongs to RobotInfo's clip range, robot_info.h:82-90): a revolute joint with limits of +-3.2 rad is clamped
        self._bounded = np.asarray(robot_model.var_bounded, dtype=bool)
        self._has_mimic = any(m >= 0 for m in robot_model.joint_mimic)
        return True

    def _joint_of_var(self, v):
        m = self.robot_model
        for j in range(m.n_links):
            fv = m.joint_first_variable[j]
            if fv >= 0 and fv <= v < fv + abi.JOINT_VAR_COUNT[m.joint_type[j]]:
                return j
        raise KeyError(v)

    def getJointNames(self):
        return self.joint_names

    def getLinkNames(self):
        return self.link_names

    def getPositionFK(self, link_names, joint_angles, poses):
        return False  # kinematics_plugin.cpp:140-145

    def getPositionIK(self, ik_pose, ik_seed_state, solution, error_code, options=None):
        return False  # kinematics_plugin.cpp:147-155

    def supportsGroup(self, jmg, error_text_out=None):
        return True  # kinematics_plugin.cpp:657-662

    # ---- problem / solver cache ----------------------------------------------------------------------------------
    def _solve_params(self, timeout=0.0):
        p = self.params
        return abi.default_solve_params(timeout=float(timeout) if timeout and timeout > 0.0 else 0.0,
                                        mode=p["mode"], fk_mode=abi.FK_EXACT if p["gpu_fk"] == "exact" else abi.FK_LINEAR,
                                        population=int(p["gpu_population"]), islands=int(p["gpu_islands"]), max_steps=int(p["gpu_max_steps"]),
                                        random_seed=int(p["random_seed"]), dpos=float(p["dpos"]), drot=float(p["drot"]), dtwist=float(p["dtwist"]),
                                        no_wipeout=1 if p["no_wipeout"] else 0)

    def _all_goals(self, options):
        bio = options if isinstance(options, BioIKKinematicsQueryOptions) else None
        goals = []
        if not bio or not bio.replace:
            goals.extend(self.default_goals)  # kinematics_plugin.cpp:550-552
        if bio:
            goals.extend(bio.goals)
        return goals, (list(bio.fixed_joints) if bio else [])

    def _solver_for(self, goals, fixed_joints, device):
        key = (device, tuple(fixed_joints), tuple((type(g).__name__, g.link_name(), g.variable_name(), g.getWeight(), g.isSecondary()) for g in goals))
        if key not in self._solvers:
            template = ProblemTemplate(self.robot_model, self.group_name, goals, fixed_joints)
            self._solvers[key] = (template, self._solver_factory(template, device))
        return self._solvers[key]

    # ---- the batched entry point (new) and the reference's single-query one ---------------------------------------------
    def searchPositionIKBatch(self, ik_poses, ik_seed_states, options=None, context_states=None, devices=None, first_query=0, timeout=0.0):
        """n independent queries sharing one goal structure.  `timeout` [s] bounds the whole call on the device clock
        (ik_parallel.h:160; every query still runs one step), <= 0: only the step budget gpu_max_steps applies.
        ik_poses [n][tips][7] (ignored when options.replace), ik_seed_states [n][group variables]
        -> (solutions [n][group variables], success [n] bool, fitness [n], error codes [n])."""
        options = options or KinematicsQueryOptions()
        m = self.robot_model
        seeds_g = np.asarray(ik_seed_states, dtype=np.float64).reshape(-1, len(self._group_vars))
        n = seeds_g.shape[0]
        goals, fixed = self._all_goals(options)
        replace = isinstance(options, BioIKKinematicsQueryOptions) and options.replace
        # seed -> full state, kinematics_plugin.cpp:465-485
        if context_states is not None:
            state = np.array(context_states, dtype=np.float64).reshape(n, m.n_variables)
        else:
            state = np.tile(m.default_positions(), (n, 1))
        state[:, self._group_vars] = seeds_g
        devices = list(devices) if devices is not None else (self.params["gpu_devices"] or [0])
        template, _ = self._solver_for(goals, fixed, devices[0])
        # per-query goal numbers: default pose goals in the model frame (:487-502, :540-546), the rest from the goal objects
        params = np.tile(template.pack_params(goals), (n, 1))
        if not replace:
            poses = np.asarray(ik_poses, dtype=np.float64).reshape(n, len(self.tip_frames), 7)
            for i in range(len(self.tip_frames)):
                off = template.param_offsets[i]
                for k in range(n):
                    r = self._base_default if context_states is None else link_transform(m, self._base_link, state[k])
                    f = frame_concat(r, poses[k, i])
                    q = f[3:7]
                    f[3:7] = q * (1.0 / np.sqrt(float(q @ q)))  # PoseGoal::setOrientation (goal_types.h:146, tf2 normalized()), :543-544
                    params[k, off:off + 7] = f
        sp = self._solve_params(timeout)
        # one C-ABI call: a single device, or contiguous shards over the listed devices (bioik_solve_batch_multi: one host thread and stream
        # per device inside the library, query-indexed RNG streams -> identical to the unsharded solve)
        handles = [self._solver_for(goals, fixed, d)[1] for d in devices]
        handles[0].set_first_query(first_query)
        if len(handles) == 1:
            sol, fit, suc, _ = handles[0].solve_batch(sp, state, params)
        else:
            sol, fit, suc, _ = handles[0].solve_batch_multi(handles[1:], sp, state, params)
        handles[0].set_first_query(0)
        active = self._solver_for(goals, fixed, devices[0])[1].active_variables
        sol = self._wrap_angles(sol, state, active)
        solutions = sol[:, self._group_vars]  # kinematics_plugin.cpp:619-629
        ok = (suc != 0) | bool(getattr(options, "return_approximate_solution", False))  # :638-641
        codes = np.where(ok, MoveItErrorCodes.SUCCESS, MoveItErrorCodes.NO_IK_SOLUTION)
        if isinstance(options, BioIKKinematicsQueryOptions) and n:
            options.solution_fitness = float(fit[-1])  # :632-634
        return solutions, ok, fit, codes

    def searchPositionIK(self, ik_poses, ik_seed_state, timeout, solution, error_code, options=None, consistency_limits=None,
                         solution_callback=None, context_state=None):
        """kinematics_plugin.cpp:437-655.  `timeout` [s] is the reference's wall-clock budget (:504, :574; honoured on the device,
        at least one step); `gpu_max_steps` bounds the call as well."""
        poses = np.asarray(ik_poses, dtype=np.float64).reshape(1, -1, 7) if len(ik_poses) else np.zeros((1, 0, 7))
        sols, ok, fit, codes = self.searchPositionIKBatch(poses, [ik_seed_state], options, None if context_state is None else [context_state],
                                                          timeout=timeout)
        solution[:] = list(sols[0])
        if not ok[0]:
            error_code.val = MoveItErrorCodes.NO_IK_SOLUTION
            return False
        if solution_callback is not None:  # :644-649
            solution_callback(ik_poses[0] if len(ik_poses) else None, solution, error_code)
            return error_code.val == MoveItErrorCodes.SUCCESS
        error_code.val = MoveItErrorCodes.SUCCESS
        return True

    # ---- kinematics_plugin.cpp:580-616 -------------------------------------------------------------------------------
    def _wrap_angles(self, sol, seed, active):
        sol = sol.copy()
        two_pi = 2 * np.pi
        if not self._has_mimic:
            for ivar in active:
                if not self._revolute[ivar]:
                    continue
                v, r = sol[:, ivar].copy(), seed[:, ivar]
                lo, hi = self._lo[ivar], self._hi[ivar]
                far = (r < v - np.pi) | (r > v + np.pi)
                w = (v - r) / two_pi + 0.5
                w = (w - np.floor(w) - 0.5) * two_pi + r
                v = np.where(far, w, v)
                v = np.where(v > hi, v - np.ceil(np.maximum(0.0, v - hi) / two_pi) * two_pi, v)
                v = np.where(v < lo, v + np.ceil(np.maximum(0.0, lo - v) / two_pi) * two_pi, v)
                sol[:, ivar] = np.clip(v, lo, hi)
        # RobotModel::enforcePositionBounds (:616)
        b = self._bounded
        sol[:, b] = np.clip(sol[:, b], self._lo[b], self._hi[b])
        cont = self._revolute & ~self._bounded
        x = sol[:, cont]
        out = (x < -np.pi) | (x > np.pi)
        y = np.fmod(x + np.pi, two_pi)
        y = np.where(y < 0.0, y + two_pi, y) - np.pi
        sol[:, cont] = np.where(out, y, x)
        return sol


def visible_devices():
    return list(range(device_count()))
