"""Plugin-shaped front end: `bio_ik_kinematics_plugin::BioIKKinematicsPlugin` (reference src/kinematics_plugin.cpp:117-671)
for Python callers, plus the batched `searchPositionIKBatch` and its submit / wait form.

Same method names, argument meaning and error behaviour as the reference's `kinematics::KinematicsBase` implementation,
with Python stand-ins for the ROS message types: a pose is 7 numbers (position xyz, orientation xyzw, like
geometry_msgs::Pose), `MoveItErrorCodes.val` carries SUCCESS / NO_IK_SOLUTION, `KinematicsQueryOptions` /
`BioIKKinematicsQueryOptions` (goals.py) carry `return_approximate_solution`, `goals`, `fixed_joints`, `replace`.
Parameters are the reference's kinematics.yaml keys (`kinematics_plugin.cpp:243-328`) plus the additive `gpu_*` keys.

This class is a FACE of the one plugin implementation, `bio_ik/plugin_core.h` (the MoveIt plugin and the header-only C++ class
are the other two): what happens around the solver call -- seed states over the context state, default goals, tip poses into the
model frame, the angle wrap, the position bounds, the error codes (:465-641) -- is `core::Engine`, bound through the C shim
`cpp/src/plugin_shim.cpp` (libbio_ik_shim.so, ctypes).  Here: argument conversion only.
"""
import ctypes as C
import os

import numpy as np

from . import abi
from .goals import BioIKKinematicsQueryOptions
from .robot import link_transform
from .solver import device_count, load_library

SHIM_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "libbio_ik_shim.so")


class _Settings(C.Structure):  # bioik_plugin_settings (cpp/src/plugin_shim.cpp)
    _fields_ = [("mode", C.c_char_p), ("gpu_fk", C.c_char_p), ("gpu_schedule", C.c_char_p), ("random_seed", C.c_int32), ("no_wipeout", C.c_int32),
                ("position_only_ik", C.c_int32), ("gpu_population", C.c_int32), ("gpu_islands", C.c_int32), ("gpu_max_steps", C.c_int32),
                ("gpu_reproducible_calls", C.c_int32), ("n_devices", C.c_int32), ("devices", C.POINTER(C.c_int32)),
                ("dpos", C.c_double), ("drot", C.c_double), ("dtwist", C.c_double), ("rotation_scale", C.c_double),
                ("center_joints_weight", C.c_double), ("avoid_joint_limits_weight", C.c_double), ("minimal_displacement_weight", C.c_double)]


class _WireGoal(C.Structure):  # bioik_plugin_goal
    _fields_ = [("opcode", C.c_int32), ("secondary", C.c_int32), ("n_numbers", C.c_int32), ("reserved", C.c_int32),
                ("link", C.c_char_p), ("variable", C.c_char_p), ("weight", C.c_double), ("numbers", C.POINTER(C.c_double))]


_shims = {}


def load_shim(path=None):
    """libbio_ik_shim.so (make -C bio_ik_amd/cpp shim; __graft_entry__.build() does).  No fallback: without it the class cannot solve."""
    from_env = path is None and "BIOIK_PLUGIN_SHIM" in os.environ
    path = path or os.environ.get("BIOIK_PLUGIN_SHIM", SHIM_PATH)
    if from_env and "hostsim" in os.path.basename(path):  # (the test-suite's shim over the host simulator is handed over explicitly, `lib=`: never through the environment)
        raise ImportError("BIOIK_PLUGIN_SHIM names the test-suite's host-simulator shim (%s): bio_ik_amd has no CPU compute path" % path)
    if path in _shims:
        return _shims[path]
    if path == SHIM_PATH:
        load_library()  # the solver library the shim is linked against (shares the HIP runtime with torch)
    if not os.path.exists(path):
        raise ImportError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`" % path)
    L = C.CDLL(path)
    vp, u32, u64, i32, dp = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32, C.POINTER(C.c_double)
    strs = C.POINTER(C.c_char_p)
    L.bioik_plugin_last_error.restype = C.c_char_p
    L.bioik_plugin_create.argtypes = [C.POINTER(abi.ModelDesc), strs, strs, strs, u32, C.POINTER(i32), u32, strs, C.POINTER(_Settings), C.POINTER(vp)]
    L.bioik_plugin_destroy.argtypes = [vp]
    L.bioik_plugin_destroy.restype = None
    L.bioik_plugin_update.argtypes = [vp, C.POINTER(_Settings)]
    L.bioik_plugin_group_variable_count.argtypes = [vp]
    L.bioik_plugin_group_variable_count.restype = u32
    L.bioik_plugin_group_variables.argtypes = [vp, C.POINTER(i32)]
    L.bioik_plugin_group_variables.restype = None
    L.bioik_plugin_submit.argtypes = [vp, u64, dp, dp, dp, dp, u32, C.POINTER(_WireGoal), i32, u32, strs, C.c_double, i32, C.POINTER(u64)]
    L.bioik_plugin_wait.argtypes = [vp, u64, dp, C.POINTER(C.c_uint8), dp]
    L.bioik_plugin_search_each.argtypes = [vp, u64, dp, dp, dp, dp, u32, C.POINTER(_WireGoal), i32, u32, strs, C.c_double, i32, dp, C.POINTER(C.c_uint8), dp, dp]
    L.bioik_plugin_postprocess.argtypes = [vp, u64, dp, dp, u32, C.POINTER(i32)]
    _shims[path] = L
    return L


def _strs(names):
    return (C.c_char_p * max(len(names), 1))(*[(n or "").encode() for n in names])


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
    "gpu_population": 128, "gpu_fk": "exact", "gpu_islands": 0, "gpu_max_steps": 64, "gpu_devices": None, "gpu_reproducible_calls": False, "gpu_schedule": "auto",
}


class BioIKKinematicsPlugin:
    def __init__(self, lib=None):
        """`lib`: path of another build of libbio_ik_shim.so (the test-suite links one against the host simulator of the kernels)."""
        self._lib_path = lib
        self._h = None
        self._pushed = None
        self.robot_model = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def close(self):
        if self._h is not None:
            self._L.bioik_plugin_destroy(self._h)
            self._h = None

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError(self._L.bioik_plugin_last_error().decode())

    def _settings(self):
        p = self.params
        devices = np.asarray(p["gpu_devices"] or [0], dtype=np.int32)
        s = _Settings(mode=str(p["mode"]).encode(), gpu_fk=str(p["gpu_fk"]).encode(), gpu_schedule=str(p["gpu_schedule"]).encode(), random_seed=int(p["random_seed"]),
                      no_wipeout=int(bool(p["no_wipeout"])), position_only_ik=int(bool(p["position_only_ik"])),
                      gpu_population=int(p["gpu_population"]), gpu_islands=int(p["gpu_islands"]), gpu_max_steps=int(p["gpu_max_steps"]),
                      gpu_reproducible_calls=int(bool(p["gpu_reproducible_calls"])), n_devices=len(devices), devices=abi.iptr(devices),
                      dpos=float(p["dpos"]), drot=float(p["drot"]), dtwist=float(p["dtwist"]), rotation_scale=float(p["rotation_scale"]),
                      center_joints_weight=float(p["center_joints_weight"]), avoid_joint_limits_weight=float(p["avoid_joint_limits_weight"]),
                      minimal_displacement_weight=float(p["minimal_displacement_weight"]))
        return s, devices

    def _push_params(self):
        """`params` may be edited between calls (everything but the devices takes effect)"""
        snapshot = repr(sorted(self.params.items(), key=lambda kv: kv[0]))
        if snapshot != self._pushed:
            s, keep = self._settings()
            self._chk(self._L.bioik_plugin_update(self._h, C.byref(s)))
            self._pushed = snapshot

    # ---- kinematics::KinematicsBase ---------------------------------------------------------------------------
    def initialize(self, robot_model, group_name, base_frame, tip_frames, search_discretization=0.0, params=None):
        """kinematics_plugin.cpp:337-374 (the RobotModel overload) + load() :191-335; returns True like the reference.
        Configuration errors (unknown solver mode, no device) raise RuntimeError, as the reference's ERROR macro throws."""
        self.close()
        self._L = load_shim(self._lib_path)
        self.robot_model = robot_model
        self.group_name = group_name
        self.base_frame = base_frame
        self.tip_frames = [tip_frames] if isinstance(tip_frames, str) else list(tip_frames)
        self.params = dict(DEFAULT_PARAMS)
        self.params.update(params or {})
        group = robot_model.groups[group_name]
        self.joint_names = [robot_model.joint_names[j] for j in group.active_joints]  # kinematics_plugin.cpp:226-231
        self.link_names = list(self.tip_frames)
        md = robot_model.desc()
        gj = np.asarray(group.active_joints, dtype=np.int32)
        s, keep = self._settings()
        h = C.c_void_p()
        self._chk(self._L.bioik_plugin_create(C.byref(md), _strs(robot_model.link_names), _strs(robot_model.joint_names), _strs(robot_model.variable_names),
                                              len(gj), abi.iptr(gj), len(self.tip_frames), _strs(self.tip_frames), C.byref(s), C.byref(h)))
        self._h = h
        self._pushed = repr(sorted(self.params.items(), key=lambda kv: kv[0]))
        gv = np.zeros(self._L.bioik_plugin_group_variable_count(h), dtype=np.int32)
        self._L.bioik_plugin_group_variables(h, abi.iptr(gv))
        self._group_vars = [int(v) for v in gv]
        self._base_link = robot_model.link_index(base_frame)
        self._base_default = link_transform(robot_model, self._base_link, robot_model.default_positions())
        return True

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

    # ---- the batched entry point and its submit / wait form; the reference's single-query one -------------------------------
    def searchPositionIKBatchAsync(self, ik_poses, ik_seed_states, options=None, context_state=None, timeout=0.0):
        """Marshals and enqueues n independent queries sharing one goal structure, waits for nothing; finish with
        searchPositionIKBatchWait.  ik_poses [n][tips][7] in the base frame (ignored when options.replace), ik_seed_states
        [n][group variables], context_state [variables] (None: the model's default positions, :465-472).  `timeout` [s] bounds the
        call on the device clock (ik_parallel.h:160; every query still runs one step), <= 0: only gpu_max_steps applies."""
        self._push_params()
        options = options or KinematicsQueryOptions()
        m = self.robot_model
        G = len(self._group_vars)
        seeds = np.ascontiguousarray(np.asarray(ik_seed_states, dtype=np.float64).reshape(-1, G))
        n = seeds.shape[0]
        bio = options if isinstance(options, BioIKKinematicsQueryOptions) else None
        replace = bool(bio and bio.replace)
        caller = list(bio.goals) if bio else []
        fixed = list(bio.fixed_joints) if bio else []
        wire = (_WireGoal * max(len(caller), 1))()
        keep = []
        for w, g in zip(wire, caller):
            if g.opcode is None:
                raise NotImplementedError("%s has no device implementation (host-callback goal)" % type(g).__name__)
            numbers = np.zeros(abi.GOAL_PARAM_COUNT[g.opcode])
            p = np.asarray(g.params(), dtype=np.float64)
            numbers[:len(p)] = p
            keep.append(numbers)
            w.opcode, w.secondary, w.n_numbers, w.weight = g.opcode, int(g.isSecondary()), len(numbers), g.getWeight()
            w.link = (g.link_name() or "").encode()
            w.variable = (g.variable_name() or "").encode()
            w.numbers = abi.dptr(numbers)
        if context_state is None:
            context, base = m.default_positions(), self._base_default
        else:
            context = np.ascontiguousarray(np.asarray(context_state, dtype=np.float64).reshape(m.n_variables))
            base = link_transform(m, self._base_link, context)  # :487-502
        base = np.ascontiguousarray(base, dtype=np.float64)
        poses = np.zeros(0) if replace else np.ascontiguousarray(np.asarray(ik_poses, dtype=np.float64).reshape(n, len(self.tip_frames), 7))
        ticket = C.c_uint64()
        self._chk(self._L.bioik_plugin_submit(self._h, n, abi.dptr(seeds), abi.dptr(poses) if poses.size else None, abi.dptr(base), abi.dptr(context),
                                              len(caller), wire, int(replace), len(fixed), _strs(fixed), float(timeout) if timeout and timeout > 0.0 else 0.0,
                                              int(bool(getattr(options, "return_approximate_solution", False))), C.byref(ticket)))
        return (ticket.value, n, bio)

    def searchPositionIKBatchWait(self, pending):
        """-> (solutions [n][group variables], ok [n] bool, fitness [n], error codes [n])"""
        ticket, n, bio = pending
        solutions = np.zeros((n, len(self._group_vars)))
        ok = np.zeros(n, dtype=np.uint8)
        fit = np.zeros(n)
        self._chk(self._L.bioik_plugin_wait(self._h, ticket, abi.dptr(solutions), abi.u8ptr(ok), abi.dptr(fit)))
        ok = ok != 0
        codes = np.where(ok, MoveItErrorCodes.SUCCESS, MoveItErrorCodes.NO_IK_SOLUTION)
        if bio is not None and n:
            bio.solution_fitness = float(fit[-1])  # :632-634
        return solutions, ok, fit, codes

    def searchPositionIKEach(self, ik_poses, ik_seed_states, options=None, context_state=None, timeout=0.0):
        """n poses the way MoveIt asks for them: ONE searchPositionIK call per pose, each with `timeout`, the next when the last has returned -- looped inside
        the plugin library (bioik_plugin_search_each), so that the wall time of every call is the plugin core's, not this face's marshalling.
        -> (solutions [n][group variables], ok [n] bool, fitness [n], seconds [n])"""
        self._push_params()
        options = options or KinematicsQueryOptions()
        m = self.robot_model
        G = len(self._group_vars)
        seeds = np.ascontiguousarray(np.asarray(ik_seed_states, dtype=np.float64).reshape(-1, G))
        n = seeds.shape[0]
        bio = options if isinstance(options, BioIKKinematicsQueryOptions) else None
        replace = bool(bio and bio.replace)
        caller = list(bio.goals) if bio else []
        fixed = list(bio.fixed_joints) if bio else []
        wire = (_WireGoal * max(len(caller), 1))()
        keep = []
        for w, g in zip(wire, caller):
            if g.opcode is None:
                raise NotImplementedError("%s has no device implementation (host-callback goal)" % type(g).__name__)
            numbers = np.zeros(abi.GOAL_PARAM_COUNT[g.opcode])
            pnum = np.asarray(g.params(), dtype=np.float64)
            numbers[:len(pnum)] = pnum
            keep.append(numbers)
            w.opcode, w.secondary, w.n_numbers, w.weight = g.opcode, int(g.isSecondary()), len(numbers), g.getWeight()
            w.link = (g.link_name() or "").encode()
            w.variable = (g.variable_name() or "").encode()
            w.numbers = abi.dptr(numbers)
        if context_state is None:
            context, base = m.default_positions(), self._base_default
        else:
            context = np.ascontiguousarray(np.asarray(context_state, dtype=np.float64).reshape(m.n_variables))
            base = link_transform(m, self._base_link, context)
        base = np.ascontiguousarray(base, dtype=np.float64)
        poses = np.zeros(0) if replace else np.ascontiguousarray(np.asarray(ik_poses, dtype=np.float64).reshape(n, len(self.tip_frames), 7))
        solutions, ok, fit, seconds = np.zeros((n, G)), np.zeros(n, dtype=np.uint8), np.zeros(n), np.zeros(n)
        self._chk(self._L.bioik_plugin_search_each(self._h, n, abi.dptr(seeds), abi.dptr(poses) if poses.size else None, abi.dptr(base), abi.dptr(context), len(caller), wire,
                                                   int(replace), len(fixed), _strs(fixed), float(timeout) if timeout and timeout > 0.0 else 0.0,
                                                   int(bool(getattr(options, "return_approximate_solution", False))), abi.dptr(solutions), abi.u8ptr(ok), abi.dptr(fit),
                                                   abi.dptr(seconds)))
        return solutions, ok != 0, fit, seconds

    def searchPositionIKBatch(self, ik_poses, ik_seed_states, options=None, context_state=None, timeout=0.0):
        return self.searchPositionIKBatchWait(self.searchPositionIKBatchAsync(ik_poses, ik_seed_states, options, context_state, timeout))

    def searchPositionIK(self, ik_poses, ik_seed_state, timeout, solution, error_code, options=None, consistency_limits=None,
                         solution_callback=None, context_state=None):
        """kinematics_plugin.cpp:437-655.  `timeout` [s] is the reference's wall-clock budget (:504, :574; honoured on the device,
        at least one step); `gpu_max_steps` bounds the call as well."""
        poses = np.asarray(ik_poses, dtype=np.float64).reshape(1, -1, 7) if len(ik_poses) else np.zeros((1, 0, 7))
        sols, ok, fit, codes = self.searchPositionIKBatch(poses, [ik_seed_state], options, context_state, timeout=timeout)
        solution[:] = list(sols[0])
        if not ok[0]:
            error_code.val = MoveItErrorCodes.NO_IK_SOLUTION
            return False
        if solution_callback is not None:  # :644-649
            solution_callback(ik_poses[0] if len(ik_poses) else None, solution, error_code)
            return error_code.val == MoveItErrorCodes.SUCCESS
        error_code.val = MoveItErrorCodes.SUCCESS
        return True

    def _wrap_angles(self, sol, seed, active):
        """kinematics_plugin.cpp:580-616 on full variable vectors [n][variables] (core::Engine::postprocess, exposed for the tests)"""
        out = np.ascontiguousarray(np.array(sol, dtype=np.float64))
        seed = np.ascontiguousarray(np.asarray(seed, dtype=np.float64))
        act = np.asarray(active, dtype=np.int32)
        self._chk(self._L.bioik_plugin_postprocess(self._h, out.shape[0], abi.dptr(seed), abi.dptr(out), len(act), abi.iptr(act)))
        return out


def visible_devices():
    return list(range(device_count()))
