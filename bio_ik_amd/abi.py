"""ctypes mirror of include/bioik_hip.h (the C-ABI PODs and constants).

Only data definitions live here: they are shared by the product wrapper (bio_ik_amd.solver) and, as
input data, by the test-only oracle wrapper (oracle/orc.py)."""
import ctypes as C

import numpy as np

# ---- status codes ----
OK = 0
ERR_INVALID_ARGUMENT = -1
ERR_UNSUPPORTED = -2
ERR_NO_DEVICE = -3
ERR_HIP = -4
ERR_NOT_FOUND = -5

# ---- joint types (moveit::core::JointModel::JointType, reference src/forward_kinematics.h:78-139) ----
JOINT_FIXED, JOINT_REVOLUTE, JOINT_PRISMATIC, JOINT_FLOATING, JOINT_PLANAR = 0, 1, 2, 3, 4
JOINT_VAR_COUNT = {JOINT_FIXED: 0, JOINT_REVOLUTE: 1, JOINT_PRISMATIC: 1, JOINT_FLOATING: 7, JOINT_PLANAR: 3}

# ---- goal opcodes (reference include/bio_ik/goal_types.h) ----
(GOAL_POSITION, GOAL_ORIENTATION, GOAL_POSE, GOAL_LOOK_AT, GOAL_MAX_DISTANCE, GOAL_MIN_DISTANCE, GOAL_LINE, GOAL_PLANE,
 GOAL_AVOID_JOINT_LIMITS, GOAL_CENTER_JOINTS, GOAL_REGULARIZATION, GOAL_MINIMAL_DISPLACEMENT, GOAL_JOINT_VARIABLE,
 GOAL_SIDE, GOAL_DIRECTION, GOAL_CONE, GOAL_BALANCE) = range(17)
GOAL_PARAM_COUNT = {GOAL_POSITION: 3, GOAL_ORIENTATION: 4, GOAL_POSE: 8, GOAL_LOOK_AT: 6, GOAL_MAX_DISTANCE: 4,
                    GOAL_MIN_DISTANCE: 4, GOAL_LINE: 6, GOAL_PLANE: 6, GOAL_AVOID_JOINT_LIMITS: 0, GOAL_CENTER_JOINTS: 0,
                    GOAL_REGULARIZATION: 0, GOAL_MINIMAL_DISPLACEMENT: 0, GOAL_JOINT_VARIABLE: 1, GOAL_SIDE: 6,
                    GOAL_DIRECTION: 6, GOAL_CONE: 11, GOAL_BALANCE: 6}

# ---- solver modes (IKFactory names, reference src/ik_evolution_2.cpp:652-654) ----
MODE_BIO2, MODE_BIO2_MEMETIC, MODE_BIO2_MEMETIC_L, MODE_GD_C, MODE_JAC, MODE_GD, MODE_GD_R = 0, 1, 2, 3, 4, 5, 6
SCHEDULE_LATENCY, SCHEDULE_THROUGHPUT, SCHEDULE_AUTO = 0, 1, 2
SCHEDULE_BY_NAME = {"latency": SCHEDULE_LATENCY, "throughput": SCHEDULE_THROUGHPUT, "auto": SCHEDULE_AUTO}
MODE_BY_NAME = {"bio2": MODE_BIO2, "bio2_memetic": MODE_BIO2_MEMETIC, "bio2_memetic_l": MODE_BIO2_MEMETIC_L, "gd_c": MODE_GD_C, "jac": MODE_JAC, "gd": MODE_GD, "gd_r": MODE_GD_R}
FK_LINEAR, FK_EXACT = 0, 1

_pd = C.POINTER(C.c_double)
_pi = C.POINTER(C.c_int32)
_pu8 = C.POINTER(C.c_uint8)


class ModelDesc(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_links", C.c_uint32), ("n_variables", C.c_uint32), ("reserved", C.c_uint32),
                ("link_parent", _pi), ("link_origin", _pd), ("joint_type", _pi), ("joint_axis", _pd),
                ("joint_first_variable", _pi), ("joint_mimic", _pi), ("joint_mimic_factor", _pd),
                ("joint_mimic_offset", _pd), ("var_min", _pd), ("var_max", _pd), ("var_bounded", _pu8),
                ("var_max_velocity", _pd), ("link_mass", _pd), ("link_center", _pd)]


class GoalDesc(C.Structure):
    _fields_ = [("type", C.c_int32), ("link", C.c_int32), ("variable", C.c_int32), ("secondary", C.c_int32),
                ("weight", C.c_double)]


class ProblemDesc(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_group_joints", C.c_uint32), ("group_joints", _pi),
                ("n_goals", C.c_uint32), ("n_fixed_joints", C.c_uint32), ("goals", C.POINTER(GoalDesc)),
                ("fixed_joints", _pi)]


class SolveParams(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("mode", C.c_int32), ("fk_mode", C.c_int32), ("population", C.c_int32),
                ("islands", C.c_int32), ("max_steps", C.c_int32), ("random_seed", C.c_uint64), ("dpos", C.c_double),
                ("drot", C.c_double), ("dtwist", C.c_double), ("no_wipeout", C.c_int32), ("schedule", C.c_int32), ("timeout", C.c_double),
                ("island_sync", C.c_int32), ("reserved0", C.c_int32)]


ISLANDS_AUTO = 0  # bioik_solve_params::islands: as many islands as the idle part of the chip carries (include/bioik_hip.h)


def default_solve_params(**kw):
    """bioik_default_solve_params + keyword overrides (yaml-key spelling: mode may be a string)."""
    p = SolveParams()
    p.struct_size = C.sizeof(SolveParams)
    p.mode = MODE_BIO2_MEMETIC
    p.fk_mode = FK_EXACT
    p.population = 128
    p.islands = 1
    p.max_steps = 64
    p.random_seed = 0
    p.dpos = -1.0
    p.drot = -1.0
    p.dtwist = 1e-5
    p.no_wipeout = 0
    p.schedule = SCHEDULE_LATENCY
    p.timeout = 0.0
    p.island_sync = 0
    p.reserved0 = 0
    for k, v in kw.items():
        if k == "mode" and isinstance(v, str):
            v = MODE_BY_NAME[v]
        if k == "schedule" and isinstance(v, str):
            v = SCHEDULE_BY_NAME[v]
        if not hasattr(p, k):
            raise TypeError("unknown solve parameter %r" % k)
        setattr(p, k, v)
    return p


def dptr(a):
    return a.ctypes.data_as(_pd)


def iptr(a):
    return a.ctypes.data_as(_pi)


def u8ptr(a):
    return a.ctypes.data_as(_pu8)


def as_f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a
