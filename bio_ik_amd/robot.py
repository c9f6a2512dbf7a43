"""Flat robot-model description = what the reference reads from moveit::core::RobotModel.

`RobotModel` is a small URDF-like builder (links added parent-first, one parent joint per link) that
flattens into the `bioik_model_desc` POD of include/bioik_hip.h.  It plays the role of
moveit::core::RobotModel for the reference call sites src/forward_kinematics.h:192-213, 230-246,
268-329 and include/bio_ik/robot_info.h:70-106.  `JointModelGroup` plays the role of
moveit::core::JointModelGroup (getActiveJointModels / getVariableNames / end-effector tips) used by
src/problem.cpp:117-124, 201-204 and src/kinematics_plugin.cpp:218-237.

Fixtures (SURVEY.md Appendix C — PR2 numbers restated from pr2_description, no URDF is on disk, hence
"PR2-like"): `pr2_like()` (torso + both arms; groups right_arm, left_arm, all) and `snake(n)`.
"""
import ctypes as C
import math

import numpy as np

from . import abi


def quat_from_rpy(r, p, y):
    """URDF rpy (fixed-axis XYZ = yaw*pitch*roll) -> quaternion xyzw."""
    cr, sr = math.cos(r / 2), math.sin(r / 2)
    cp, sp = math.cos(p / 2), math.sin(p / 2)
    cy, sy = math.cos(y / 2), math.sin(y / 2)
    return (sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy,
            cr * cp * cy + sr * sp * sy)


class JointModelGroup:
    def __init__(self, name, active_joints, tips):
        self.name = name
        self.active_joints = list(active_joints)  # link indices, getActiveJointModels() order
        self.tips = list(tips)                    # link indices of the end-effector tips


class RobotModel:
    def __init__(self, name="robot"):
        self.name = name
        self.link_names = []
        self.joint_names = []
        self.link_parent = []
        self.link_origin = []
        self.link_mass = []     # urdf <inertial> mass per link (0 = none)
        self.link_center = []   # urdf <inertial> origin xyz per link
        self.joint_type = []
        self.joint_axis = []
        self.joint_first_variable = []
        self.joint_mimic = []
        self.joint_mimic_factor = []
        self.joint_mimic_offset = []
        self.variable_names = []
        self.var_min = []
        self.var_max = []
        self.var_bounded = []
        self.var_max_velocity = []
        self.groups = {}
        self._keep = None

    # ---- construction ----
    def add_link(self, link_name, parent=None, joint_name=None, joint_type="fixed", xyz=(0, 0, 0), rpy=(0, 0, 0),
                 axis=(0, 0, 1), lower=0.0, upper=0.0, velocity=0.0, mimic=None, quat=None, mass=0.0, com=(0, 0, 0)):
        """Add `link_name` attached to `parent` through joint `joint_name` (URDF semantics).
        joint_type: fixed | revolute | continuous | prismatic | floating | planar.  mimic = (joint, factor, offset).
        mass / com: the link's URDF <inertial> mass and origin position (BalanceGoal, goal_types.cpp:236-247)."""
        if link_name in self.link_names:
            raise ValueError("duplicate link %r" % link_name)
        if parent is None:
            pidx = -1
            if self.link_names:
                raise ValueError("only the first link may be the root")
        else:
            pidx = self.link_names.index(parent)
        jt = {"fixed": abi.JOINT_FIXED, "revolute": abi.JOINT_REVOLUTE, "continuous": abi.JOINT_REVOLUTE,
              "prismatic": abi.JOINT_PRISMATIC, "floating": abi.JOINT_FLOATING, "planar": abi.JOINT_PLANAR}[joint_type]
        q = quat if quat is not None else quat_from_rpy(*rpy)
        idx = len(self.link_names)
        self.link_names.append(link_name)
        self.joint_names.append(joint_name or (link_name + "_joint"))
        self.link_parent.append(pidx)
        self.link_origin.append([float(xyz[0]), float(xyz[1]), float(xyz[2]), q[0], q[1], q[2], q[3]])
        self.link_mass.append(float(mass))
        self.link_center.append([float(com[0]), float(com[1]), float(com[2])])
        self.joint_type.append(jt)
        a = np.asarray(axis, dtype=np.float64)
        if jt in (abi.JOINT_REVOLUTE, abi.JOINT_PRISMATIC):
            a = a / np.linalg.norm(a)  # urdf axis is normalised by the URDF parser
        self.joint_axis.append([float(a[0]), float(a[1]), float(a[2])])
        nv = abi.JOINT_VAR_COUNT[jt]
        self.joint_first_variable.append(len(self.variable_names) if nv else -1)
        if mimic is not None:
            self.joint_mimic.append(self.joint_names.index(mimic[0]))
            self.joint_mimic_factor.append(float(mimic[1]))
            self.joint_mimic_offset.append(float(mimic[2]))
        else:
            self.joint_mimic.append(-1)
            self.joint_mimic_factor.append(1.0)
            self.joint_mimic_offset.append(0.0)
        jn = self.joint_names[-1]
        if jt == abi.JOINT_REVOLUTE or jt == abi.JOINT_PRISMATIC:
            self.variable_names.append(jn)
            if joint_type == "continuous":
                # MoveIt RevoluteJointModel continuous: bounds [-pi, pi], position_bounded_ = false
                self.var_min.append(-math.pi)
                self.var_max.append(math.pi)
                self.var_bounded.append(0)
            else:
                self.var_min.append(float(lower))
                self.var_max.append(float(upper))
                self.var_bounded.append(1)
            self.var_max_velocity.append(float(velocity))
        elif jt == abi.JOINT_FLOATING:
            for s, lo, hi, b in (("trans_x", -1e300, 1e300, 0), ("trans_y", -1e300, 1e300, 0), ("trans_z", -1e300, 1e300, 0),
                                 ("rot_x", -1.0, 1.0, 1), ("rot_y", -1.0, 1.0, 1), ("rot_z", -1.0, 1.0, 1), ("rot_w", -1.0, 1.0, 1)):
                self.variable_names.append(jn + "/" + s)
                self.var_min.append(lo)
                self.var_max.append(hi)
                self.var_bounded.append(b)
                self.var_max_velocity.append(float(velocity))
        elif jt == abi.JOINT_PLANAR:
            for s, lo, hi, b in (("x", -1e300, 1e300, 0), ("y", -1e300, 1e300, 0), ("theta", -math.pi, math.pi, 0)):
                self.variable_names.append(jn + "/" + s)
                self.var_min.append(lo)
                self.var_max.append(hi)
                self.var_bounded.append(b)
                self.var_max_velocity.append(float(velocity))
        self._keep = None
        return idx

    def add_group(self, name, joints=None, chain=None, tips=None):
        """Group from an explicit joint-name list or a (base_link, tip_link) chain, like an SRDF <group>."""
        if chain is not None:
            base, tip = chain
            links = []
            l = self.link_names.index(tip)
            b = self.link_names.index(base)
            while l != b:
                if l < 0:
                    raise ValueError("%r is not an ancestor of %r" % (base, tip))
                links.append(l)
                l = self.link_parent[l]
            links.reverse()
            tips = tips or [tip]
        else:
            links = [self.joint_names.index(j) for j in joints]
        active = [l for l in links if self.joint_type[l] != abi.JOINT_FIXED and self.joint_mimic[l] < 0]
        tip_idx = [self.link_names.index(t) for t in (tips or [])]
        g = JointModelGroup(name, active, tip_idx)
        self.groups[name] = g
        return g

    # ---- queries ----
    @property
    def n_links(self):
        return len(self.link_names)

    @property
    def n_variables(self):
        return len(self.variable_names)

    def link_index(self, name):
        try:
            return self.link_names.index(name)
        except ValueError:
            raise KeyError("link not found: %s" % name)  # reference problem.cpp:141

    def joint_index(self, name):
        try:
            return self.joint_names.index(name)
        except ValueError:
            raise KeyError("joint not found: %s" % name)

    def variable_index(self, name):
        try:
            return self.variable_names.index(name)
        except ValueError:
            raise KeyError("joint variable not found: %s" % name)  # reference problem.cpp:125

    def default_positions(self):
        """RobotModel::getVariableDefaultPositions: 0 if within bounds else the midpoint (floating: identity quat)."""
        out = np.zeros(self.n_variables)
        for v in range(self.n_variables):
            lo, hi = self.var_min[v], self.var_max[v]
            if not (lo <= 0.0 <= hi):
                out[v] = 0.5 * (lo + hi)
            if self.variable_names[v].endswith("/rot_w"):
                out[v] = 1.0
        return out

    def resolve_mimic_chains(self):
        """A joint that mimics a joint that itself mimics another follows the joint at the END of the chain with the composed factor and offset, as MoveIt's
        RobotModel::buildMimic leaves every model before bio_ik sees it: x = f1 (f2 y + o2) + o1 -> factor f1 f2, offset o1 + f1 o2."""
        for _ in range(len(self.joint_mimic) + 1):
            changed = False
            for i, m in enumerate(self.joint_mimic):
                if m >= 0 and self.joint_mimic[m] >= 0:
                    self.joint_mimic_offset[i] = self.joint_mimic_offset[i] + self.joint_mimic_factor[i] * self.joint_mimic_offset[m]
                    self.joint_mimic_factor[i] = self.joint_mimic_factor[i] * self.joint_mimic_factor[m]
                    self.joint_mimic[i] = self.joint_mimic[m]
                    changed = True
            if not changed:
                return
        raise ValueError("mimic joints that follow each other in a circle")

    # ---- flattening ----
    def arrays(self):
        if self._keep is None:
            self.resolve_mimic_chains()
            k = {}
            k["link_parent"] = np.asarray(self.link_parent, dtype=np.int32)
            k["link_origin"] = np.asarray(self.link_origin, dtype=np.float64).reshape(-1, 7)
            k["joint_type"] = np.asarray(self.joint_type, dtype=np.int32)
            k["joint_axis"] = np.asarray(self.joint_axis, dtype=np.float64).reshape(-1, 3)
            k["joint_first_variable"] = np.asarray(self.joint_first_variable, dtype=np.int32)
            k["joint_mimic"] = np.asarray(self.joint_mimic, dtype=np.int32)
            k["joint_mimic_factor"] = np.asarray(self.joint_mimic_factor, dtype=np.float64)
            k["joint_mimic_offset"] = np.asarray(self.joint_mimic_offset, dtype=np.float64)
            k["var_min"] = np.asarray(self.var_min, dtype=np.float64)
            k["var_max"] = np.asarray(self.var_max, dtype=np.float64)
            k["var_bounded"] = np.asarray(self.var_bounded, dtype=np.uint8)
            k["var_max_velocity"] = np.asarray(self.var_max_velocity, dtype=np.float64)
            k["link_mass"] = np.asarray(self.link_mass, dtype=np.float64)
            k["link_center"] = np.asarray(self.link_center, dtype=np.float64).reshape(-1, 3)
            self._keep = k
        return self._keep

    def desc(self):
        """bioik_model_desc pointing into arrays kept alive by this object."""
        k = self.arrays()
        d = abi.ModelDesc()
        d.struct_size = C.sizeof(abi.ModelDesc)
        d.n_links = self.n_links
        d.n_variables = self.n_variables
        d.link_parent = abi.iptr(k["link_parent"])
        d.link_origin = abi.dptr(k["link_origin"])
        d.joint_type = abi.iptr(k["joint_type"])
        d.joint_axis = abi.dptr(k["joint_axis"])
        d.joint_first_variable = abi.iptr(k["joint_first_variable"])
        d.joint_mimic = abi.iptr(k["joint_mimic"])
        d.joint_mimic_factor = abi.dptr(k["joint_mimic_factor"])
        d.joint_mimic_offset = abi.dptr(k["joint_mimic_offset"])
        d.var_min = abi.dptr(k["var_min"])
        d.var_max = abi.dptr(k["var_max"])
        d.var_bounded = abi.u8ptr(k["var_bounded"])
        d.var_max_velocity = abi.dptr(k["var_max_velocity"])
        if np.any(k["link_mass"] > 0.0):
            d.link_mass = abi.dptr(k["link_mass"])
            d.link_center = abi.dptr(k["link_center"])
        return d


# ---------------------------------------------------------------------------------------------
# fixtures
# ---------------------------------------------------------------------------------------------
def _pr2_arm(m, side, y_sign):
    s = side
    pan_lo, pan_hi = (-2.2854, 0.7146) if side == "r" else (-0.7146, 2.2854)
    roll_lo, roll_hi = (-3.9, 0.8) if side == "r" else (-0.8, 3.9)
    m.add_link(s + "_shoulder_pan_link", "torso_lift_link", s + "_shoulder_pan_joint", "revolute", xyz=(0.0, y_sign * 0.188, 0.0),
               axis=(0, 0, 1), lower=pan_lo, upper=pan_hi, velocity=2.088)
    m.add_link(s + "_shoulder_lift_link", s + "_shoulder_pan_link", s + "_shoulder_lift_joint", "revolute", xyz=(0.1, 0, 0),
               axis=(0, 1, 0), lower=-0.5236, upper=1.3963, velocity=2.082)
    m.add_link(s + "_upper_arm_roll_link", s + "_shoulder_lift_link", s + "_upper_arm_roll_joint", "revolute", xyz=(0, 0, 0),
               axis=(1, 0, 0), lower=roll_lo, upper=roll_hi, velocity=3.27)
    m.add_link(s + "_upper_arm_link", s + "_upper_arm_roll_link", s + "_upper_arm_joint", "fixed")
    m.add_link(s + "_elbow_flex_link", s + "_upper_arm_link", s + "_elbow_flex_joint", "revolute", xyz=(0.4, 0, 0),
               axis=(0, 1, 0), lower=-2.3213, upper=0.0, velocity=3.3)
    m.add_link(s + "_forearm_roll_link", s + "_elbow_flex_link", s + "_forearm_roll_joint", "continuous", xyz=(0, 0, 0),
               axis=(1, 0, 0), velocity=3.6)
    m.add_link(s + "_forearm_link", s + "_forearm_roll_link", s + "_forearm_joint", "fixed")
    m.add_link(s + "_wrist_flex_link", s + "_forearm_link", s + "_wrist_flex_joint", "revolute", xyz=(0.321, 0, 0),
               axis=(0, 1, 0), lower=-2.18, upper=0.0, velocity=3.078)
    m.add_link(s + "_wrist_roll_link", s + "_wrist_flex_link", s + "_wrist_roll_joint", "continuous", xyz=(0, 0, 0),
               axis=(1, 0, 0), velocity=3.6)


def pr2_like():
    """PR2-like torso + both arms.  Groups: right_arm, left_arm (7 DOF, torso inactive), all (15 DOF)."""
    m = RobotModel("pr2_like")
    m.add_link("base_footprint")
    m.add_link("base_link", "base_footprint", "base_footprint_joint", "fixed", xyz=(0, 0, 0.051))
    m.add_link("torso_lift_link", "base_link", "torso_lift_joint", "prismatic", xyz=(-0.05, 0, 0.739675), axis=(0, 0, 1),
               lower=0.0, upper=0.33, velocity=0.013)
    _pr2_arm(m, "r", -1.0)
    _pr2_arm(m, "l", +1.0)
    m.add_group("right_arm", chain=("torso_lift_link", "r_wrist_roll_link"))
    m.add_group("left_arm", chain=("torso_lift_link", "l_wrist_roll_link"))
    arm = lambda s: [s + j for j in ("_shoulder_pan_joint", "_shoulder_lift_joint", "_upper_arm_roll_joint", "_elbow_flex_joint",
                                     "_forearm_roll_joint", "_wrist_flex_joint", "_wrist_roll_joint")]
    m.add_group("all", joints=["torso_lift_joint"] + arm("r") + arm("l"), tips=["r_wrist_roll_link", "l_wrist_roll_link"])
    return m


def snake(n=31, link_length=0.1, limit=1.5, velocity=1.0):
    """n-DOF snake chain: revolute joints with alternating y/z axes, 0.1 m links, +-1.5 rad, vmax 1."""
    m = RobotModel("snake%d" % n)
    m.add_link("base")
    prev = "base"
    for i in range(n):
        name = "seg%d" % i
        m.add_link(name, prev, "j%d" % i, "revolute", xyz=(link_length if i else 0.0, 0, 0), axis=(0, 1, 0) if i % 2 == 0 else (0, 0, 1),
                   lower=-limit, upper=limit, velocity=velocity)
        prev = name
    m.add_link("tip", prev, "tip_joint", "fixed", xyz=(link_length, 0, 0))
    m.add_group("snake", chain=("base", "tip"))
    return m


# ---------------------------------------------------------------------------------------------
# host-side frame helpers for the plugin boundary (goal poses into the model frame,
# reference src/kinematics_plugin.cpp:487-502: RobotState::getGlobalLinkTransform(getBaseFrame()))
# ---------------------------------------------------------------------------------------------
def quat_rotate(q, v):
    x, y, z, w = q
    t = 2.0 * np.cross([x, y, z], v)
    return np.asarray(v, dtype=np.float64) + w * t + np.cross([x, y, z], t)


def quat_multiply(p, q):
    px, py, pz, pw = p
    qx, qy, qz, qw = q
    return np.array([pw * qx + px * qw + py * qz - pz * qy, pw * qy - px * qz + py * qw + pz * qx,
                     pw * qz + px * qy - py * qx + pz * qw, pw * qw - px * qx - py * qy - pz * qz])


def frame_concat(a, b):
    """a o b for frames (px py pz qx qy qz qw)"""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.concatenate([a[:3] + quat_rotate(a[3:], b[:3]), quat_multiply(a[3:], b[3:])])


def link_transform(model, link, positions):
    """Global frame of `link` at the given variable positions (fixed / revolute / prismatic joints): setup-time helper of
    the plugin mirror, not part of the solver."""
    chain = []
    l = link
    while l >= 0:
        chain.append(l)
        l = model.link_parent[l]
    f = np.array([0, 0, 0, 0, 0, 0, 1.0])
    for l in reversed(chain):
        f = frame_concat(f, model.link_origin[l])
        jt = model.joint_type[l]
        if jt == abi.JOINT_REVOLUTE:
            h = 0.5 * positions[model.joint_first_variable[l]]
            a = np.asarray(model.joint_axis[l])
            f = frame_concat(f, np.concatenate([[0, 0, 0], a * math.sin(h), [math.cos(h)]]))
        elif jt == abi.JOINT_PRISMATIC:
            a = np.asarray(model.joint_axis[l]) * positions[model.joint_first_variable[l]]
            f = frame_concat(f, np.concatenate([a, [0, 0, 0, 1.0]]))
        elif jt != abi.JOINT_FIXED:
            raise NotImplementedError("floating / planar joints above the base frame")
    return f
