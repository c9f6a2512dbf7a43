"""Goal (cost) classes: host-side mirror of reference include/bio_ik/goal.h:97-129 and goal_types.h:56-712.

Same class names, constructor arguments, setters and normalisation behaviour as the reference; instead of a
virtual `evaluate` every built-in goal serialises itself into (opcode, link/variable, weight, secondary) for the
problem template and a flat parameter vector per query (`params()`), which is what the device evaluates.
Goals that need host callbacks or FCL (JointFunctionGoal, LinkFunctionGoal, TouchGoal) have no
device opcode; constructing a problem with them raises NotImplementedError.
"""
import math

import numpy as np

from . import abi


def _vec3(v):
    a = np.asarray(v, dtype=np.float64).reshape(3)
    return a


def _normalized(v):
    a = np.asarray(v, dtype=np.float64)
    return a / math.sqrt(float(np.dot(a, a)))


class Goal:
    """reference goal.h:97-119"""
    opcode = None

    def __init__(self):
        self.weight_ = 1.0
        self.secondary_ = False

    def isSecondary(self):
        return self.secondary_

    def getWeight(self):
        return self.weight_

    def setWeight(self, w):
        self.weight_ = float(w)

    # --- serialisation hooks (the "describe-to-POD" of DESIGN.md §2) ---
    def link_name(self):
        return None

    def variable_name(self):
        return None

    def params(self):
        return np.zeros(0)


class LinkGoalBase(Goal):
    """goal_types.h:56-78"""

    def __init__(self, link_name="", weight=1.0):
        super().__init__()
        self.weight_ = float(weight)
        self.link_name_ = link_name

    def setLinkName(self, n):
        self.link_name_ = n

    def getLinkName(self):
        return self.link_name_

    def link_name(self):
        return self.link_name_


class PositionGoal(LinkGoalBase):
    opcode = abi.GOAL_POSITION

    def __init__(self, link_name="", position=(0, 0, 0), weight=1.0):
        super().__init__(link_name, weight)
        self.position_ = _vec3(position)

    def getPosition(self):
        return self.position_

    def setPosition(self, p):
        self.position_ = _vec3(p)

    def params(self):
        return self.position_.copy()


class OrientationGoal(LinkGoalBase):
    opcode = abi.GOAL_ORIENTATION

    def __init__(self, link_name="", orientation=(0, 0, 0, 1), weight=1.0):
        super().__init__(link_name, weight)
        self.orientation_ = _normalized(orientation)  # goal_types.h:110

    def getOrientation(self):
        return self.orientation_

    def setOrientation(self, q):
        self.orientation_ = _normalized(q)

    def params(self):
        return self.orientation_.copy()


class PoseGoal(LinkGoalBase):
    opcode = abi.GOAL_POSE

    def __init__(self, link_name="", position=(0, 0, 0), orientation=(0, 0, 0, 1), weight=1.0):
        super().__init__(link_name, weight)
        self.position_ = _vec3(position)
        self.orientation_ = _normalized(orientation)  # goal_types.h:139
        self.rotation_scale_ = 0.5                    # goal_types.h:133,140

    def getPosition(self):
        return self.position_

    def setPosition(self, p):
        self.position_ = _vec3(p)

    def getOrientation(self):
        return self.orientation_

    def setOrientation(self, q):
        self.orientation_ = _normalized(q)

    def getRotationScale(self):
        return self.rotation_scale_

    def setRotationScale(self, s):
        self.rotation_scale_ = float(s)

    def params(self):
        return np.concatenate([self.position_, self.orientation_, [self.rotation_scale_]])


class LookAtGoal(LinkGoalBase):
    opcode = abi.GOAL_LOOK_AT

    def __init__(self, link_name="", axis=(1, 0, 0), target=(0, 0, 0), weight=1.0):
        super().__init__(link_name, weight)
        self.axis_ = _vec3(axis)  # constructor does not normalise (goal_types.h:194-199); the setter does (:202)
        self.target_ = _vec3(target)

    def setAxis(self, a):
        self.axis_ = _normalized(a)

    def setTarget(self, t):
        self.target_ = _vec3(t)

    def getAxis(self):
        return self.axis_

    def getTarget(self):
        return self.target_

    def params(self):
        return np.concatenate([self.axis_, self.target_])


class _DistanceGoal(LinkGoalBase):
    def __init__(self, link_name="", target=(0, 0, 0), distance=1.0, weight=1.0):
        super().__init__(link_name, weight)
        self.target = _vec3(target)
        self.distance = float(distance)

    def getTarget(self):
        return self.target

    def setTarget(self, t):
        self.target = _vec3(t)

    def getDistance(self):
        return self.distance

    def setDistance(self, d):
        self.distance = float(d)

    def params(self):
        return np.concatenate([self.target, [self.distance]])


class MaxDistanceGoal(_DistanceGoal):
    opcode = abi.GOAL_MAX_DISTANCE


class MinDistanceGoal(_DistanceGoal):
    opcode = abi.GOAL_MIN_DISTANCE


class LineGoal(LinkGoalBase):
    opcode = abi.GOAL_LINE

    def __init__(self, link_name="", position=(0, 0, 0), direction=(1, 0, 0), weight=1.0):
        super().__init__(link_name, weight)
        self.position = _vec3(position)
        self.direction = _normalized(direction)  # goal_types.h:286

    def setPosition(self, p):
        self.position = _vec3(p)

    def setDirection(self, d):
        self.direction = _normalized(d)

    def getPosition(self):
        return self.position

    def getDirection(self):
        return self.direction

    def params(self):
        return np.concatenate([self.position, self.direction])


class PlaneGoal(LinkGoalBase):
    opcode = abi.GOAL_PLANE

    def __init__(self, link_name="", position=(0, 0, 0), normal=(0, 0, 1), weight=1.0):
        super().__init__(link_name, weight)
        self.position = _vec3(position)
        self.normal = _normalized(normal)  # goal_types.h:314

    def setPosition(self, p):
        self.position = _vec3(p)

    def setNormal(self, n):
        self.normal = _normalized(n)

    def getPosition(self):
        return self.position

    def getNormal(self):
        return self.normal

    def params(self):
        return np.concatenate([self.position, self.normal])


class _JointSetGoal(Goal):
    def __init__(self, weight=1.0, secondary=True):
        super().__init__()
        self.weight_ = float(weight)
        self.secondary_ = bool(secondary)


class AvoidJointLimitsGoal(_JointSetGoal):
    opcode = abi.GOAL_AVOID_JOINT_LIMITS


class CenterJointsGoal(_JointSetGoal):
    opcode = abi.GOAL_CENTER_JOINTS


class MinimalDisplacementGoal(_JointSetGoal):
    opcode = abi.GOAL_MINIMAL_DISPLACEMENT


class RegularizationGoal(Goal):
    opcode = abi.GOAL_REGULARIZATION

    def __init__(self, weight=1.0):
        super().__init__()
        self.weight_ = float(weight)


class JointVariableGoal(Goal):
    opcode = abi.GOAL_JOINT_VARIABLE

    def __init__(self, variable_name="", variable_position=0.0, weight=1.0, secondary=False):
        super().__init__()
        self.variable_name_ = variable_name
        self.variable_position = float(variable_position)
        self.weight_ = float(weight)
        self.secondary_ = bool(secondary)

    def getVariablePosition(self):
        return self.variable_position

    def setVariablePosition(self, p):
        self.variable_position = float(p)

    def getVariableName(self):
        return self.variable_name_

    def setVariableName(self, n):
        self.variable_name_ = n

    def variable_name(self):
        return self.variable_name_

    def params(self):
        return np.array([self.variable_position])


class _AxisDirectionGoal(LinkGoalBase):
    def __init__(self, link_name="", axis=(0, 0, 1), direction=(0, 0, 1), weight=1.0):
        super().__init__(link_name, weight)
        self.axis = _vec3(axis)  # constructors do not normalise (goal_types.h:596-601, 627-632); setters do
        self.direction = _vec3(direction)

    def setAxis(self, a):
        self.axis = _normalized(a)

    def setDirection(self, d):
        self.direction = _normalized(d)

    def getAxis(self):
        return self.axis

    def getDirection(self):
        return self.direction

    def params(self):
        return np.concatenate([self.axis, self.direction])


class SideGoal(_AxisDirectionGoal):
    opcode = abi.GOAL_SIDE


class DirectionGoal(_AxisDirectionGoal):
    opcode = abi.GOAL_DIRECTION


class ConeGoal(LinkGoalBase):
    opcode = abi.GOAL_CONE

    def __init__(self, link_name="", axis=(0, 0, 1), direction=(0, 0, 1), angle=0.0, weight=1.0, position=None,
                 position_weight=None):
        super().__init__(link_name, weight)
        # three reference constructors, goal_types.h:673-699
        self.position = _vec3(position if position is not None else (0, 0, 0))
        self.position_weight = float(position_weight if position_weight is not None else (1.0 if position is not None else 0.0))
        self.axis = _vec3(axis)
        self.direction = _vec3(direction)
        self.angle = float(angle)

    def setPosition(self, p):
        self.position = _vec3(p)

    def setPositionWeight(self, w):
        self.position_weight = float(w)

    def setAxis(self, a):
        self.axis = _normalized(a)

    def setDirection(self, d):
        self.direction = _normalized(d)

    def setAngle(self, a):
        self.angle = float(a)

    def params(self):
        return np.concatenate([self.position, [self.position_weight], self.axis, self.direction, [self.angle]])


class BalanceGoal(Goal):
    """goal_types.h:540-566, goal_types.cpp:231-272: keeps the centre of mass of the whole robot (every link with a URDF <inertial>)
    over `target`, measured perpendicular to `axis` (the direction of gravity).  The link masses are part of the robot model
    (RobotModel.add_link(mass=, com=) / the URDF reader); every link with mass becomes a tip of the problem."""
    opcode = abi.GOAL_BALANCE

    def __init__(self, target=(0, 0, 0), weight=1.0):
        super().__init__()
        self.target = _vec3(target)
        self.axis = _vec3((0, 0, 1))
        self.weight_ = float(weight)

    def getTarget(self):
        return self.target

    def getAxis(self):
        return self.axis

    def setTarget(self, t):
        self.target = _vec3(t)

    def setAxis(self, a):
        self.axis = _vec3(a)  # (the reference's setter does not normalise, goal_types.h:562)

    def params(self):
        return np.concatenate([self.target, self.axis])


class _HostOnlyGoal(Goal):
    """Goals evaluated through arbitrary host code in the reference: no device opcode."""
    opcode = None


class JointFunctionGoal(_HostOnlyGoal):
    def __init__(self, variable_names=(), function=None, weight=1.0, secondary=False):
        super().__init__()
        self.variable_names = list(variable_names)
        self.function = function
        self.weight_ = float(weight)
        self.secondary_ = bool(secondary)


class LinkFunctionGoal(_HostOnlyGoal):
    def __init__(self, link_name="", function=None, weight=1.0):
        super().__init__()
        self.link_name_ = link_name
        self.function = function
        self.weight_ = float(weight)


class BioIKKinematicsQueryOptions:
    """reference goal.h:121-129 (kinematics::KinematicsQueryOptions fields included)."""

    def __init__(self):
        self.goals = []
        self.fixed_joints = []
        self.replace = False
        self.solution_fitness = 0.0
        self.return_approximate_solution = False
