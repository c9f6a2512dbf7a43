"""bio_ik_amd — MI355X-native bio2_memetic IK solver (hot path of TAMS-Group/bio_ik) behind a C-ABI.

Host-side mirror of the reference interface for the batched hot path: goal classes (goals.py), robot model
(robot.py), problem template (problem.py), the HIP solver binding (solver.py) and the plugin-shaped front end
(plugin.py).  The compute lives in bio_ik_amd/csrc (HIP, gfx950) and is reached only through include/bioik_hip.h.
"""
from . import abi  # noqa: F401
from .goals import *  # noqa: F401,F403
from .problem import ProblemTemplate  # noqa: F401
from .robot import RobotModel, JointModelGroup, pr2_like, snake  # noqa: F401
from .urdf import load_urdf  # noqa: F401
from .plugin import BioIKKinematicsPlugin, KinematicsQueryOptions, MoveItErrorCodes  # noqa: F401
