import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pr2():
    from bio_ik_amd import pr2_like
    return pr2_like()


@pytest.fixture(scope="session")
def snake31():
    from bio_ik_amd import snake
    return snake(31)


@pytest.fixture(scope="session")
def templates(pr2, snake31):
    """The three problem templates of BASELINE.json configs C2, C3, C4."""
    from bio_ik_amd import AvoidJointLimitsGoal, MinimalDisplacementGoal, PoseGoal, ProblemTemplate
    return {
        "c2": ProblemTemplate(pr2, "right_arm", [PoseGoal("r_wrist_roll_link")]),
        "c3": ProblemTemplate(pr2, "all", [PoseGoal("r_wrist_roll_link"), PoseGoal("l_wrist_roll_link"), MinimalDisplacementGoal()]),
        "c4": ProblemTemplate(snake31, "snake", [PoseGoal("tip"), AvoidJointLimitsGoal()]),
    }


@pytest.fixture(scope="session")
def oracles(templates):
    from oracle import orc
    return {k: orc.Oracle(t) for k, t in templates.items()}


def random_configuration(model, rng, n=None):
    """uniform in [min,max] per variable (continuous joints: [-pi,pi]); reference README.md:410-418"""
    lo = np.asarray(model.var_min)
    hi = np.asarray(model.var_max)
    shape = (model.n_variables,) if n is None else (n, model.n_variables)
    return lo + (hi - lo) * rng.random(shape)
