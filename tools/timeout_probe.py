"""round 6: how long does ONE pose per plugin call take when its goal cannot be reached -- i.e. how tightly the caller's timeout bounds the call (ik_parallel.h:160)?
usage: python tools/timeout_probe.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bio_ik_amd import PoseGoal, ProblemTemplate, pr2_like  # noqa: E402
from bio_ik_amd.goals import BioIKKinematicsQueryOptions  # noqa: E402
from bio_ik_amd.plugin import BioIKKinematicsPlugin  # noqa: E402
from bio_ik_amd.solver import HipSolver  # noqa: E402
from bio_ik_amd.workload import make_queries  # noqa: E402

model = pr2_like()
template = ProblemTemplate(model, "right_arm", [PoseGoal("r_wrist_roll_link")])
h = HipSolver(template, device=0)
n = 64
seeds, params, _ = make_queries(template, h.active_variables, h.fk_genes, n, seed=5)
plug = BioIKKinematicsPlugin()
plug.initialize(model, "right_arm", model.link_names[0], ["r_wrist_roll_link"], params={"random_seed": 1})
gv = plug._group_vars
poses = params[:, None, 0:7].copy()
poses[:, 0, :3] += 10.0  # unreachable
opts = BioIKKinematicsQueryOptions()
plug.searchPositionIKEach(poses[:4], seeds[:4, gv], opts, timeout=0.02)
for timeout in (0.0005, 0.001, 0.002, 0.005, 0.02):
    _, ok, _, sec = plug.searchPositionIKEach(poses, seeds[:, gv], opts, timeout=timeout)
    print("timeout %5.1f ms: calls take mean %.3f ms, min %.3f, max %.3f (success %.2f)" % (1e3 * timeout, 1e3 * sec.mean(), 1e3 * sec.min(), 1e3 * sec.max(), ok.mean()), flush=True)
