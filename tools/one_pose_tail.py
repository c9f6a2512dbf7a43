"""round 6: the slowest calls of bench.py's one_pose_timeouts leg (7-DOF arm, PoseGoal, one pose per plugin call): which calls, how long, did they succeed.
usage: python tools/one_pose_tail.py [timeout_ms] [gpu_max_steps]"""
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

timeout = 1e-3 * float(sys.argv[1]) if len(sys.argv) > 1 else 0.001
max_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
model = pr2_like()
template = ProblemTemplate(model, "right_arm", [PoseGoal("r_wrist_roll_link")])
h = HipSolver(template, device=0)
n = 512
seeds, params, _ = make_queries(template, h.active_variables, h.fk_genes, n, seed=0x0E905E)
plug = BioIKKinematicsPlugin()
plug.initialize(model, "right_arm", model.link_names[0], ["r_wrist_roll_link"], params={"random_seed": 1, "gpu_devices": [0], "gpu_max_steps": max_steps})
gv = plug._group_vars
poses = params[:, None, 0:7].copy()
opts = BioIKKinematicsQueryOptions()
plug.searchPositionIKEach(poses[:8], seeds[:8, gv], opts, timeout=0.02)
for rep in range(3):
    _, ok, _, sec = plug.searchPositionIKEach(poses, seeds[:, gv], opts, timeout=timeout)
    order = np.argsort(-sec)[:12]
    print("pass %d, timeout %.1f ms, max_steps %d: mean %.3f ms, median %.3f, p99 %.3f, success %.3f" % (rep, 1e3 * timeout, max_steps, 1e3 * sec.mean(), 1e3 * np.median(sec), 1e3 * np.quantile(sec, 0.99), ok.mean()))
    print("   slowest: " + ", ".join("#%d %.2f ms %s" % (i, 1e3 * sec[i], "ok" if ok[i] else "--") for i in order), flush=True)
