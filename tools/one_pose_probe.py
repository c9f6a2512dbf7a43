"""round 6: the latency of ONE pose per plugin call (bioik_plugin_search_each), call by call: mean, percentiles, and where the slow calls are -- their indices and times.
usage: python tools/one_pose_probe.py [timeout_ms] [calls] [arm|arm_md|all|snake] [gpu_islands] [gpu_population] [gpu_max_steps] [exact|linear]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bio_ik_amd import AvoidJointLimitsGoal, MinimalDisplacementGoal, PoseGoal, ProblemTemplate, pr2_like, snake  # noqa: E402
from bio_ik_amd.goals import BioIKKinematicsQueryOptions  # noqa: E402
from bio_ik_amd.plugin import BioIKKinematicsPlugin  # noqa: E402
from bio_ik_amd.solver import HipSolver  # noqa: E402
from bio_ik_amd.workload import make_queries  # noqa: E402


def main():
    timeout = float(sys.argv[1]) * 1e-3 if len(sys.argv) > 1 else 0.005
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    which = sys.argv[3] if len(sys.argv) > 3 else "arm"
    islands = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    pop = int(sys.argv[5]) if len(sys.argv) > 5 else 128
    max_steps = int(sys.argv[6]) if len(sys.argv) > 6 else 4096
    fk = sys.argv[7] if len(sys.argv) > 7 else "exact"
    model, group, tips, extra = {"arm": (pr2_like(), "right_arm", ["r_wrist_roll_link"], []),
                                 "arm_md": (pr2_like(), "right_arm", ["r_wrist_roll_link"], [MinimalDisplacementGoal()]),
                                 "all": (pr2_like(), "all", ["r_wrist_roll_link", "l_wrist_roll_link"], [MinimalDisplacementGoal()]),
                                 "snake": (snake(31), "snake", ["tip"], [AvoidJointLimitsGoal()])}[which]
    template = ProblemTemplate(model, group, [PoseGoal(t) for t in tips] + extra)
    h = HipSolver(template, device=0)
    seeds, params, _ = make_queries(template, h.active_variables, h.fk_genes, n, seed=0x0E905E)
    plug = BioIKKinematicsPlugin()
    plug.initialize(model, group, model.link_names[0], tips, params={"random_seed": 1, "gpu_max_steps": max_steps, "gpu_islands": islands, "gpu_population": pop, "gpu_fk": fk})
    gv = plug._group_vars
    poses = np.stack([params[:, 8 * t:8 * t + 7] for t in range(len(tips))], axis=1)
    opts = BioIKKinematicsQueryOptions()
    opts.goals = list(extra)
    plug.searchPositionIKEach(poses[:8], seeds[:8, gv], opts, timeout=0.02)
    print("%s, gpu_islands %d, gpu_population %d, gpu_max_steps %d, gpu_fk %s" % (which, islands, pop, max_steps, fk))
    for rep in range(2):
        _, ok, _, sec = plug.searchPositionIKEach(poses, seeds[:, gv], opts, timeout=timeout)
        order = np.argsort(sec)[::-1][:8]
        print("timeout %.1f ms: success %.4f mean %.3f ms median %.3f p90 %.3f p99 %.3f max %.3f | slowest calls (index: ms): %s" % (
            1e3 * timeout, ok.mean(), 1e3 * sec.mean(), 1e3 * np.median(sec), 1e3 * np.quantile(sec, 0.9), 1e3 * np.quantile(sec, 0.99), 1e3 * sec.max(),
            ", ".join("%d: %.2f" % (i, 1e3 * sec[i]) for i in order)), flush=True)


if __name__ == "__main__":
    main()
