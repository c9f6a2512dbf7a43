"""Per-phase shader cycles of a solve of C3 / C4 (profiling build: hipcc ... -DBIOIK_PHASE_TIMING -o build/libphase.so).
usage: BIOIK_HIP_LIBRARY=build/libphase.so python tools/phase_probe_config.py [c2|ref|c3|c4] [queries] [latency|throughput]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bio_ik_amd import AvoidJointLimitsGoal, MinimalDisplacementGoal, PoseGoal, ProblemTemplate, abi, pr2_like, snake  # noqa: E402
from bio_ik_amd.solver import HipSolver  # noqa: E402
from bio_ik_amd.workload import make_queries  # noqa: E402

NAMES = ["init", "reproduce", "fitness", "preselect.score", "preselect.sort", "species", "check", "preselect/candidate", "sel.top2", "sel.xwave", "sel.copy", "sel.barrier",
         "mem.approx", "mem.grad", "mem.norm", "mem.line", "mem.accept", "mem.tail", "rank", "#mem_iter", "#steps", "linearise", "mem.support_cols", "mem.support_eval"]


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 3072
    if cfg == "c2":
        t, pop, steps = ProblemTemplate(pr2_like(), "right_arm", [PoseGoal("r_wrist_roll_link")]), 128, 32
    elif cfg == "ref":  # C2 at the reference's own parameters: 16 children per species, linearised phenotypes
        t, pop, steps = ProblemTemplate(pr2_like(), "right_arm", [PoseGoal("r_wrist_roll_link")]), 16, 32
    elif cfg == "c3":
        t, pop, steps = ProblemTemplate(pr2_like(), "all", [PoseGoal("r_wrist_roll_link"), PoseGoal("l_wrist_roll_link"), MinimalDisplacementGoal()]), 128, 16
    else:
        t, pop, steps = ProblemTemplate(snake(31), "snake", [PoseGoal("tip"), AvoidJointLimitsGoal()]), 512, 8
    h = HipSolver(t, device=0)
    seeds, params, _ = make_queries(t, h.active_variables, h.fk_genes, n, seed=5)
    p = abi.default_solve_params(population=pop, max_steps=steps, random_seed=1, fk_mode=abi.FK_LINEAR if cfg == "ref" else abi.FK_EXACT,
                                 schedule=sys.argv[3] if len(sys.argv) > 3 else "latency")
    p.dtwist = 1e-300  # no query may succeed: every workgroup runs the whole budget
    path = "/tmp/phase_%s.bin" % cfg
    os.environ["BIOIK_PHASE_DUMP"] = path
    h.solve_batch(p, seeds, params)
    a = np.fromfile(path, dtype=np.uint64).reshape(-1, 28).astype(np.float64)
    m = a.mean(axis=0)
    tot = m[:19].sum() + m[21] + m[22] + m[23]
    wall = (a[:, 25] - a[:, 24]).mean() * 10.0
    print("== %s, %d queries, %d steps each: %.1f us per step wall, %.0f shader cycles per step (lane 0 of the workgroup)" % (cfg, n, steps, wall / 1e3 / m[20], tot / m[20]))
    for i, name in enumerate(NAMES):
        if name.startswith("#") or m[i] == 0:
            continue
        print("   %-20s %9.0f cycles/step %5.1f%%" % (name, m[i] / m[20], 100 * m[i] / tot))


if __name__ == "__main__":
    main()
