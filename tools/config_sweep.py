"""Throughput of the BASELINE.json configurations that are parity cases, not bench lines: C2 (global and tracking seeds), C3
(PR2 'all': two PoseGoals + secondary MinimalDisplacementGoal, pop=128), C4 (31-DOF snake: PoseGoal + secondary
AvoidJointLimitsGoal, pop=512); 4096 queries per launch, three launches in flight as in bench.py.  Prints one line per case."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bio_ik_amd import AvoidJointLimitsGoal, MinimalDisplacementGoal, PoseGoal, ProblemTemplate, abi, pr2_like, snake  # noqa: E402
from bio_ik_amd.solver import HipSolver  # noqa: E402
from bio_ik_amd.workload import make_queries  # noqa: E402


def sec(g):
    g.secondary_ = True
    return g


def run(name, template, pop, kind, max_steps, n=4096, steps=9, nfl=3):
    dev = torch.device("cuda", 0)
    h = HipSolver(template, device=0)
    seeds, params, _ = make_queries(template, h.active_variables, h.fk_genes, n, seed=0xB101C, kind=kind)
    p = abi.default_solve_params(population=pop, max_steps=max_steps, random_seed=1, fk_mode=abi.FK_EXACT)
    ds, dp = torch.from_numpy(seeds).to(dev), torch.from_numpy(params).to(dev)
    bufs = [(torch.empty((n, h.V), dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.float64, device=dev),
             torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev)) for _ in range(nfl)]
    streams = [torch.cuda.Stream(dev) for _ in range(nfl)]

    def launch(i):
        o = bufs[i % nfl]
        h.solve_batch_device(p, n, ds.data_ptr(), dp.data_ptr(), o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), streams[i % nfl].cuda_stream)

    for i in range(nfl):
        launch(i)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        launch(i)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    t0 = time.perf_counter()
    launch(0)
    torch.cuda.synchronize(dev)
    d1 = time.perf_counter() - t0
    suc, st = bufs[0][2].cpu().numpy(), bufs[0][3].cpu().numpy()
    print("%-22s D=%2d T=%d pop=%3d %-8s: %8.0f solves/s  %6.2f ms/batch (three in flight)  %6.2f ms alone  success %.4f  mean steps %.2f" %
          (name, h.D, h.T, pop, kind, suc.sum() / dt, dt * 1e3, d1 * 1e3, suc.mean(), st.mean()), flush=True)


if __name__ == "__main__":
    pr2 = pr2_like()
    c2 = ProblemTemplate(pr2, "right_arm", [PoseGoal("r_wrist_roll_link")])
    c3 = ProblemTemplate(pr2, "all", [PoseGoal("r_wrist_roll_link"), PoseGoal("l_wrist_roll_link"), sec(MinimalDisplacementGoal())])
    c4 = ProblemTemplate(snake(31), "snake", [PoseGoal("tip"), sec(AvoidJointLimitsGoal())])
    run("C2 right_arm", c2, 128, "global", 64)
    run("C2 right_arm", c2, 128, "tracking", 64)
    run("C3 all (2 tips + sec)", c3, 128, "global", 64)
    run("C3 all (2 tips + sec)", c3, 128, "tracking", 64)
    run("C4 snake31 (+ sec)", c4, 512, "global", 32)
    run("C4 snake31 (+ sec)", c4, 512, "tracking", 32)
