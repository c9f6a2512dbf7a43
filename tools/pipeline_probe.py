"""host-pointer pipeline probe: time of every bioik_solve_batch_submit / _wait call for a stream of 4096-query batches, by schedule and solves in flight
usage: python tools/pipeline_probe.py"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from bio_ik_amd import PoseGoal, ProblemTemplate, abi, pr2_like  # noqa: E402
from bio_ik_amd.solver import HipSolver  # noqa: E402
from bio_ik_amd.workload import make_queries  # noqa: E402

t = ProblemTemplate(pr2_like(), "right_arm", [PoseGoal("r_wrist_roll_link")])
h = HipSolver(t, device=0)
batches = [make_queries(t, h.active_variables, h.fk_genes, 4096, seed=500 + k)[:2] for k in range(6)]
for schedule in ("latency", "throughput", "auto"):
    p = abi.default_solve_params(population=128, max_steps=64, random_seed=1, schedule=schedule)
    for npipe in (1, 3, 6):
        for i in range(6):
            h.wait_batch(h.submit_batch(p, *batches[i % 6]))
        pending, ts, tw = [], [], []
        t0 = time.perf_counter()
        n = 30
        for i in range(n):
            a = time.perf_counter()
            pending.append(h.submit_batch(p, *batches[i % 6]))
            ts.append(time.perf_counter() - a)
            if len(pending) == npipe:
                a = time.perf_counter()
                h.wait_batch(pending.pop(0))
                tw.append(time.perf_counter() - a)
        while pending:
            h.wait_batch(pending.pop(0))
        dt = (time.perf_counter() - t0) / n
        print("%-10s %d in flight: %.2f ms per batch = %.0f solves/s | submit %.2f ms (max %.2f) wait %.2f ms (max %.2f)" %
              (schedule, npipe, dt * 1e3, 4096 * 0.9965 / dt, np.mean(ts) * 1e3, np.max(ts) * 1e3, np.mean(tw) * 1e3, np.max(tw) * 1e3))
