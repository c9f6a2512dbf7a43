"""One query per call: time against the step budget (no query may succeed: dtwist tiny), islands 1 and 8 -- the slope is the lone step, the intercept what a
call costs besides its steps (launches, the synchronisation, the kernel's own prologue / epilogue)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from bio_ik_amd import PoseGoal, ProblemTemplate, abi, pr2_like
from bio_ik_amd.solver import HipSolver
from bio_ik_amd.workload import make_queries
t = ProblemTemplate(pr2_like(), "right_arm", [PoseGoal("r_wrist_roll_link")])
h = HipSolver(t, device=0)
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
seeds, params, _ = make_queries(t, h.active_variables, h.fk_genes, n, seed=5)
ds, dp = torch.from_numpy(seeds).to(dev), torch.from_numpy(params).to(dev)
o = (torch.empty((n, h.V), dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev))
s = torch.cuda.Stream(dev)
for islands in (1, 8):
    xs, ys, ks = [], [], []
    for steps in (1, 2, 4, 8, 16, 32):
        p = abi.default_solve_params(population=128, max_steps=steps, random_seed=1, islands=islands, island_sync=1 if islands > 1 else 0, dtwist=1e-300)
        def call():
            h.solve_batch_device(p, n, ds.data_ptr(), dp.data_ptr(), o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), s.cuda_stream)
        for _ in range(5): call(); s.synchronize()
        ts, ev = [], []
        for _ in range(30):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            with torch.cuda.stream(s):
                e0.record(s); call(); e1.record(s)
            s.synchronize(); ts.append(time.perf_counter() - t0); ev.append(e0.elapsed_time(e1))
        xs.append(steps), ys.append(1e3 * np.median(ts)), ks.append(1e3 * np.median(ev))
        print("n %d islands %d steps %2d: wall %.1f us (min %.1f), events %.1f us" % (n, islands, steps, 1e3 * np.median(ts), 1e3 * min(ts), 1e3 * np.median(ev)), flush=True)
    a, b = np.polyfit(xs, ys, 1); c, d = np.polyfit(xs, ks, 1)
    print("n %d islands %d: wall = %.1f us + %.1f us per step; events = %.1f us + %.1f us per step" % (n, islands, b, a, d, c), flush=True)
