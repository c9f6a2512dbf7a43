"""One pose per call through the HOST-pointer entry (bioik_solve_batch: what the plugin's searchPositionIK uses), PCIe inclusive, against the device-pointer entry"""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from bio_ik_amd import PoseGoal, ProblemTemplate, abi, pr2_like
from bio_ik_amd.solver import HipSolver
from bio_ik_amd.workload import make_queries
t = ProblemTemplate(pr2_like(), "right_arm", [PoseGoal("r_wrist_roll_link")])
h = HipSolver(t, device=0)
dev = torch.device("cuda", 0)
for n in (1, 16):
    p = abi.default_solve_params(population=128, max_steps=64, random_seed=1, islands=abi.ISLANDS_AUTO)
    sets = [make_queries(t, h.active_variables, h.fk_genes, n, seed=2000 + r)[:2] for r in range(40)]
    h.solve_batch(p, *sets[0]); h.solve_batch(p, *sets[0])
    ts = []
    stp = []
    for s_, p_ in sets:
        t0 = time.perf_counter(); r_ = h.solve_batch(p, s_, p_); ts.append(time.perf_counter() - t0); stp.append(int(r_[3].max()))
    k_ = int(np.argmax(ts))
    print("   slowest host call %d: %.3f ms, steps %d; the same queries again: %s ms" % (k_, 1e3 * ts[k_], stp[k_], [round(1e3 * (lambda t0: (h.solve_batch(p, *sets[k_]), time.perf_counter() - t0)[1])(time.perf_counter()), 3) for _ in range(3)]))
    o = (torch.empty((n, h.V), dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev))
    st = torch.cuda.Stream(dev)
    td = []
    for s_, p_ in sets:
        ds, dp = torch.from_numpy(s_).to(dev), torch.from_numpy(p_).to(dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        h.solve_batch_device(p, n, ds.data_ptr(), dp.data_ptr(), o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), st.cuda_stream); st.synchronize()
        td.append(time.perf_counter() - t0)
    print("   slowest host-pointer calls (ms):", [round(1e3 * x, 3) for x in sorted(ts)[-5:]], "at call", int(np.argmax(ts)), "slowest device-pointer calls:", [round(1e3 * x, 3) for x in sorted(td)[-5:]], "at call", int(np.argmax(td)))
    print("n %d: host-pointer entry %.3f ms per call (median %.3f), device-pointer entry %.3f ms (median %.3f)" % (n, 1e3 * np.mean(ts), 1e3 * np.median(ts), 1e3 * np.mean(td), 1e3 * np.median(td)), flush=True)
