"""Back-to-back eager solves of one batch on ONE stream, each checked against the host-pointer solve (diagnostics for the hand-over when the chip runs empty)."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from bio_ik_amd import PoseGoal, ProblemTemplate, abi, pr2_like
from bio_ik_amd.solver import HipSolver
from bio_ik_amd.workload import make_queries
t = ProblemTemplate(pr2_like(), "right_arm", [PoseGoal("r_wrist_roll_link")])
h = HipSolver(t, device=0)
dev = torch.device("cuda", 0)
n = int(sys.argv[1])
seeds, params, _ = make_queries(t, h.active_variables, h.fk_genes, n, seed=3)
p = abi.default_solve_params(population=128, max_steps=64, random_seed=1)
ds, dp = torch.from_numpy(seeds).to(dev), torch.from_numpy(params).to(dev)
o = (torch.empty((n, h.V), dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev))
ref = h.solve_batch(p, seeds, params)
s = torch.cuda.Stream(dev)
for i in range(4):
    o[0].zero_(), o[2].zero_(), o[3].zero_()
    torch.cuda.synchronize()
    h.solve_batch_device(p, n, ds.data_ptr(), dp.data_ptr(), o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), s.cuda_stream)
    torch.cuda.synchronize()
    sol, suc, stp = o[0].cpu().numpy(), o[2].cpu().numpy(), o[3].cpu().numpy()
    bad = np.where((sol != ref[0]).any(axis=1))[0]
    print("eager", i, len(bad) == 0, "rows that differ:", len(bad), [(int(r), int(ref[3][r]), int(stp[r]), int(ref[2][r]), int(suc[r])) for r in bad[:6]], flush=True)
