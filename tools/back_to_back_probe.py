"""Three solves of one batch enqueued back to back on ONE stream (no host synchronisation between them), each into its own result arrays, then compared row by row
with the host-pointer solve (diagnostics: is the two-launch defect of graph replays, DESIGN.md section 8, a property of graphs or of queued solves?)."""
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
outs = [(torch.zeros((n, h.V), dtype=torch.float64, device=dev), torch.zeros(n, dtype=torch.float64, device=dev), torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.int32, device=dev)) for _ in range(4)]
ref = h.solve_batch(p, seeds, params)
s = torch.cuda.Stream(dev)
torch.cuda.synchronize()
for o in outs:
    h.solve_batch_device(p, n, ds.data_ptr(), dp.data_ptr(), o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), s.cuda_stream)
torch.cuda.synchronize()
for i, o in enumerate(outs):
    sol, suc, stp = o[0].cpu().numpy(), o[2].cpu().numpy(), o[3].cpu().numpy()
    bad = np.where((sol != ref[0]).any(axis=1))[0]
    print("queued", i, len(bad) == 0, "rows that differ:", len(bad), [(int(r), int(ref[3][r]), int(stp[r]), int(ref[2][r]), int(suc[r])) for r in bad[:6]], flush=True)
