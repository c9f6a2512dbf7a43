import sys, numpy as np, torch, time
sys.path.insert(0, '.')
from bio_ik_amd import PoseGoal, ProblemTemplate, abi, pr2_like
from bio_ik_amd.solver import HipSolver
from bio_ik_amd.workload import make_queries
t = ProblemTemplate(pr2_like(), "right_arm", [PoseGoal("r_wrist_roll_link")])
h = HipSolver(t, device=0)
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256  # (4096: the two-launch solve with its stream-ordered scratch inside the capture)
seeds, params, _ = make_queries(t, h.active_variables, h.fk_genes, n, seed=3)
p = abi.default_solve_params(population=128, max_steps=64, random_seed=1)
ds, dp = torch.from_numpy(seeds).to(dev), torch.from_numpy(params).to(dev)
o = (torch.empty((n, h.V), dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev))
ref = h.solve_batch(p, seeds, params)
s = torch.cuda.Stream(dev)
with torch.cuda.stream(s):
    h.solve_batch_device(p, n, ds.data_ptr(), dp.data_ptr(), o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), s.cuda_stream)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    h.solve_batch_device(p, n, ds.data_ptr(), dp.data_ptr(), o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), s.cuda_stream)
o[0].zero_()
g.replay(); torch.cuda.synchronize()
print("graph replay identical:", np.array_equal(o[0].cpu().numpy(), ref[0]), np.array_equal(o[2].cpu().numpy(), ref[2]))
