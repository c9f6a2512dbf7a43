#!/bin/bash
# round 4, GPU session 7: the reference's own parameters (pop 16, linearised FK) under other lane mappings (diagnostic switches), fixed work:
# what a dense variant for small populations could give
O=gpurun_out/r04s7; mkdir -p $O
probe() { python - "$@" <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from bio_ik_amd import PoseGoal, ProblemTemplate, abi, pr2_like
from bio_ik_amd.solver import HipSolver
from bio_ik_amd.workload import make_queries
t = ProblemTemplate(pr2_like(), "right_arm", [PoseGoal("r_wrist_roll_link")])
h = HipSolver(t, device=0)
n = int(sys.argv[1]); pop = int(sys.argv[2])
seeds, params, _ = make_queries(t, h.active_variables, h.fk_genes, n, seed=5)
dev = torch.device("cuda", 0)
ds, dp = torch.from_numpy(seeds).to(dev), torch.from_numpy(params).to(dev)
o = (torch.empty((n, h.V), dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev))
p = abi.default_solve_params(population=pop, max_steps=32, random_seed=1, fk_mode=abi.FK_LINEAR)
p.dtwist = 1e-300
st = torch.cuda.Stream(dev)
def go():
    h.solve_batch_device(p, n, ds.data_ptr(), dp.data_ptr(), o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), st.cuda_stream)
for _ in range(3): go()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): go()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print("pop %d linear, %d queries x 32 steps: %.3f ms -> %.0f steps/ms   env %s" % (pop, n, dt * 1e3, n * 32 / dt / 1e3, {k: v for k, v in os.environ.items() if k.startswith("BIOIK_SOLVE")}))
PY
}
for n in 4096 8192; do
probe $n 16
BIOIK_SOLVE_THREADS=64 BIOIK_SOLVE_SPECIES_PARALLEL=1 BIOIK_SOLVE_COLUMNLESS=1 probe $n 16
BIOIK_SOLVE_THREADS=64 BIOIK_SOLVE_SPECIES_PARALLEL=1 BIOIK_SOLVE_STORE_CHILDREN=0 probe $n 16
BIOIK_SOLVE_THREADS=64 BIOIK_SOLVE_SPECIES_PARALLEL=0 probe $n 16
probe $n 32
BIOIK_SOLVE_THREADS=64 BIOIK_SOLVE_SPECIES_PARALLEL=1 BIOIK_SOLVE_COLUMNLESS=1 probe $n 32
done 2>&1 | grep -v amdgpu.ids | tee $O/ref_params_mappings.log
