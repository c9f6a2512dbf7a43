"""Worker of tests/test_batch_gloo.py: world_size-2 run of bio_ik_amd.batch.solve_sharded over gloo (CPU).
The compute back end is the host simulator of the kernels (TEST INFRASTRUCTURE, injected through HipSolver(lib=...))."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch.distributed as dist
    from bio_ik_amd import PoseGoal, ProblemTemplate, abi, pr2_like, solver
    from bio_ik_amd.batch import solve_sharded
    from bio_ik_amd.workload import make_queries
    dist.init_process_group(backend="gloo")
    rank = dist.get_rank()
    lib = solver.load_library(os.path.join(ROOT, "tests", "hostsim", "libbioik_hostsim.so"))
    t = ProblemTemplate(pr2_like(), "right_arm", [PoseGoal("r_wrist_roll_link")])
    h = solver.HipSolver(t, lib=lib)
    p = abi.default_solve_params(population=16, max_steps=2, random_seed=5)
    n = 5
    seeds = params = None
    if rank == 0:
        seeds, params, _ = make_queries(t, h.active_variables, h.fk_genes, n, seed=21)
    res = solve_sharded(h, p, seeds, params)
    if rank == 0:
        whole = h.solve_batch(p, seeds, params)
        ok = all(np.array_equal(a, b) for a, b in zip(res, whole))
        np.save(sys.argv[1], np.array([1 if ok else 0, dist.get_world_size()]))
    else:
        assert res is None
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
