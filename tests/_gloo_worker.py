"""Worker of tests/test_batch_gloo.py: world_size-2 run of bio_ik_amd.batch.solve_sharded / solve_mixed over gloo (CPU).
The compute back end is the host simulator of the kernels (TEST INFRASTRUCTURE, injected through HipSolver(lib=...))."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch.distributed as dist
    from bio_ik_amd import AvoidJointLimitsGoal, PoseGoal, ProblemTemplate, abi, pr2_like, snake, solver
    from bio_ik_amd.batch import solve_mixed, solve_sharded
    from bio_ik_amd.workload import make_queries
    backend = os.environ.get("BIOIK_WORKER_BACKEND", "gloo")
    dist.init_process_group(backend=backend)
    rank = dist.get_rank()
    if backend == "nccl":  # on a GPU box (tests/test_gpu_parity.py): the product library, exchange buffers and shards stay in HBM
        import torch
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        lib, device = None, "cuda:%d" % int(os.environ.get("LOCAL_RANK", "0"))
    else:
        lib, device = solver.load_library(os.path.join(ROOT, "tests", "hostsim", "libbioik_hostsim.so")), None
    t = ProblemTemplate(pr2_like(), "right_arm", [PoseGoal("r_wrist_roll_link")])
    h = solver.HipSolver(t, lib=lib) if lib is not None else solver.HipSolver(t, device=int(os.environ.get("LOCAL_RANK", "0")))
    p = abi.default_solve_params(population=16, max_steps=2, random_seed=5)
    n = 5
    seeds = params = None
    if rank == 0:
        seeds, params, _ = make_queries(t, h.active_variables, h.fk_genes, n, seed=21)
    res = solve_sharded(h, p, seeds, params, device=device)
    # a mixed batch sorted by model (BASELINE.json configs[4] in miniature): a PR2 block and a snake block, 3 + 4 queries
    t2 = ProblemTemplate(snake(6), "snake", [PoseGoal("tip"), AvoidJointLimitsGoal()])
    h2 = solver.HipSolver(t2, lib=lib) if lib is not None else solver.HipSolver(t2, device=int(os.environ.get("LOCAL_RANK", "0")))
    p2 = abi.default_solve_params(population=8, max_steps=2, random_seed=9)
    seeds2 = params2 = None
    if rank == 0:
        seeds2, params2, _ = make_queries(t2, h2.active_variables, h2.fk_genes, 4, seed=22)
    mixed = solve_mixed([(h, p, None if seeds is None else seeds[:3], None if params is None else params[:3]), (h2, p2, seeds2, params2)], device=device)
    if rank == 0:
        whole = h.solve_batch(p, seeds, params)
        ok = all(np.array_equal(a, b) for a, b in zip(res, whole))
        ok = ok and all(np.array_equal(a, b) for a, b in zip(mixed[0], h.solve_batch(p, seeds[:3], params[:3])))
        ok = ok and all(np.array_equal(a, b) for a, b in zip(mixed[1], h2.solve_batch(p2, seeds2, params2)))
        np.save(sys.argv[1], np.array([1 if ok else 0, dist.get_world_size()]))
        if backend == "nccl":  # the result is on disk; do not wait for the communicator's teardown
            sys.stdout.flush()
            os._exit(0)
    else:
        assert res is None and mixed is None
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
