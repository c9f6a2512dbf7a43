"""N>1 path on CPU: two processes over gloo run the sharded batch solve (bio_ik_amd/batch.py) — uneven shards,
query-indexed RNG streams — and rank 0 checks the gathered result against the unsharded solve, bit for bit."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sharded_solve_over_gloo(hostsim_lib, tmp_path):
    out = str(tmp_path / "result.npy")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29631",
           os.path.join(ROOT, "tests", "_gloo_worker.py"), out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = np.load(out)
    assert res[0] == 1 and res[1] == 2


def test_shard_bounds():
    from bio_ik_amd.batch import shard_bounds
    assert shard_bounds(10, 4) == [0, 2, 5, 7, 10]
    assert shard_bounds(262144, 8)[1] == 32768
    assert shard_bounds(3, 8)[-1] == 3


def test_solve_mixed_without_a_process_group(hostsim_lib):
    """one process, no launcher: the batch layer degenerates to local copies (rank 0 of 1) and still returns, block by block, what the
    per-block solves return"""
    from bio_ik_amd import AvoidJointLimitsGoal, PoseGoal, ProblemTemplate, abi, pr2_like, snake
    from bio_ik_amd.batch import solve_mixed
    from bio_ik_amd.solver import HipSolver
    from bio_ik_amd.workload import make_queries
    t1 = ProblemTemplate(pr2_like(), "right_arm", [PoseGoal("r_wrist_roll_link")])
    t2 = ProblemTemplate(snake(6), "snake", [PoseGoal("tip"), AvoidJointLimitsGoal()])
    h1, h2 = HipSolver(t1, lib=hostsim_lib), HipSolver(t2, lib=hostsim_lib)
    p1 = abi.default_solve_params(population=16, max_steps=2, random_seed=5)
    p2 = abi.default_solve_params(population=8, max_steps=2, random_seed=9)
    s1, g1, _ = make_queries(t1, h1.active_variables, h1.fk_genes, 3, seed=21)
    s2, g2, _ = make_queries(t2, h2.active_variables, h2.fk_genes, 4, seed=22)
    mixed = solve_mixed([(h1, p1, s1, g1), (h2, p2, s2, g2)])
    assert all(np.array_equal(a, b) for a, b in zip(mixed[0], h1.solve_batch(p1, s1, g1)))
    assert all(np.array_equal(a, b) for a, b in zip(mixed[1], h2.solve_batch(p2, s2, g2)))
