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
