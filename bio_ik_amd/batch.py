"""Multi-GPU form of the batched solve: one process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on
MI355X nodes, "gloo" in the CPU test-suite).  IK queries are independent (reference islands only meet in a best-of,
src/ik_parallel.h:220-269), so the batch is split into contiguous shards with NO data-path collective; the collectives
below only move the queries out and the results back when they originate on one rank:

    rank 0:  seeds [n][V], goal_params [n][P]  --scatter-->  shard r = [r*n/W, (r+1)*n/W)
    rank r:  bioik_problem_set_first_query(shard begin); bioik_solve_batch(shard)      (no communication)
    rank 0:  <--gather--  solutions [n][V], fitness, success, steps

Because the RNG stream of a query is keyed by its GLOBAL index, the sharded result is bit-identical to the unsharded one."""
import numpy as np


def shard_bounds(n, world):
    return [(r * n) // world for r in range(world + 1)]


def solve_sharded(solver, params, seeds=None, goal_params=None, n=None, device=None, group=None):
    """Collective call.  `seeds`/`goal_params` are needed on rank 0 only (NumPy, host); every rank passes its own
    `solver` (a HipSolver on its GPU) and the same `params`.  Returns (solutions, fitness, success, steps) on rank 0,
    None elsewhere.  `device` = torch device used for the exchange buffers (cuda:<local rank> with nccl, cpu with gloo)."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = torch.device(device) if device is not None else torch.device("cpu")
    V, P = solver.V, max(solver.P, 1)
    meta = torch.zeros(1, dtype=torch.int64, device=dev)
    if rank == 0:
        seeds = np.ascontiguousarray(seeds, dtype=np.float64).reshape(-1, V)
        n = seeds.shape[0]
        goal_params = np.ascontiguousarray(goal_params, dtype=np.float64).reshape(n, -1) if solver.P else np.zeros((n, 1))
        meta[0] = n
    dist.broadcast(meta, src=0, group=group)
    n = int(meta.item())
    b = shard_bounds(n, world)
    cap = max(b[r + 1] - b[r] for r in range(world))  # equal-size exchange buffers, padded
    mine = b[rank + 1] - b[rank]
    inbuf = torch.zeros((cap, V + P), dtype=torch.float64, device=dev)
    if rank == 0:
        full = torch.from_numpy(np.concatenate([seeds, goal_params], axis=1))
        chunks = []
        for r in range(world):
            c = torch.zeros((cap, V + P), dtype=torch.float64)
            c[:b[r + 1] - b[r]] = full[b[r]:b[r + 1]]
            chunks.append(c.to(dev))
        dist.scatter(inbuf, scatter_list=chunks, src=0, group=group)
    else:
        dist.scatter(inbuf, src=0, group=group)
    local = inbuf[:mine].cpu().numpy()
    solver.set_first_query(b[rank])
    sol, fit, suc, steps = solver.solve_batch(params, local[:, :V], local[:, V:V + solver.P]) if mine else (
        np.zeros((0, V)), np.zeros(0), np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.int32))
    solver.set_first_query(0)
    out = torch.zeros((cap, V + 3), dtype=torch.float64, device=dev)
    if mine:
        out[:mine] = torch.from_numpy(np.concatenate([sol, fit[:, None], suc[:, None].astype(np.float64), steps[:, None].astype(np.float64)], axis=1)).to(dev)
    if rank == 0:
        gathered = [torch.zeros_like(out) for _ in range(world)]
        dist.gather(out, gather_list=gathered, dst=0, group=group)
        res = np.concatenate([gathered[r][:b[r + 1] - b[r]].cpu().numpy() for r in range(world)], axis=0)
        return res[:, :V], res[:, V], res[:, V + 1].astype(np.int32), res[:, V + 2].astype(np.int32)
    dist.gather(out, dst=0, group=group)
    return None
