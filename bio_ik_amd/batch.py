"""Multi-GPU form of the batched solve: one process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on
MI355X nodes, "gloo" in the CPU test-suite).  IK queries are independent (reference islands only meet in a best-of,
src/ik_parallel.h:220-269), so the batch is split into contiguous shards with NO data-path collective; the collectives
below only move the queries out and the results back when they originate on one rank:

    rank 0:  seeds [n][V], goal_params [n][P]  --scatter-->  shard r = [r*n/W, (r+1)*n/W)
    rank r:  bioik_problem_set_first_query(shard begin); bioik_solve_batch[_device](shard)      (no communication)
    rank 0:  <--gather--  solutions [n][V], fitness, success, steps

Because the RNG stream of a query is keyed by its GLOBAL index, the sharded result is bit-identical to the unsharded one.
With the exchange buffers on a GPU (nccl) the shard never leaves HBM: the scattered rows are split into the seed / goal
arrays on the device, `bioik_solve_batch_device` runs on a HIP stream of its own, and the result rows are packed on the
device for the gather.  A MIXED batch (BASELINE.json configs[4]: PR2 and snake queries) is sorted by model into homogeneous
blocks (one problem handle each, every workgroup of a launch runs the same joint program); `solve_mixed` shards every block
over all ranks, so each GPU holds the same mix, and runs a rank's blocks concurrently on separate streams."""
import numpy as np


def shard_bounds(n, world):
    return [(r * n) // world for r in range(world + 1)]


class _Shard:
    """One block's exchange state on one rank."""
    pass


def _rank_world(group):
    """(rank, world) of the process group; a process without torch.distributed (one GPU, no launcher) is rank 0 of 1 and the
    collectives below degenerate to local copies"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 0, 1, False
    return dist.get_rank(group), dist.get_world_size(group), True


def _scatter(solver, seeds, goal_params, dev, group):
    import torch
    import torch.distributed as dist
    rank, world, live = _rank_world(group)
    V, P = solver.V, max(solver.P, 1)
    meta = torch.zeros(1, dtype=torch.int64, device=dev)
    if rank == 0:
        seeds = np.ascontiguousarray(seeds, dtype=np.float64).reshape(-1, V)
        n = seeds.shape[0]
        goal_params = np.ascontiguousarray(goal_params, dtype=np.float64).reshape(n, -1) if solver.P else np.zeros((n, 1))
        meta[0] = n
    if live:
        dist.broadcast(meta, src=0, group=group)
    s = _Shard()
    s.solver, s.V, s.P = solver, V, P
    s.n = int(meta.item())
    s.b = shard_bounds(s.n, world)
    s.cap = max(s.b[r + 1] - s.b[r] for r in range(world))  # equal-size exchange buffers, padded
    s.mine = s.b[rank + 1] - s.b[rank]
    s.first = s.b[rank]
    if not live:  # single process: the rows go straight to the device, no exchange buffer
        s.inbuf = torch.from_numpy(np.concatenate([seeds, goal_params], axis=1)).to(dev)
        return s
    s.inbuf = torch.zeros((s.cap, V + P), dtype=torch.float64, device=dev)
    if rank == 0:
        full = torch.from_numpy(np.concatenate([seeds, goal_params], axis=1))
        chunks = []
        for r in range(world):
            c = torch.zeros((s.cap, V + P), dtype=torch.float64, pin_memory=dev.type == "cuda")  # (page-locked: one DMA per shard, no staging copy)
            c[:s.b[r + 1] - s.b[r]] = full[s.b[r]:s.b[r + 1]]
            chunks.append(c.to(dev, non_blocking=dev.type == "cuda"))
        dist.scatter(s.inbuf, scatter_list=chunks, src=0, group=group)
    else:
        dist.scatter(s.inbuf, src=0, group=group)
    return s


def _launch(s, params, dev, stream=None):
    """Solve this rank's shard.  On a GPU: enqueue on `stream` and return (the rows are packed in _finish); on the CPU
    (gloo tests, host-simulator back end) the host-pointer entry point is synchronous."""
    import torch
    V, P = s.V, s.P
    s.stream = stream
    if dev.type == "cuda":
        # everything of this shard — the zero-filled result rows included — is allocated and written on the shard's own stream, which
        # first waits for what the current stream did to the exchange buffer (the scatter): no write is unordered against another
        stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(stream):
            s.out = torch.zeros((s.cap, V + 3), dtype=torch.float64, device=dev)
    else:
        s.out = torch.zeros((s.cap, V + 3), dtype=torch.float64, device=dev)
    if not s.mine:
        return
    s.solver.set_first_query(s.first)  # read when the launch is enqueued
    if dev.type == "cuda":
        with torch.cuda.stream(stream):
            s.d_seeds = s.inbuf[:s.mine, :V].contiguous()
            s.d_par = s.inbuf[:s.mine, V:V + P].contiguous()
            s.d_sol = torch.empty((s.mine, V), dtype=torch.float64, device=dev)
            s.d_fit = torch.empty(s.mine, dtype=torch.float64, device=dev)
            s.d_suc = torch.empty(s.mine, dtype=torch.int32, device=dev)
            s.d_steps = torch.empty(s.mine, dtype=torch.int32, device=dev)
            s.solver.solve_batch_device(params, s.mine, s.d_seeds.data_ptr(), s.d_par.data_ptr(), s.d_sol.data_ptr(), s.d_fit.data_ptr(),
                                        s.d_suc.data_ptr(), s.d_steps.data_ptr(), stream.cuda_stream)
            s.out[:s.mine, :V] = s.d_sol
            s.out[:s.mine, V] = s.d_fit
            s.out[:s.mine, V + 1] = s.d_suc.to(torch.float64)
            s.out[:s.mine, V + 2] = s.d_steps.to(torch.float64)
    else:
        local = s.inbuf[:s.mine].numpy()
        sol, fit, suc, steps = s.solver.solve_batch(params, local[:, :V], local[:, V:V + s.solver.P])
        s.out[:s.mine] = torch.from_numpy(np.concatenate([sol, fit[:, None], suc[:, None].astype(np.float64), steps[:, None].astype(np.float64)], axis=1))
    s.solver.set_first_query(0)


def _gather(s, dev, group):
    import torch
    import torch.distributed as dist
    rank, world, live = _rank_world(group)
    if dev.type == "cuda" and s.stream is not None:
        cur = torch.cuda.current_stream(dev)
        cur.wait_stream(s.stream)
        for tns in (s.out, getattr(s, "d_seeds", None), getattr(s, "d_par", None), getattr(s, "d_sol", None), getattr(s, "d_fit", None),
                    getattr(s, "d_suc", None), getattr(s, "d_steps", None)):
            if tns is not None:
                tns.record_stream(cur)  # allocated under the shard's stream, consumed (or freed) under the current one
    V = s.V
    if not live:
        res = s.out[:s.n].cpu().numpy()
        return res[:, :V], res[:, V], res[:, V + 1].astype(np.int32), res[:, V + 2].astype(np.int32)
    if rank == 0:
        gathered = [torch.zeros_like(s.out) for _ in range(world)]
        dist.gather(s.out, gather_list=gathered, dst=0, group=group)
        res = np.concatenate([gathered[r][:s.b[r + 1] - s.b[r]].cpu().numpy() for r in range(world)], axis=0)
        return res[:, :V], res[:, V], res[:, V + 1].astype(np.int32), res[:, V + 2].astype(np.int32)
    dist.gather(s.out, dst=0, group=group)
    return None


_STREAMS = {}


def _block_stream(dev, k):
    """the HIP stream block k of a mixed batch runs on: created once per (device, k) and kept (HIP multiplexes streams onto four hardware
    queues per device; a fresh stream per call would also pay its creation inside every solve)"""
    import torch
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), k)
    if key not in _STREAMS:
        _STREAMS[key] = torch.cuda.Stream(dev)
    return _STREAMS[key]


def _device(device):
    import torch
    return torch.device(device) if device is not None else torch.device("cpu")


def solve_sharded(solver, params, seeds=None, goal_params=None, n=None, device=None, group=None):
    """Collective call.  `seeds`/`goal_params` are needed on rank 0 only (NumPy, host); every rank passes its own
    `solver` (a HipSolver on its GPU) and the same `params`.  Returns (solutions, fitness, success, steps) on rank 0,
    None elsewhere.  `device` = torch device used for the exchange buffers (cuda:<local rank> with nccl, cpu with gloo)."""
    res = solve_mixed([(solver, params, seeds, goal_params)], device=device, group=group)
    return None if res is None else res[0]


def solve_mixed(blocks, device=None, group=None):
    """Collective call over a batch sorted by model: `blocks` = [(solver, params, seeds, goal_params), ...], one homogeneous
    block per problem template (the same list of solvers and params on every rank; seeds / goal_params on rank 0 only).
    Every block is sharded over all ranks; a rank's blocks run concurrently (one HIP stream each).  Returns on rank 0 a list
    with one (solutions, fitness, success, steps) per block, None elsewhere."""
    import torch
    dev = _device(device)
    shards = [_scatter(solver, seeds, gp, dev, group) for solver, _, seeds, gp in blocks]
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)  # the scattered rows are complete before the solver streams read them
    for k, (s, (_, params, _, _)) in enumerate(zip(shards, blocks)):
        _launch(s, params, dev, _block_stream(dev, k) if dev.type == "cuda" else None)
    out = [_gather(s, dev, group) for s in shards]
    return None if out[0] is None else out
