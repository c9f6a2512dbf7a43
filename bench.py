#!/usr/bin/env python
"""bench.py — IK solves/sec of the MI355X bio2_memetic hot path (BASELINE.json metric).

A "step" is one pass of the hot path over one batch: `bioik_solve_batch_device` on 4096 independent PR2-like
right-arm 7-DOF PoseGoal queries (BASELINE.json configs[1]: pop=128, 1xMI355X), inputs already resident in HBM.
`value` = successful solves of all ranks / wall time of the K timed steps (max over ranks).  N>1: one process per
GPU (torch.distributed, backend nccl = RCCL), every rank solves its own 4096-query shard (weak scaling, no data-path
collective: queries are independent; RCCL only carries the barrier and the two scalar reductions of the timing).
Consecutive steps are issued round-robin on three HIP streams (`--in-flight`, config.batches_in_flight): the tail of one
launch — a handful of queries that use the whole step budget, 64 sequential steps wherever they start — overlaps the bulk
of the next ones; `one_batch_at_a_time` holds the same measurement with strictly one launch after the other
(profiles/r01_inflight_sweep.log: 1 / 2 / 3 in flight).

Extra objects on the JSON line:
  roofline     dominant kernel k_solve against the HBM roofline: ALGORITHMIC bytes per launch (SURVEY.md §8d,
               B_gen = 8*(pop*(3D+1) + 8D) per generation per species per query, from the device step counters)
               / mean launch duration measured with HIP events on the launch stream, vs 8 TB/s.  The fused kernel
               keeps the population in LDS, so the measured HBM traffic (profiles/) is far below the algorithmic
               figure; the kernel is FP64-VALU bound.
  cpu_baseline the reference's own CPU code (oracle/_ref: the reference sources compiled unmodified, Release flags) when the
               prebuilt library is present, else the oracle port; timed on this host, rank 0, N=1 only, one thread, on a
               bounded sample of the same queries.  The port at the GPU run's own parameters is reported beside it.
  tracking_seeds  the same template and parameters on seeds near the target (SURVEY.md section 8(d)'s second workload).
  reference_parameters  the GPU on the same queries at the reference's own parameters (its population, its linearised
               phenotypes): the like-for-like figure next to cpu_baseline.value.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BATCH = int(os.environ.get("BIOIK_BENCH_BATCH", "4096"))      # experiments only: the reported metric uses the defaults
POP = int(os.environ.get("BIOIK_BENCH_POP", "128"))
FK_MODE = os.environ.get("BIOIK_BENCH_FK", "exact")
MAX_STEPS = int(os.environ.get("BIOIK_BENCH_MAX_STEPS", "64"))
HBM_PEAK = 8.0e12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--timed-only", action="store_true",
                    help="only the warm-up and the K timed steps (no one-at-a-time leg, no secondary measurements): what the rocprofv3 kernel "
                         "statistics in profiles/ are taken over, so that their average duration is that of the timed launches")
    ap.add_argument("--cpu-sample", type=int, default=2048)
    ap.add_argument("--in-flight", type=int, default=int(os.environ.get("BIOIK_BENCH_IN_FLIGHT", "3")),
                    help="batches in flight: consecutive steps are issued round-robin on this many HIP streams (1 = strictly one after the other)")
    args = ap.parse_args()

    import numpy as np
    import torch

    from bio_ik_amd import PoseGoal, ProblemTemplate, abi, pr2_like
    from bio_ik_amd.solver import HipSolver
    from bio_ik_amd.workload import make_queries

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: bio_ik_amd has no CPU compute path")
    # one rank per GPU; BIOIK_BENCH_BACKEND=gloo lets several ranks share the GPUs of a smaller box (RCCL refuses two ranks on one
    # device): a dry run of the N>1 control flow, not a measurement
    backend = os.environ.get("BIOIK_BENCH_BACKEND", "nccl")
    gpu = local_rank % torch.cuda.device_count()
    if world > 1:
        torch.cuda.set_device(gpu)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", gpu))
        else:
            dist.init_process_group(backend=backend)
    dev = torch.device("cuda", gpu)
    torch.cuda.set_device(dev)

    template = ProblemTemplate(pr2_like(), "right_arm", [PoseGoal("r_wrist_roll_link")])
    h = HipSolver(template, device=gpu)
    D, V, T = h.D, h.V, h.T
    # synthetic queries of the reference's own self-test recipe (README.md:410-418); every rank draws its own shard
    seeds, params, _ = make_queries(template, h.active_variables, h.fk_genes, BATCH, seed=0xB101C + rank)
    h.set_first_query(rank * BATCH)
    p = abi.default_solve_params(population=POP, max_steps=MAX_STEPS, random_seed=1, fk_mode=abi.FK_EXACT if FK_MODE == "exact" else abi.FK_LINEAR)
    if "BIOIK_BENCH_DTWIST" in os.environ:  # experiments only (e.g. 1e-300: no query ever succeeds, every workgroup runs max_steps)
        p.dtwist = float(os.environ["BIOIK_BENCH_DTWIST"])

    d_seeds = torch.from_numpy(seeds).to(dev)
    d_params = torch.from_numpy(params).to(dev)
    # Consecutive steps go round-robin to `in_flight` HIP streams with their own result buffers: a launch of 4096 queries
    # ends with a long tail (the few queries that use the whole step budget run ~13 ms each) during
    # which the chip is nearly empty; with more launches in flight the next batches' bulk fills it.  Every step is a complete
    # pass of the hot path over one batch and every batch's results are complete when the timed region ends.
    nfl = max(1, args.in_flight)
    streams = [torch.cuda.Stream(dev) for _ in range(nfl)]
    bufs = [(torch.empty((BATCH, V), dtype=torch.float64, device=dev), torch.empty(BATCH, dtype=torch.float64, device=dev),
             torch.empty(BATCH, dtype=torch.int32, device=dev), torch.empty(BATCH, dtype=torch.int32, device=dev)) for _ in range(nfl)]
    torch.cuda.synchronize(dev)

    def step(i):
        o, st = bufs[i % nfl], streams[i % nfl]
        h.solve_batch_device(p, BATCH, d_seeds.data_ptr(), d_params.data_ptr(), o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(),
                             st.cuda_stream)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed(n_steps, n_warm):
        for i in range(n_warm):
            step(i)
        barrier()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_steps)]
        t0 = time.perf_counter()
        for i, (a, b) in enumerate(ev):
            a.record(streams[i % nfl])  # events on the stream the kernel is launched on
            step(i)
            b.record(streams[i % nfl])
        barrier()
        el = time.perf_counter() - t0
        return el, (float(np.mean([a.elapsed_time(b) for a, b in ev])) if ev else 0.0)

    elapsed, kernel_ms = timed(args.steps, args.warmup)
    d_sol, d_fit, d_suc, d_steps = bufs[0]
    identical = all(bool(torch.equal(o[0], d_sol)) and bool(torch.equal(o[2], d_suc)) and bool(torch.equal(o[3], d_steps)) for o in bufs[1:])
    sequential = None
    if nfl > 1 and not args.timed_only:  # the same steps strictly one after the other, for the record
        nfl_saved, nfl = nfl, 1
        sel, skm = timed(min(args.steps, 10), 1)
        nfl = nfl_saved
        sequential = (sel / max(min(args.steps, 10), 1), skm)

    suc = d_suc.cpu().numpy()
    steps_q = d_steps.cpu().numpy()
    n_success = int(suc.sum())
    total_success = float(n_success * args.steps)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        ts = torch.tensor([total_success], dtype=torch.float64, device=dev)
        dist.all_reduce(ts, op=dist.ReduceOp.SUM)
        total_success = float(ts.item())

    # result-level check of what was timed: every success reproduces its goal pose under the device's own exact FK
    sol = d_sol.cpu().numpy()
    ok_idx = np.nonzero(suc)[0][:256]
    tips = np.stack([h.fk_genes(sol[i], sol[i][h.active_variables][None, :])[0] for i in ok_idx]) if len(ok_idx) else np.zeros((0, T, 7))
    pos_err = float(np.linalg.norm(tips[:, 0, :3] - params[ok_idx, :3], axis=1).max()) if len(ok_idx) else 0.0
    rot_err = float((2.0 * np.arccos(np.minimum(1.0, np.abs(np.einsum("ij,ij->i", tips[:, 0, 3:], params[ok_idx, 3:7]))))).max()) if len(ok_idx) else 0.0

    # algorithmic bytes of one launch
    gens_per_step = 2 * (8 if p.mode != abi.MODE_BIO2 else 16)
    b_gen = 8 * (POP * (3 * D + 1) + 8 * D)
    generations = float(steps_q.astype(np.float64).sum()) * gens_per_step
    alg_bytes = generations * b_gen + BATCH * (8 * (7 * T + V) + 8 * (V + 3))
    achieved = alg_bytes / (kernel_ms * 1e-3) if kernel_ms > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("k_solve_hbm_bytes_per_launch")
        except Exception:
            traffic = None

    out = {
        "metric": "IK solves/sec (pop=128, 7-DOF PoseGoal batch)",
        "value": total_success / elapsed if elapsed > 0 else 0.0,
        "unit": "solves/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "PR2-like right_arm 7-DOF, batch of 4096 independent PoseGoals per GPU, bio2_memetic pop=128, exact FK per individual",
                   "batch_per_gpu": BATCH, "population": POP, "max_steps": MAX_STEPS, "dtwist": 1e-5, "sharding": "queries split across ranks, no collective",
                   "batches_in_flight": nfl},
        "success_rate": float(suc.mean()),
        "mean_steps_per_solve": float(steps_q.mean()),
        "child_evaluations_per_s": generations * POP * args.steps * world / elapsed if elapsed > 0 else 0.0,  # fitness evaluations of children (rank 0's count x ranks)
        "max_pos_err_m_of_successes": pos_err,
        "max_rot_err_rad_of_successes": rot_err,
        "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": achieved / HBM_PEAK, "traffic": traffic,
                     "kernel": "k_solve_lean", "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": alg_bytes,
                     "note": "population is LDS-resident: measured HBM traffic << algorithmic bytes; kernel is FP64-VALU bound (DESIGN.md §6); "
                             "kernel_ms is the event-bracketed duration of one launch while %d launches share the chip" % nfl,
                     "chip_level_achieved": alg_bytes * args.steps / elapsed / 1e9 if elapsed > 0 else 0.0},
        "results_identical_across_streams": identical,
    }
    if sequential is not None and world == 1:
        out["one_batch_at_a_time"] = {"value": n_success / sequential[0], "unit": "solves/s", "ms_per_step": sequential[0] * 1e3, "kernel_ms": sequential[1],
                                      "roofline_frac": alg_bytes / (sequential[1] * 1e-3) / HBM_PEAK if sequential[1] > 0 else 0.0}

    # Secondary measurement (north-star layout): the population genotype array resident in HBM, genes [unit][D][pop]
    # (individual index fastest), one launch = exact-FK fitness of every individual.  Reported next to the solver line;
    # it is not what `value` measures.
    if rank == 0 and world == 1 and os.environ.get("BIOIK_BENCH_STREAM", "1") != "0" and not args.timed_only:
        units = int(os.environ.get("BIOIK_BENCH_STREAM_UNITS", "16384"))
        stream = torch.cuda.current_stream(dev)
        g = torch.rand((units, D, POP), dtype=torch.float64, device=dev) * 2.0 - 1.0
        f = torch.empty((units, POP), dtype=torch.float64, device=dev)
        us = d_seeds[torch.arange(units, device=dev) % BATCH].contiguous()
        up = d_params[torch.arange(units, device=dev) % BATCH].contiguous()
        for _ in range(2):
            h.stream_fitness_device(units, POP, us.data_ptr(), up.data_ptr(), g.data_ptr(), f.data_ptr(), stream.cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record(stream)
        for _ in range(reps):
            h.stream_fitness_device(units, POP, us.data_ptr(), up.data_ptr(), g.data_ptr(), f.data_ptr(), stream.cuda_stream)
        e1.record(stream)
        torch.cuda.synchronize(dev)
        ms = e0.elapsed_time(e1) / reps
        sbytes = units * POP * 8 * (D + 1) + units * 8 * (V + h.P)
        out["streamed_fitness"] = {"kernel": "k_stream_fitness", "individuals_per_launch": units * POP, "ms": ms,
                                   "evaluations_per_s": units * POP / (ms * 1e-3), "algorithmic_bytes_per_launch": sbytes,
                                   "achieved_GBps": sbytes / (ms * 1e-3) / 1e9, "frac_of_hbm_peak": sbytes / (ms * 1e-3) / HBM_PEAK,
                                   "layout": "genes [unit][D][pop] f64 in HBM, 512-byte segments per wavefront load"}

    if rank == 0 and world == 1 and not args.timed_only:
        # the host-pointer entry point (what the plugin calls): staging into page-locked memory, one DMA each way, the launch
        ts = []
        for _ in range(3):
            t1 = time.perf_counter()
            hs = h.solve_batch(p, seeds, params)
            ts.append(time.perf_counter() - t1)
        out["host_pointer_entry"] = {"ms_per_call": min(ts) * 1e3, "solves_per_s": float(hs[2].sum()) / min(ts),
                                     "results_identical_to_device_entry": bool(np.array_equal(hs[0], sol) and np.array_equal(hs[2], suc)),
                                     "note": "bioik_solve_batch: host arrays in and out (PCIe-inclusive), one launch at a time; never `value`"}

    if rank == 0 and world == 1 and not args.timed_only:
        # The same queries at the REFERENCE'S OWN parameters (2 species x (2 elites + 16 children), linearised phenotypes, budget
        # 512 steps as in the cpu_baseline leg): the like-for-like figure next to cpu_baseline.value; never `value`.
        pr = abi.default_solve_params(population=16, max_steps=512, random_seed=1, fk_mode=abi.FK_LINEAR)
        for i in range(2):
            o = bufs[i % nfl]
            h.solve_batch_device(pr, BATCH, d_seeds.data_ptr(), d_params.data_ptr(), o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), streams[i % nfl].cuda_stream)
        torch.cuda.synchronize(dev)
        kr = 8
        t1 = time.perf_counter()
        for i in range(kr):
            o = bufs[i % nfl]
            h.solve_batch_device(pr, BATCH, d_seeds.data_ptr(), d_params.data_ptr(), o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), streams[i % nfl].cuda_stream)
        torch.cuda.synchronize(dev)
        dtr = (time.perf_counter() - t1) / kr
        rsuc = bufs[0][2].cpu().numpy()
        out["reference_parameters"] = {"value": float(rsuc.sum()) / dtr, "unit": "solves/s", "ms_per_step": dtr * 1e3, "success_rate": float(rsuc.mean()),
                                       "mean_steps_per_solve": float(bufs[0][3].cpu().numpy().mean()), "population": 16, "fk": "linear", "max_steps": 512,
                                       "batches_in_flight": nfl}

    if rank == 0 and world == 1 and not args.timed_only:
        # The "tracking" workload of SURVEY.md section 8(d) (seed = target + N(0, 0.1 rad), as the reference's ik_test does): same
        # template, same parameters, same issue pattern; reported next to the global-seed figure, never `value`.
        tseeds, tparams, _ = make_queries(template, h.active_variables, h.fk_genes, BATCH, seed=0xB101C + rank, kind="tracking")
        dts, dtp = torch.from_numpy(tseeds).to(dev), torch.from_numpy(tparams).to(dev)

        def tstep(i):
            o = bufs[i % nfl]
            h.solve_batch_device(p, BATCH, dts.data_ptr(), dtp.data_ptr(), o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), streams[i % nfl].cuda_stream)
        for i in range(nfl):
            tstep(i)
        torch.cuda.synchronize(dev)
        kt = 30
        t1 = time.perf_counter()
        for i in range(kt):
            tstep(i)
        torch.cuda.synchronize(dev)
        dtt = (time.perf_counter() - t1) / kt
        tsuc = bufs[0][2].cpu().numpy()
        out["tracking_seeds"] = {"value": float(tsuc.sum()) / dtt, "unit": "solves/s", "ms_per_step": dtt * 1e3, "success_rate": float(tsuc.mean()),
                                 "mean_steps_per_solve": float(bufs[0][3].cpu().numpy().mean()), "batches_in_flight": nfl,
                                 "sample": "4096 queries, seed = target configuration + N(0, 0.1 rad) clipped to the joint limits"}

    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.timed_only:
        from oracle import orc, ref
        ns = min(args.cpu_sample, BATCH)
        # (1) the oracle (a port of the reference algorithm, reference Release flags, libm) with the SAME parameters as the
        #     GPU run: 128 children per species, exact FK per individual, same step budget
        o = orc.Oracle(template, kind="ref")
        t1 = time.perf_counter()
        _, _, osuc, osteps = o.solve_batch(p, orc.RNG_COUNTER, seeds[:ns], params[:ns], n_threads=1)
        dt = time.perf_counter() - t1
        port = {"value": float(osuc.sum()) / dt, "unit": "solves/s", "cores": 1, "kind": "port", "success_rate": float(osuc.mean()), "seconds": dt,
                "sample": "first %d of the same 4096 queries, same parameters as the GPU run (pop=%d, exact FK, max_steps=%d), 1 thread" % (ns, POP, MAX_STEPS)}
        ncpu = os.cpu_count() or 1
        t1 = time.perf_counter()
        _, _, asuc, _ = o.solve_batch(p, orc.RNG_COUNTER, seeds, params, n_threads=ncpu)
        dt = time.perf_counter() - t1
        port["all_cores"] = {"value": float(asuc.sum()) / dt, "cores": ncpu, "seconds": dt, "sample": "all 4096 queries, query-parallel"}
        if ref.release_available():
            # (2) the REFERENCE'S OWN code (oracle/_ref: src/ik_evolution_2.cpp etc. compiled unmodified with the reference's
            #     Release flags): bio2_memetic as shipped — 2 species x (2 elites + 16 children), linearised FK, its own RNG,
            #     one island = one thread; budget form of the island loop (success test after every step, <= 512 steps)
            pr = abi.default_solve_params(mode="bio2_memetic", random_seed=1)
            r = ref.Reference(template, pr, release=True)
            r.solve_batch(seeds[:4], params[:4], 512)  # constructs the solver (fills its 2 x 64 MiB random tables) outside the timing
            dts = []
            for _ in range(5):  # the whole batch, five times: median rate (one pass is < 1 s of CPU time)
                t1 = time.perf_counter()
                _, _, rsuc, rsteps = r.solve_batch(seeds, params, 512)
                dts.append(time.perf_counter() - t1)
            dt = float(np.median(dts))
            cb = {"value": float(rsuc.sum()) / dt, "unit": "solves/s", "cores": 1, "kind": "reference", "success_rate": float(rsuc.mean()), "seconds": float(sum(dts)),
                  "mean_steps": float(rsteps.mean()), "host_cpus": ncpu, "rate_min_max": [float(rsuc.sum()) / max(dts), float(rsuc.sum()) / min(dts)],
                  "sample": "the same 4096 queries, five passes, median pass; the reference's own bio2_memetic (oracle/_ref, reference Release flags), its "
                            "hard-coded population (2 species x (2+16)), linearised FK, 1 island, success test after every step, <= 512 steps, 1 thread",
                  "port_same_parameters": port}
        else:
            cb = dict(port)
            cb["host_cpus"] = ncpu
        out["cpu_baseline"] = cb
        out["speedup_vs_cpu_1thread"] = out["value"] / cb["value"] if cb["value"] > 0 else None
        out["speedup_vs_port_same_parameters_1thread"] = out["value"] / port["value"] if port["value"] > 0 else None
        if "reference_parameters" in out and cb.get("kind") == "reference" and cb["value"] > 0:
            out["speedup_at_reference_parameters_vs_cpu_1thread"] = out["reference_parameters"]["value"] / cb["value"]

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
