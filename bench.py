#!/usr/bin/env python
"""bench.py — IK solves/sec of the MI355X bio2_memetic hot path (BASELINE.json metric).

A "step" is one pass of the hot path over one batch: `bioik_solve_batch_device` on 4096 independent PR2-like
right-arm 7-DOF PoseGoal queries (BASELINE.json configs[1]: pop=128, 1xMI355X), inputs already resident in HBM.
`value` = successful solves of all ranks / wall time of the K timed steps (max over ranks).  N>1: one process per
GPU (torch.distributed, backend nccl = RCCL), every rank solves its own 4096-query shard (weak scaling, no data-path
collective: queries are independent; RCCL only carries the barrier and the two scalar reductions of the timing).
Consecutive steps are issued round-robin on six to ten HIP streams (`--in-flight`, config.batches_in_flight: by default the largest of ten, eight, six,
five, four that divides the number of timed steps; twenty would give the headline +2 % at the driver's 20 steps, but twenty-six streams in one process cost the
C3 / C4 legs 11 % and the host-pointer leg a third, whatever GPU_MAX_HW_QUEUES says: profiles/r04_streams_in_process.log) under
bioik_solve_params::schedule = BIOIK_SCHEDULE_THROUGHPUT (`--schedule`): the library's mapping for streams of batches (both species of a query on
one wavefront: 27 % more steps per ms on a full chip, a step 2.5 x as long), under which the tail of one launch — a handful of queries that
use the whole step budget, 64 sequential steps wherever they start — lasts ~16 ms and six or more solves in flight (a hardware queue each:
GPU_MAX_HW_QUEUES) keep the chip full (profiles/r03_inflight_and_schedule.log).  The same workload under BIOIK_SCHEDULE_LATENCY — the default of
the library, what an isolated call wants — is on the line as `latency_schedule_three_in_flight` (the protocol of `value` in earlier rounds)
and `one_batch_at_a_time` (strictly one solve after the other).  A step of this bench is one call of `bioik_solve_batch_device`.

Extra objects on the JSON line:
  roofline     the solve's kernels (k_solve_lean dominant) against the roof that binds them, FP64 vector arithmetic: ALGORITHMIC flops per solve
               (130 L_m + 40 R + 25 G_pose per exact-FK evaluation of an individual, times the evaluations counted by
               the device step counters) / mean launch duration measured with HIP events on the launch stream, vs the 78.6 TFLOP/s
               FP64 vector peak of MI355X.  `roofline.hbm` holds the HBM figure the same way (§8d B_gen = 8*(pop*(3D+1) + 8D)
               bytes per generation per species per query vs 8 TB/s): the fused kernel keeps the population in LDS, so those bytes
               are notional and `traffic` (measured, profiles/) is far below them.
  configs      the other single-GPU configurations of BASELINE.json (c3: PR2 `all`, two tips + MinimalDisplacement; c4: 31-DOF snake,
               pop=512 + AvoidJointLimits) at 4096 queries per launch, six timed launches per stream (`batches_timed`): solves/s, success, ms per
               batch, their own roofline figures.
  cpu_baseline the reference's own CPU code (oracle/_ref: the reference sources compiled unmodified, Release flags) when the
               prebuilt library is present, else the oracle port; timed on this host, rank 0, N=1 only, one thread, on a
               bounded sample of the same queries.  The port at the GPU run's own parameters is reported beside it.
  tracking_seeds  the same template and parameters on seeds near the target (SURVEY.md section 8(d)'s second workload).
  reference_parameters  the GPU on the same queries at the reference's own parameters (its population, its linearised
               phenotypes): the like-for-like figure next to cpu_baseline.value.
  longer_run   the timed protocol once more over 3 K steps, behind the timed region: the steady rate of a batch on a full chip and what a run costs at its ends
               (the last batches' stragglers finishing on an empty chip); never `value`.
  small_batches, small_batches_secondary_goals, one_pose_timeouts   calls that cannot fill the chip: ms per call for 1 ... 1024 queries; ONE pose per plugin call under the
               reference's timeouts (1 / 5 / 20 ms) with the reference's own wall-clock loop beside every cell.  The one-pose leg runs first (see main()).
  summary      the figures a reader looks for first, last on the line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# The HIP runtime maps a process's streams onto FOUR hardware queues unless told otherwise; the (up to) ten solves this bench keeps in flight want a
# queue each (profiles/r03_inflight_and_schedule.log), and so do the library's own six streams of the host-pointer leg, which live in the same
# process: with sixteen queues for those seventeen streams the host-pointer pipeline shared queues and ran at 7.3e5 instead of 9.0e5 (session 58 in
# the same log): twenty-four.  Must be set before the runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

BATCH = int(os.environ.get("BIOIK_BENCH_BATCH", "4096"))      # experiments only: the reported metric uses the defaults
POP = int(os.environ.get("BIOIK_BENCH_POP", "128"))
FK_MODE = os.environ.get("BIOIK_BENCH_FK", "exact")
MAX_STEPS = int(os.environ.get("BIOIK_BENCH_MAX_STEPS", "64"))
HBM_PEAK = 8.0e12
FP64_PEAK = 78.6e12  # MI355X FP64 vector (non-matrix) peak, /opt/skills/guides/MI355X_MICROARCH.md


def kernel_sources_hash():
    """sha256 over the kernel sources (bio_ik_amd/csrc/*.h, *.hip): what a profile under profiles/ is valid for"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "bio_ik_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def flops_per_evaluation(n_moving, n_revolute, n_pose_goals):
    """algorithmic flops of ONE exact-FK fitness evaluation of ONE individual"""
    return 130.0 * n_moving + 40.0 * n_revolute + 25.0 * n_pose_goals


def _timed_device_solves(h, p, n, inputs, bufs, streams, reps):
    """`inputs`: one (seeds, goal parameters) pair of device tensors per stream (one batch of queries per stream in flight)"""
    import torch
    nfl = len(streams)
    ev = []

    def step(i):
        o, (d_seeds, d_params) = bufs[i % nfl], inputs[i % len(inputs)]
        h.solve_batch_device(p, n, d_seeds.data_ptr(), d_params.data_ptr(), o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), streams[i % nfl].cuda_stream)
    for i in range(nfl):
        step(i)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for i in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(streams[i % nfl])
        step(i)
        b.record(streams[i % nfl])
        ev.append((a, b))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t1) / reps
    return dt, float(sum(a.elapsed_time(b) for a, b in ev) / len(ev))


def other_configs(dev, nfl, streams, cpu_baseline=False):
    """BASELINE.json configs[2] and configs[3] at 4096 queries per launch, the same issue pattern as the headline figure"""
    import time

    import numpy as np
    import torch

    from bio_ik_amd import AvoidJointLimitsGoal, MinimalDisplacementGoal, PoseGoal, ProblemTemplate, abi, pr2_like, snake
    from bio_ik_amd.solver import HipSolver
    from bio_ik_amd.workload import make_queries
    n = 4096
    res = {}
    for name, template, pop, max_steps, n_moving, n_rev, n_pose, reps in (
            # (c3: fifteen moving joints -- the torso and two arms of seven; the kernels walk the torso once per arm, the algorithm needs it once)
            ("c3", ProblemTemplate(pr2_like(), "all", [PoseGoal("r_wrist_roll_link"), PoseGoal("l_wrist_roll_link"), MinimalDisplacementGoal()]), 128, 128, 15, 14, 2, 6),
            ("c4", ProblemTemplate(snake(31), "snake", [PoseGoal("tip"), AvoidJointLimitsGoal()]), 512, 32, 31, 31, 1, 6)):
        h = HipSolver(template, device=dev.index)
        p = abi.default_solve_params(population=pop, max_steps=max_steps, random_seed=1)
        inputs = []
        for k in range(nfl if not os.environ.get("BIOIK_BENCH_SAME_BATCH") else 1):  # one batch of queries per stream, as for the headline figure
            seeds, params, _ = make_queries(template, h.active_variables, h.fk_genes, n, seed=0xB101C + 1000 * k)
            inputs.append((torch.from_numpy(seeds).to(dev), torch.from_numpy(params).to(dev)))
        bufs = [(torch.empty((n, h.V), dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.float64, device=dev),
                 torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev)) for _ in range(nfl)]
        reps = max(reps, nfl) * int(os.environ.get("BIOIK_BENCH_CONFIG_ROUNDS", "6"))  # timed launches: six per stream, what the headline's default K = 60 over ten streams gives (one per stream = start of the first to end of the last launch, no steady state: 5 % less; profiles/r03_c34_launches_per_stream.log)
        dt, kernel_ms = _timed_device_solves(h, p, n, inputs, bufs, streams, reps)
        share = [len(range(k, reps, nfl)) for k in range(nfl)]
        suc_mean = sum(float(o[2].double().mean().item()) * share[k] for k, o in enumerate(bufs)) / reps  # over the timed launches
        steps_sum = sum(float(o[3].double().sum().item()) * share[k] for k, o in enumerate(bufs)) / reps
        suc, steps = np.full(n, suc_mean), np.full(n, steps_sum / n)
        gens = steps.sum() * 16
        # both configurations have a secondary goal: a generation walks a random prefix of its pre-selected children, uniform on
        # 1 ... lambda - 1 (ik_evolution_2.cpp:366-378).  Counted, not estimated: the prefix lengths are a function of the counter RNG (query, step,
        # generation, species), so the steps the device reports for every query say how many children it walked (workload.preselected_children)
        from bio_ik_amd.workload import preselected_children
        cum = preselected_children(p.random_seed, 0, n, max_steps, pop)
        walked = sum(float(cum[np.arange(n), o[3].cpu().numpy().astype(np.int64)].sum()) * share[k] for k, o in enumerate(bufs)) / reps  # per launch, over the timed launches
        evaluations = walked + 4.0 * steps.sum()
        flops = evaluations * flops_per_evaluation(n_moving, n_rev, n_pose)
        b_gen = 8 * (pop * (3 * h.D + 1) + 8 * h.D)
        res[name] = {"value": float(suc.sum()) / dt, "evaluations_per_launch": evaluations, "evaluations_note": "children walked = the random prefixes of the pre-selection, counted from the device-reported steps and the counter RNG (workload.preselected_children) + four exact evaluations per step", "unit": "solves/s", "ms_per_step": dt * 1e3, "success_rate": float(suc.mean()), "mean_steps_per_solve": float(steps.mean()),
                     "batch": n, "population": pop, "max_steps": max_steps, "D": h.D, "tips": h.T, "batches_in_flight": nfl, "batches_timed": reps, "kernel_ms": kernel_ms,
                     "roofline": {"bound": "fp64_valu", "achieved": flops / (kernel_ms * 1e-3) / 1e12, "peak": FP64_PEAK / 1e12, "unit": "TFLOP/s",
                                  "frac": flops / (kernel_ms * 1e-3) / FP64_PEAK, "chip_level_frac": flops / dt / FP64_PEAK,
                                  "hbm_notional_frac": gens * b_gen / (kernel_ms * 1e-3) / HBM_PEAK}}
        if cpu_baseline:
            # the reference's own code on one host core, the first queries of the same batch (a bounded sample: it solves tens to hundreds
            # of these per second): its hard-coded population, linearised phenotypes, success test after every step, <= 512 steps
            try:
                from oracle import ref
                if ref.release_available():
                    ns = 96 if name == "c3" else 192
                    seeds0, params0, _ = make_queries(template, h.active_variables, h.fk_genes, n, seed=0xB101C)
                    r = ref.Reference(template, abi.default_solve_params(mode="bio2_memetic", random_seed=1), release=True)
                    r.solve_batch(seeds0[:2], params0[:2], 8)  # (constructs the solver outside the timing)
                    t1 = time.perf_counter()
                    _, _, rsuc, rsteps = r.solve_batch(seeds0[:ns], params0[:ns], 512)
                    dtc = time.perf_counter() - t1
                    res[name]["cpu_baseline"] = {"value": float(rsuc.sum()) / dtc, "unit": "solves/s", "cores": 1, "kind": "reference", "seconds": dtc,
                                                 "success_rate": float(rsuc.mean()), "mean_steps": float(rsteps.mean()),
                                                 "sample": "first %d queries of the batch on stream 0; the reference's own bio2_memetic (oracle/_ref, Release flags), "
                                                           "2 species x (2+16), linearised FK, <= 512 steps, 1 thread" % ns}
                    if res[name]["cpu_baseline"]["value"] > 0:
                        res[name]["speedup_vs_cpu_1thread"] = res[name]["value"] / res[name]["cpu_baseline"]["value"]
            except Exception as e:  # (the headline line must not depend on this leg)
                res[name]["cpu_baseline"] = {"error": repr(e)}
    return res


def bench_c5(args, rank, world, gpu, dev):
    """BASELINE.json configs[4]: one mixed batch of PR2 right-arm (pop=128) and 31-DOF snake (pop=512) queries held by rank 0, sorted by
    model into two homogeneous blocks, each sharded over all ranks (scatter -> bioik_solve_batch_device on its own stream -> gather).
    A step = the whole batch end to end; value = successes / wall time.  Strong scaling: the global batch is fixed."""
    import numpy as np
    import torch

    from bio_ik_amd import AvoidJointLimitsGoal, PoseGoal, ProblemTemplate, abi, pr2_like, snake
    from bio_ik_amd.batch import solve_mixed
    from bio_ik_amd.solver import HipSolver
    from bio_ik_amd.workload import make_queries
    total = int(os.environ.get("BIOIK_BENCH_C5_BATCH", "262144"))
    t2 = ProblemTemplate(pr2_like(), "right_arm", [PoseGoal("r_wrist_roll_link")])
    t4 = ProblemTemplate(snake(31), "snake", [PoseGoal("tip"), AvoidJointLimitsGoal()])
    h2, h4 = HipSolver(t2, device=gpu), HipSolver(t4, device=gpu)
    p2 = abi.default_solve_params(population=128, max_steps=64, random_seed=1)
    p4 = abi.default_solve_params(population=512, max_steps=32, random_seed=1)
    blocks = [(h2, p2, None, None), (h4, p4, None, None)]
    if rank == 0:
        s2, g2, _ = make_queries(t2, h2.active_variables, h2.fk_genes, total // 2, seed=0xB101C)
        s4, g4, _ = make_queries(t4, h4.active_variables, h4.fk_genes, total - total // 2, seed=0xB101C + 1)
        blocks = [(h2, p2, s2, g2), (h4, p4, s4, g4)]

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize(dev)
    res = None
    for _ in range(max(args.warmup, 1)):
        res = solve_mixed(blocks, device=str(dev))
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = solve_mixed(blocks, device=str(dev))
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank == 0:
        n_success = float(res[0][2].sum() + res[1][2].sum())
        out = {"metric": "IK solves/sec (mixed PR2 pop=128 / 31-DOF snake pop=512 batch, sharded end to end)", "value": n_success * args.steps / elapsed, "unit": "solves/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / max(args.steps, 1) * 1e3, "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "BASELINE.json configs[4]: %d mixed queries (half PR2-like right arm 7-DOF PoseGoal pop=128, half 31-DOF snake PoseGoal + "
                                      "AvoidJointLimits pop=512), sorted by model, every block sharded over all ranks; host arrays on rank 0 in and out" % total,
                          "global_batch": total, "sharding": "scatter -> solve -> gather (torch.distributed; RCCL with N>1), no collective between shards"},
               "success_rate": {"pr2": float(res[0][2].mean()), "snake": float(res[1][2].mean())},
               "mean_steps_per_solve": {"pr2": float(res[0][3].mean()), "snake": float(res[1][3].mean())}}
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def small_batches_secondary(dev, cpu=True):
    """small_batches for the two usual MoveIt configurations WITH a secondary goal: the 7-joint arm with a MinimalDisplacementGoal (128 children) and the 31-joint
    chain with an AvoidJointLimitsGoal (512 children, BASELINE.json configs[3]): ms per call for 1 / 16 / 256 queries, the reference's thread beside them"""
    from bio_ik_amd import AvoidJointLimitsGoal, MinimalDisplacementGoal, PoseGoal, ProblemTemplate, pr2_like, snake
    from bio_ik_amd.solver import HipSolver
    from bio_ik_amd.workload import make_queries
    out = {}
    for key, template, pop in (("arm_minimal_displacement", ProblemTemplate(pr2_like(), "right_arm", [PoseGoal("r_wrist_roll_link"), MinimalDisplacementGoal()]), 128),
                               ("snake31_avoid_joint_limits", ProblemTemplate(snake(31), "snake", [PoseGoal("tip"), AvoidJointLimitsGoal()]), 512)):
        h = HipSolver(template, device=dev.index)
        seeds, params, _ = make_queries(template, h.active_variables, h.fk_genes, 1024, seed=0x5EC0)
        out[key] = small_batches(h, template, dev, seeds, params, cpu=cpu, sizes=(1, 16, 256), pop=pop, batch=1024, ref_steps=2048)["sizes"]
    return out


def small_batches(h, template, dev, seeds, params, cpu=True, sizes=(1, 16, 64, 256, 1024), pop=None, batch=None, ref_steps=512):
    """Calls that cannot fill the chip (MoveIt's own pattern is ONE pose per call, kinematics_plugin.cpp:437-655): ms per call for n = 1 ... 1024 queries of
    the headline workload, device arrays in and out, one call at a time, under the plugin's default (islands = BIOIK_ISLANDS_AUTO: as many islands as the idle
    part of the chip carries, stopping each other) and with one island; beside them the reference's own code on one host thread and on all of them for the
    same n queries."""
    import numpy as np
    import torch
    from bio_ik_amd import abi
    out = []
    pop = POP if pop is None else pop
    batch = BATCH if batch is None else batch
    pa = abi.default_solve_params(population=pop, max_steps=MAX_STEPS, random_seed=1, islands=abi.ISLANDS_AUTO)
    p1 = abi.default_solve_params(population=pop, max_steps=MAX_STEPS, random_seed=1)
    s = torch.cuda.Stream(dev)
    r = None
    if cpu:
        try:
            from oracle import ref
            if ref.release_available():
                r = ref.Reference(template, abi.default_solve_params(mode="bio2_memetic", random_seed=1), release=True)
                r.solve_batch(seeds[:4], params[:4], ref_steps)
        except Exception:
            r = None
    for n in sizes:
        reps = 16 if n <= 256 else 4
        o = (torch.empty((n, h.V), dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.float64, device=dev),
             torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev))
        sets = [((k * n) % (batch - n + 1)) for k in range(reps)]  # windows of the batch: different queries per call
        dsets = [(torch.from_numpy(seeds[a:a + n]).to(dev), torch.from_numpy(params[a:a + n]).to(dev)) for a in sets]
        e = {"n": n}
        for key, p in (("gpu_ms", pa), ("gpu_one_island_ms", p1)):
            def call(ds, dp):
                h.solve_batch_device(p, n, ds.data_ptr(), dp.data_ptr(), o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), s.cuda_stream)
                s.synchronize()
            call(*dsets[0]), call(*dsets[0])
            ts, suc, st = [], 0.0, []
            for ds, dp in dsets:
                t1 = time.perf_counter()
                call(ds, dp)
                ts.append(time.perf_counter() - t1)
                suc += float(o[2].sum().item())
                st.append(float(o[3].max().item()))
            e[key] = 1e3 * float(np.mean(ts))
            if key == "gpu_ms":
                e["gpu_solves_per_s"] = suc / float(np.sum(ts))
                e["gpu_success_rate"] = suc / (reps * n)
                e["gpu_islands"] = max(max(4, min(16, 2048 // n)), min(64, 2048 // (4 * n))) if n <= 1024 else 1  # (bioik_compile.cpp: normalize_params)
                e["gpu_max_steps_of_a_call"] = float(np.mean(st))
        if r is not None:
            ts = []
            for a in sets[:8 if n <= 256 else 2]:
                t1 = time.perf_counter()
                r.solve_batch(seeds[a:a + n], params[a:a + n], ref_steps)
                ts.append(time.perf_counter() - t1)
            e["reference_1thread_ms"] = 1e3 * float(np.mean(ts))
        out.append(e)
    return {"sizes": out, "what": "ms per call, one call at a time, device arrays; gpu_ms: islands = BIOIK_ISLANDS_AUTO (the plugin's default); reference: oracle/_ref on ONE host thread, "
                                  "its own parameters (pop 16, linearised FK, <= 512 steps), the same queries"}


def one_pose_timeouts(dev, cpu=True, n_calls=512):
    """The drop-in claim where MoveIt lives: ONE pose per searchPositionIK call under the reference's timeouts (/root/reference/README.md:74-101: 1 ms recommended
    for 6 - 7-DOF arms, 5 ms and 20 ms in its PR2 yaml; the loop is src/ik_parallel.h:160-190).  For four problems -- the PoseGoal arm, the arm with a secondary
    MinimalDisplacementGoal, PR2 `all` (two PoseGoals + MinimalDisplacementGoal), the 31-joint snake with AvoidJointLimitsGoal -- `n_calls` single-pose calls each
    through the PLUGIN path (bio_ik_amd/cpp/bio_ik/plugin_core.h: Engine::submit / wait per pose, looped inside the plugin library: bioik_plugin_search_each) at
    the plugin's defaults (gpu_population 128, gpu_islands 0 = sized to the idle chip) with timeout = 1 / 5 / 20 ms: success rate and mean latency per call; beside
    every cell the reference's own code (oracle/_ref, Release build, ONE solver thread, its wall-clock loop with the same timeout) on a bounded sample of the same
    queries (<= 2.5 s of CPU per cell)."""
    import numpy as np

    from bio_ik_amd import AvoidJointLimitsGoal, MinimalDisplacementGoal, PoseGoal, ProblemTemplate, abi, pr2_like, snake
    from bio_ik_amd.goals import BioIKKinematicsQueryOptions
    from bio_ik_amd.plugin import BioIKKinematicsPlugin
    from bio_ik_amd.solver import HipSolver
    from bio_ik_amd.workload import make_queries
    rows = []
    for name, model, group, tips, extra in (
            ("7-DOF arm, PoseGoal", pr2_like(), "right_arm", ["r_wrist_roll_link"], []),
            ("7-DOF arm, PoseGoal + MinimalDisplacementGoal", pr2_like(), "right_arm", ["r_wrist_roll_link"], [MinimalDisplacementGoal()]),
            ("PR2 all (15 DOF), 2 PoseGoals + MinimalDisplacementGoal", pr2_like(), "all", ["r_wrist_roll_link", "l_wrist_roll_link"], [MinimalDisplacementGoal()]),
            ("31-joint snake, PoseGoal + AvoidJointLimitsGoal", snake(31), "snake", ["tip"], [AvoidJointLimitsGoal()])):
        template = ProblemTemplate(model, group, [PoseGoal(t) for t in tips] + extra)
        h = HipSolver(template, device=dev.index)
        seeds, params, _ = make_queries(template, h.active_variables, h.fk_genes, n_calls, seed=0x0E905E)
        plug = BioIKKinematicsPlugin()
        # (gpu_max_steps: the MoveIt face's default, bio_ik/plugin_core.h Settings -- the timeout is what ends a call, as in the reference)
        plug.initialize(model, group, model.link_names[0], tips, params={"random_seed": 1, "gpu_devices": [dev.index], "gpu_max_steps": 4096})
        gv = plug._group_vars
        poses = np.stack([params[:, 8 * t:8 * t + 7] for t in range(len(tips))], axis=1)  # [n][tips][7]: the PoseGoals' numbers, in the root's frame
        opts = BioIKKinematicsQueryOptions()
        opts.goals = list(extra)
        plug.searchPositionIKEach(poses[:8], seeds[:8, gv], opts, timeout=0.02)  # (handles, streams, the launcher's first calls)
        r = None
        if cpu:
            try:
                from oracle import ref
                if ref.release_available():
                    r = ref.Reference(template, abi.default_solve_params(mode="bio2_memetic", random_seed=1), release=True)
                    if not hasattr(r.L, "ref_solve_batch_timeout"):
                        r = None
                    else:
                        r.solve_batch_timeout(seeds[:4], params[:4], 0.005)
            except Exception:
                r = None
        cells = []
        for timeout in (0.001, 0.005, 0.020):
            _, ok, _, sec = plug.searchPositionIKEach(poses, seeds[:, gv], opts, timeout=timeout)
            c = {"timeout_ms": 1e3 * timeout, "gpu_success_rate": float(ok.mean()), "gpu_mean_ms": 1e3 * float(sec.mean()), "gpu_p99_ms": 1e3 * float(np.quantile(sec, 0.99))}
            if r is not None:
                m = max(16, min(n_calls, int(2.5 / timeout)))
                _, _, suc, _, rsec = r.solve_batch_timeout(seeds[:m], params[:m], timeout)
                c["reference_success_rate"], c["reference_mean_ms"], c["reference_calls"] = float(suc.mean()), 1e3 * float(rsec.mean()), m
            cells.append(c)
        rows.append({"problem": name, "calls": n_calls, "cells": cells})
        plug.close()
    return {"rows": rows, "what": "one pose per searchPositionIK call through the plugin core (plugin defaults: gpu_population 128, gpu_islands 0), timeout 1 / 5 / 20 ms: success rate, "
                                  "mean and 99th-percentile latency per call; reference = oracle/_ref, Release build, one solver thread, its own wall-clock loop (src/ik_parallel.h:160-190), "
                                  "its own parameters (16 children, linearised phenotypes), the same queries"}


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
    ap.add_argument("--config", default="c2", choices=["c2", "c5"],
                    help="c2 (default): BASELINE.json configs[1], the configuration the metric is quoted on.  c5: configs[4], a mixed PR2 / snake "
                         "batch that rank 0 holds, sorted by model and sharded over all ranks end to end (scatter -> solve -> gather, "
                         "bio_ik_amd.batch.solve_mixed); BIOIK_BENCH_C5_BATCH sets the global batch (default 262144)")
    ap.add_argument("--in-flight", type=int, default=int(os.environ.get("BIOIK_BENCH_IN_FLIGHT", "0")),
                    help="batches in flight: consecutive steps are issued round-robin on this many HIP streams (1 = strictly one after the other).  "
                         "Default (0): under the throughput schedule the first of ten / eight / six / five / four that divides the K timed steps (else six), so that "
                         "every stream solves the same number of batches (K = 20 and K = 60: ten; profiles/r03_inflight_and_schedule.log, short runs) --, three "
                         "under the latency schedule")
    ap.add_argument("--schedule", default=os.environ.get("BIOIK_BENCH_SCHEDULE", "throughput"), choices=["throughput", "latency"],
                    help="bioik_solve_params::schedule of the timed steps (include/bioik_hip.h): throughput = the mapping for streams of batches (six in "
                         "flight), latency = the mapping for isolated calls (three in flight fill the chip); the one-at-a-time leg always runs under latency")
    args = ap.parse_args()

    # `python bench.py --gpus N` by itself (no launcher around it): re-execute under torch.distributed.run, one rank per GPU of this node.  On a box
    # with fewer GPUs than ranks the ranks share the devices over gloo (RCCL refuses two ranks on one device): a dry run of the N > 1 control flow,
    # flagged as such on the JSON line.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.execvp(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus, "--master-addr", "127.0.0.1",
                                   "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])

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
    backend = os.environ.get("BIOIK_BENCH_BACKEND", "nccl" if torch.cuda.device_count() >= world else "gloo")
    gpu = local_rank % torch.cuda.device_count()
    if world > 1:
        torch.cuda.set_device(gpu)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", gpu))
        else:
            dist.init_process_group(backend=backend)
    dev = torch.device("cuda", gpu)
    torch.cuda.set_device(dev)
    group_world = dist.get_world_size() if world > 1 else 1  # read back from the process group (what RCCL / gloo actually formed)

    if args.config == "c5":
        return bench_c5(args, rank, world, gpu, dev)

    template = ProblemTemplate(pr2_like(), "right_arm", [PoseGoal("r_wrist_roll_link")])
    h = HipSolver(template, device=gpu)
    D, V, T = h.D, h.V, h.T
    # synthetic queries of the reference's own self-test recipe (README.md:410-418); every rank draws its own shard
    seeds, params, _ = make_queries(template, h.active_variables, h.fk_genes, BATCH, seed=0xB101C + rank)
    h.set_first_query(rank * BATCH)
    p = abi.default_solve_params(population=POP, max_steps=MAX_STEPS, random_seed=1, fk_mode=abi.FK_EXACT if FK_MODE == "exact" else abi.FK_LINEAR)
    p_latency = abi.default_solve_params(population=POP, max_steps=MAX_STEPS, random_seed=1, fk_mode=abi.FK_EXACT if FK_MODE == "exact" else abi.FK_LINEAR)
    p.schedule = abi.SCHEDULE_BY_NAME[args.schedule]
    if "BIOIK_BENCH_DTWIST" in os.environ:  # experiments only (e.g. 1e-300: no query ever succeeds, every workgroup runs max_steps)
        p.dtwist = p_latency.dtwist = float(os.environ["BIOIK_BENCH_DTWIST"])

    if os.environ.get("BIOIK_BENCH_ORDER"):  # experiment: the queries of a batch sorted by the steps they will need (asc | desc), known from a first solve
        first = h.solve_batch(p, seeds, params)
        order = np.argsort(first[3], kind="stable")
        if os.environ["BIOIK_BENCH_ORDER"] == "desc":
            order = order[::-1]
        seeds, params = np.ascontiguousarray(seeds[order]), np.ascontiguousarray(params[order])
    d_seeds = torch.from_numpy(seeds).to(dev)
    d_params = torch.from_numpy(params).to(dev)
    # Consecutive steps go round-robin to `in_flight` HIP streams with their own result buffers: a launch of 4096 queries
    # ends with a long tail (the few queries that use the whole step budget run ~13 ms each) during
    # which the chip is nearly empty; with more launches in flight the next batches' bulk fills it.  Every step is a complete
    # pass of the hot path over one batch and every batch's results are complete when the timed region ends.
    if args.in_flight <= 0:
        args.in_flight = 3 if args.schedule == "latency" else next((k for k in (10, 8, 6, 5, 4) if args.steps % k == 0), 6)
    nfl = max(1, args.in_flight)
    # The one-pose-per-call leg runs FIRST, while this process has no streams of its own yet: a process with two dozen hardware queues (GPU_MAX_HW_QUEUES above)
    # and more streams than that -- this one, once all its legs have run -- has its queues time-sliced by the hardware scheduler, and a lone short launch then
    # waits for its queue's turn now and then: 1 % of the calls took 8 - 10 ms and overran their timeout (profiles/r06_one_pose_queue_oversubscription.log:
    # reproduced with idle torch streams, gone with four hardware queues or fewer streams).  A MoveIt process has the plugin's streams and no others.
    one_pose = None
    if rank == 0 and world == 1 and not args.timed_only and os.environ.get("BIOIK_BENCH_ONE_POSE", "1") != "0":
        try:
            one_pose = one_pose_timeouts(dev, cpu=not args.no_cpu_baseline)
        except Exception as e:  # (a secondary leg must not take the headline with it)
            one_pose = {"error": repr(e), "rows": []}
    streams = [torch.cuda.Stream(dev) for _ in range(nfl)]
    bufs = [(torch.empty((BATCH, V), dtype=torch.float64, device=dev), torch.empty(BATCH, dtype=torch.float64, device=dev),
             torch.empty(BATCH, dtype=torch.int32, device=dev), torch.empty(BATCH, dtype=torch.int32, device=dev)) for _ in range(nfl)]
    torch.cuda.synchronize(dev)

    # Every stream solves its OWN batch of synthetic queries (same recipe, different draws).  With one batch repeated on all streams the
    # solves in flight have identical per-query durations and their long-running workgroups coincide: 6.5 instead of 6.0 ms per batch
    # for one-launch solves (profiles/r02_two_launch_sweep.log, distinct-batch block).  BIOIK_BENCH_SAME_BATCH=1 restores the repetition.
    inputs = [(d_seeds, d_params)] * nfl
    if not os.environ.get("BIOIK_BENCH_SAME_BATCH"):
        inputs = [(d_seeds, d_params)]
        for k in range(1, nfl):
            sk, pk, _ = make_queries(template, h.active_variables, h.fk_genes, BATCH, seed=0xB101C + rank + 1000 * k)
            inputs.append((torch.from_numpy(sk).to(dev), torch.from_numpy(pk).to(dev)))

    step_params = [p]  # (the legs below swap the schedule)

    def step(i):
        o, st = bufs[i % nfl], streams[i % nfl]
        ds, dp = inputs[i % nfl]
        h.solve_batch_device(step_params[0], BATCH, ds.data_ptr(), dp.data_ptr(), o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(),
                             st.cuda_stream)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    stagger_ms = float(os.environ.get("BIOIK_BENCH_STAGGER_MS", "0"))

    def open_streams():
        # set-up, not a step: the first solve on a HIP stream creates its hardware queue and the queue's scratch memory (tens of ms on some boxes:
        # profiles/r03_inflight_and_schedule.log, session 55).  With W < streams that would fall into the timed region.
        for i in range(nfl):
            step(i)
        barrier()

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
            if stagger_ms > 0.0 and i < nfl - 1:  # experiment (BIOIK_BENCH_STAGGER_MS): the first solves of the streams start this far apart
                time.sleep(stagger_ms * 1e-3)
        barrier()
        el = time.perf_counter() - t0
        return el, (float(np.mean([a.elapsed_time(b) for a, b in ev])) if ev else 0.0)

    open_streams()
    elapsed, kernel_ms = timed(args.steps, args.warmup)
    d_sol, d_fit, d_suc, d_steps = bufs[0]
    # what the timed steps produced, per stream: successes and step() calls of the batch that stream solves, times its share of the steps
    share = [len(range(k, args.steps, nfl)) for k in range(nfl)]
    succ_timed = float(sum(int(o[2].sum().item()) * share[k] for k, o in enumerate(bufs)))
    steps_timed = float(sum(float(o[3].double().sum().item()) * share[k] for k, o in enumerate(bufs)))
    # determinism across streams: batch 0 solved once more on another stream gives the same bits
    identical = True
    if nfl > 1:
        chk = (torch.empty_like(d_sol), torch.empty_like(d_fit), torch.empty_like(d_suc), torch.empty_like(d_steps))
        h.solve_batch_device(p, BATCH, d_seeds.data_ptr(), d_params.data_ptr(), chk[0].data_ptr(), chk[1].data_ptr(), chk[2].data_ptr(), chk[3].data_ptr(),
                             streams[1].cuda_stream)
        torch.cuda.synchronize(dev)
        identical = bool(torch.equal(chk[0], d_sol)) and bool(torch.equal(chk[2], d_suc)) and bool(torch.equal(chk[3], d_steps))
    # The same protocol over three times as many steps, right behind the timed region: a run of K steps is K batches at the steady rate plus a fixed piece -- the
    # chip filling up at the start and, mostly, the stragglers of the last batches finishing on a nearly empty chip at the end (up to 64 sequential steps each).
    # Two run lengths separate the two: what the kernels do on a full chip, and what the protocol's ends cost.
    longer = None
    if world == 1 and not args.timed_only and args.steps >= nfl and os.environ.get("BIOIK_BENCH_LONGER", "1") != "0":
        k3 = 3 * args.steps
        el3, _ = timed(k3, 0)
        sh3 = [len(range(k, k3, nfl)) for k in range(nfl)]
        suc3 = float(sum(int(o[2].sum().item()) * sh3[k] for k, o in enumerate(bufs)))
        longer = {"steps": k3, "value": suc3 / el3, "ms_per_step": 1e3 * el3 / k3,
                  "steady_ms_per_step": 1e3 * (el3 - elapsed) / (k3 - args.steps), "fixed_ms_per_run": 1e3 * (elapsed - (el3 - elapsed) / (k3 - args.steps) * args.steps)}
    sequential = latency3 = None
    if nfl > 1 and not args.timed_only:
        # the same steps strictly one after the other, and three in flight, under BIOIK_SCHEDULE_LATENCY: what an isolated call takes, and the
        # protocol of the rounds before the schedule existed
        nfl_saved, nfl = nfl, 1
        step_params[0] = p_latency
        sel, skm = timed(min(args.steps, 10), 1)
        sequential = (sel / max(min(args.steps, 10), 1), skm)
        if nfl_saved >= 3:
            nfl = 3
            n3 = min(args.steps, 30)
            l3, _ = timed(n3, 3)
            sh3 = [len(range(k, n3, 3)) for k in range(3)]
            latency3 = (float(sum(int(bufs[k][2].sum().item()) * sh3[k] for k in range(3))) / l3, l3 / n3)
        nfl = nfl_saved
        step_params[0] = p

    suc = d_suc.cpu().numpy()
    steps_q = d_steps.cpu().numpy()
    n_success = int(suc.sum())
    total_success = succ_timed
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        ts = torch.tensor([total_success], dtype=torch.float64, device=dev)
        dist.all_reduce(ts, op=dist.ReduceOp.SUM)
        total_success = float(ts.item())

    # result-level check of what was timed: every success reproduces its goal pose -- under the ORACLE's exact FK in its reference-pinned arithmetic
    # (mode 0: the reference's own unfused expressions + libm; the checker, outside every timed region), the device's own FK only where the oracle's
    # library is not on the box
    sol = d_sol.cpu().numpy()
    ok_idx = np.nonzero(suc)[0][:256]
    pose_check_by = "oracle FK, reference-pinned arithmetic (mode 0)"
    try:
        from oracle import orc as _orc
        _o = _orc.Oracle(template)
        if int(_orc.lib().orc_get_trig_mode()) != 0:
            _orc.set_trig_mode(0)
        tips = _o.fk(sol[ok_idx]) if len(ok_idx) else np.zeros((0, T, 7))
    except Exception as e:
        pose_check_by = "device FK (oracle unavailable: %r)" % e
        tips = np.stack([h.fk_genes(sol[i], sol[i][h.active_variables][None, :])[0] for i in ok_idx]) if len(ok_idx) else np.zeros((0, T, 7))
    pos_err = float(np.linalg.norm(tips[:, 0, :3] - params[ok_idx, :3], axis=1).max()) if len(ok_idx) else 0.0
    rot_err = float((2.0 * np.arccos(np.minimum(1.0, np.abs(np.einsum("ij,ij->i", tips[:, 0, 3:], params[ok_idx, 3:7]))))).max()) if len(ok_idx) else 0.0

    # algorithmic bytes of one launch
    gens_per_step = 2 * (8 if p.mode != abi.MODE_BIO2 else 16)
    b_gen = 8 * (POP * (3 * D + 1) + 8 * D)
    steps_per_launch = steps_timed / max(args.steps, 1)  # step() calls of one batch: the mean over the timed launches (each stream has its own batch)
    generations = steps_per_launch * gens_per_step
    alg_bytes = generations * b_gen + BATCH * (8 * (7 * T + V) + 8 * (V + 3))
    achieved = alg_bytes / (kernel_ms * 1e-3) if kernel_ms > 0 else 0.0
    # algorithmic flops of one launch: every child of every generation + the four exact evaluations of a step's species management
    n_moving, n_rev = 8, 7  # PR2-like right arm: torso (prismatic, inactive) + 7 revolute joints on the chain
    fpe = flops_per_evaluation(n_moving, n_rev, 1)
    evaluations = generations * POP + 4.0 * steps_per_launch
    alg_flops = evaluations * fpe
    # measured HBM traffic of a solve: a rocprofv3 PMC figure from profiles/ (it cannot be collected inside this process), valid only for
    # the kernel sources it was measured on -- traffic.json carries their hash, and a figure taken on other sources is not reported
    traffic, traffic_note = None, "no profiles/traffic.json"
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("kernel_sources_sha256") == kernel_sources_hash():
                traffic, traffic_note = tj.get("k_solve_hbm_bytes_per_launch"), "measured on these kernel sources (%s)" % tj.get("source")
            else:
                traffic_note = "profiles/traffic.json was measured on other kernel sources (%s): not reported" % tj.get("source")
        except Exception as e:
            traffic_note = "profiles/traffic.json unreadable: %r" % e

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
        "process_group": {"backend": backend if world > 1 else None, "world_size": group_world, "devices_visible": torch.cuda.device_count(),
                          "note": None if world <= 1 or torch.cuda.device_count() >= world else "fewer devices than ranks: the ranks share them (dry run of the N > 1 control flow, not a scaling figure)"},
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic (one batch of queries per stream in flight: same recipe, different draws)",
        "config": {"workload": "PR2-like right_arm 7-DOF, batch of 4096 independent PoseGoals per GPU, bio2_memetic pop=128, exact FK per individual",
                   "batch_per_gpu": BATCH, "population": POP, "max_steps": MAX_STEPS, "dtwist": 1e-5, "sharding": "queries split across ranks, no collective",
                   "batches_in_flight": nfl, "schedule": args.schedule, "hardware_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                   "stream_setup": "one untimed solve per stream before the W warm-up steps (creates the stream's hardware queue and its scratch memory)"},
        "success_rate": succ_timed / max(args.steps * BATCH, 1),
        "mean_steps_per_solve": steps_per_launch / BATCH,
        "child_evaluations_per_s": generations * POP * args.steps * world / elapsed if elapsed > 0 else 0.0,  # fitness evaluations of children (rank 0's count x ranks)
        "max_pos_err_m_of_successes": pos_err,
        "max_rot_err_rad_of_successes": rot_err,
        "pose_check": pose_check_by,
        "roofline": {"bound": "fp64_valu", "achieved": alg_flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0, "peak": FP64_PEAK / 1e12, "unit": "TFLOP/s",
                     "frac": alg_flops * args.steps / elapsed / FP64_PEAK if elapsed > 0 else 0.0,  # chip level: the flops of all timed solves over the timed region's wall time
                     "frac_per_solve": alg_flops / (kernel_ms * 1e-3) / FP64_PEAK if kernel_ms > 0 else 0.0, "traffic": traffic, "traffic_provenance": traffic_note,
                     "kernel": "k_solve_lean_cl64w4 + k_solve_lean_cl4h" if args.schedule == "throughput" else "k_solve_lean_cl64w4 + k_solve_lean_cl4h (latency schedule)",
                     "kernel_trace_name": "k_solve_lean_cl64w4",  # (the launch that does the bulk of a solve; its stragglers continue under k_solve_lean_cl4h when the chip runs empty)
                     "kernel_ms": kernel_ms,
                     "algorithmic_flops_per_launch": alg_flops,
                     "flops_per_evaluation": fpe, "evaluations_per_launch": evaluations,
                     "chip_level_achieved": alg_flops * args.steps / elapsed / 1e12 if elapsed > 0 else 0.0,
                     "chip_level_frac": alg_flops * args.steps / elapsed / FP64_PEAK if elapsed > 0 else 0.0,
                     "note": "FP64 vector arithmetic binds this kernel (no MFMA: chains of 3-vector / quaternion products); flops = SURVEY.md section 8(d) "
                             "formula x fitness evaluations counted on the device; under either schedule a chip-filling solve is a launch of k_solve_lean_cl64w4 (both "
                             "species of a query on one wavefront) whose stragglers continue under k_solve_lean_cl4h (k_solve_lean_cl4 with two helper wavefronts) when the chip runs empty "
                             "(round 5; the schedules differ in the threshold); kernel_ms is the event-bracketed duration of a solve -- both launches -- while %d solves "
                             "share the chip, so `frac_per_solve` is per solve and `frac` = `chip_level_frac` is all solves over the wall time; `traffic` = measured HBM "
                             "bytes per launch (rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE, profiles/)" % nfl,
                     "hbm": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": achieved / HBM_PEAK,
                             "algorithmic_bytes_per_launch": alg_bytes, "chip_level_achieved": alg_bytes * args.steps / elapsed / 1e9 if elapsed > 0 else 0.0,
                             "note": "notional: SURVEY.md section 8(d) B_gen bytes per generation if the population streamed through HBM; the population is "
                                     "LDS-resident and the measured traffic is `roofline.traffic`"}},
        "results_identical_across_streams": identical,
    }
    if longer is not None:
        longer["frac"] = alg_flops * longer["steps"] / (longer["ms_per_step"] * 1e-3 * longer["steps"]) / FP64_PEAK
        longer["steady_frac"] = alg_flops / (longer["steady_ms_per_step"] * 1e-3) / FP64_PEAK if longer["steady_ms_per_step"] > 0 else None
        longer["note"] = ("the timed protocol once more over three times the steps, behind the timed region: `steady_ms_per_step` = the extra steps' extra time (the kernels on "
                          "a full chip), `fixed_ms_per_run` = what the K-step region takes beyond K steady steps (the chip filling up, and the last batches' stragglers -- up "
                          "to 64 sequential steps each -- finishing on an empty chip); never `value`")
        out["longer_run"] = longer
    if latency3 is not None and world == 1:
        out["latency_schedule_three_in_flight"] = {"value": latency3[0], "unit": "solves/s", "ms_per_step": latency3[1] * 1e3, "batches_in_flight": 3,
                                                   "note": "BIOIK_SCHEDULE_LATENCY, three solves in flight: the protocol of `value` up to round 3's first profile"}
    if sequential is not None and world == 1:
        out["one_batch_at_a_time"] = {"value": n_success / sequential[0], "unit": "solves/s", "ms_per_step": sequential[0] * 1e3, "kernel_ms": sequential[1],
                                      "schedule": "latency",
                                      "roofline_frac": alg_flops / (sequential[1] * 1e-3) / FP64_PEAK if sequential[1] > 0 else 0.0,
                                      "hbm_roofline_frac": alg_bytes / (sequential[1] * 1e-3) / HBM_PEAK if sequential[1] > 0 else 0.0}

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
                                   "fp64_TFLOPs": units * POP / (ms * 1e-3) * fpe / 1e12, "frac_of_fp64_peak": units * POP / (ms * 1e-3) * fpe / FP64_PEAK,
                                   "bound": "fp64_valu (the walk's arithmetic, not the 8 (D + 1) bytes per individual, limits this kernel: both roofs of SURVEY.md section 8(d) are on the line)",
                                   "layout": "genes [unit][D][pop] f64 in HBM, 512-byte segments per wavefront load"}

    if rank == 0 and world == 1 and not args.timed_only:
        # the host-pointer entry point (what the plugin calls): staging into page-locked memory, one DMA each way, the launch
        ts = []
        for _ in range(6):  # (every IO slot of the handle once, untimed: its stream and its arenas are made at first use)
            h.solve_batch(p_latency, seeds, params)
        for _ in range(3):
            t1 = time.perf_counter()
            hs = h.solve_batch(p_latency, seeds, params)  # (an isolated call: BIOIK_SCHEDULE_LATENCY)
            ts.append(time.perf_counter() - t1)
        out["host_pointer_entry"] = {"ms_per_call": min(ts) * 1e3, "solves_per_s": float(hs[2].sum()) / min(ts),
                                     "results_identical_to_device_entry": bool(np.array_equal(hs[0], sol) and np.array_equal(hs[2], suc)),
                                     "note": "bioik_solve_batch: host arrays in and out (PCIe-inclusive), one launch at a time; never `value`"}

    if rank == 0 and world == 1 and not args.timed_only:
        # a stream of batches through the host-pointer boundary WITHOUT waiting for each: bioik_solve_batch_submit / _wait (what
        # searchPositionIKBatchAsync of the plugin calls), three solves of the handle in flight, PCIe transfers included
        host_in = [(seeds, params)] + [tuple(x.cpu().numpy() for x in inputs[k]) for k in range(1, len(inputs))]
        kp = 24
        pending = []
        got_success = 0.0
        npipe = 6 if args.schedule == "throughput" else 3
        for i in range(npipe):
            h.wait_batch(h.submit_batch(p, *host_in[i % len(host_in)]))
        t1 = time.perf_counter()
        for i in range(kp):
            pending.append(h.submit_batch(p, *host_in[i % len(host_in)]))
            if len(pending) == npipe:
                got_success += float(h.wait_batch(pending.pop(0))[2].sum())
        while pending:
            got_success += float(h.wait_batch(pending.pop(0))[2].sum())
        dtp = (time.perf_counter() - t1) / kp
        out["host_pointer_pipelined"] = {"value": got_success / kp / dtp, "unit": "solves/s", "ms_per_step": dtp * 1e3, "batches_in_flight": npipe,
                                         "schedule": args.schedule,
                                         "note": "bioik_solve_batch_submit / bioik_solve_batch_wait: host arrays in and out (page-locked staging, PCIe-inclusive), "
                                                 "the solves of one handle in flight on the library's own streams; the rate a C++ caller of "
                                                 "searchPositionIKBatchAsync gets; never `value`"}

    if rank == 0 and world == 1 and not args.timed_only:
        # The same queries at the REFERENCE'S OWN parameters (2 species x (2 elites + 16 children), linearised phenotypes, budget
        # 512 steps as in the cpu_baseline leg): the like-for-like figure next to cpu_baseline.value; never `value`.
        pr = abi.default_solve_params(population=16, max_steps=512, random_seed=1, fk_mode=abi.FK_LINEAR)
        def rstep(i):
            o, (ds, dp) = bufs[i % nfl], inputs[i % nfl]
            h.solve_batch_device(pr, BATCH, ds.data_ptr(), dp.data_ptr(), o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), streams[i % nfl].cuda_stream)
        for i in range(nfl):
            rstep(i)
        torch.cuda.synchronize(dev)
        kr = max(3 * nfl, args.steps)  # (as many timed launches as the headline: a 512-step budget makes the last stragglers of every launch a 40 ms tail that few launches cannot amortise)
        t1 = time.perf_counter()
        for i in range(kr):
            rstep(i)
        torch.cuda.synchronize(dev)
        dtr = (time.perf_counter() - t1) / kr
        rsuc = bufs[0][2].cpu().numpy()
        out["reference_parameters"] = {"value": float(rsuc.sum()) / dtr, "unit": "solves/s", "ms_per_step": dtr * 1e3, "success_rate": float(rsuc.mean()),
                                       "mean_steps_per_solve": float(bufs[0][3].cpu().numpy().mean()), "population": 16, "fk": "linear", "max_steps": 512,
                                       "batches_in_flight": nfl, "batches_timed": kr}

    if rank == 0 and world == 1 and not args.timed_only:
        # The "tracking" workload of SURVEY.md section 8(d) (seed = target + N(0, 0.1 rad), as the reference's ik_test does): same
        # template, same parameters, same issue pattern; reported next to the global-seed figure, never `value`.
        tin = []
        for k in range(nfl if not os.environ.get("BIOIK_BENCH_SAME_BATCH") else 1):  # one batch per stream, as above
            tseeds, tparams, _ = make_queries(template, h.active_variables, h.fk_genes, BATCH, seed=0xB101C + rank + 1000 * k, kind="tracking")
            tin.append((torch.from_numpy(tseeds).to(dev), torch.from_numpy(tparams).to(dev)))

        def tstep(i):
            o, (dts, dtp) = bufs[i % nfl], tin[i % len(tin)]
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
                                 "sample": "4096 queries per stream, seed = target configuration + N(0, 0.1 rad) clipped to the joint limits"}

    if rank == 0 and world == 1 and not args.timed_only and os.environ.get("BIOIK_BENCH_CONFIGS", "1") != "0":
        cfg_nfl = min(nfl, 10)  # (ten in flight: their steps take 50 ms each, and six launches per stream are timed)
        out["configs"] = other_configs(dev, cfg_nfl, streams[:cfg_nfl], cpu_baseline=not args.no_cpu_baseline)

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
            # the reference's default of four island threads per query (ik_evolution_2.cpp:649) runs four CLONES of this island -- same
            # seed, same random tables, same trajectory (src/ik_parallel.h:141-145) -- so its solves/s per query stream is this one-thread
            # figure; what more host cores buy is more queries at once: the same code, one solver object per thread, the queries split over ALL
            # host threads, every thread repeating its share until about two seconds of work are done
            def query_parallel(flavour, nt, seconds=2.0):
                import threading
                refs = [ref.Reference(template, pr, release=flavour) for _ in range(nt)]
                bounds = [(i * BATCH) // nt for i in range(nt + 1)]
                got = [0.0] * nt

                def run(reps):
                    def work(i):
                        lo, hi = bounds[i], bounds[i + 1]
                        for _ in range(reps if hi > lo else 0):
                            got[i] += float(refs[i].solve_batch(seeds[lo:hi], params[lo:hi], 512)[2].sum())
                    t1 = time.perf_counter()
                    th = [threading.Thread(target=work, args=(i,)) for i in range(nt)]
                    [t.start() for t in th]
                    [t.join() for t in th]
                    return time.perf_counter() - t1
                for i in range(nt):
                    refs[i].solve_batch(seeds[:1], params[:1], 8)
                reps = max(1, min(2000, int(seconds / max(run(2) / 2.0, 1e-4))))  # (two untimed passes set the number of timed ones: the cores share caches and memory)
                for i in range(nt):
                    got[i] = 0.0
                dtp = run(reps)
                return {"value": sum(got) / dtp, "cores": nt, "seconds": dtp, "passes_over_the_batch": reps,
                        "sample": "the same 4096 queries split over %d host threads (all of them), one reference solver object each, %d passes" % (nt, reps)}
            try:
                cb["query_parallel"] = query_parallel(True, ncpu)
            except Exception as e:  # (the headline line must not depend on this leg)
                cb["query_parallel"] = {"error": repr(e)}
            # -march: the prebuilt reference library travels from the container that holds the reference tree to the box that times it, so
            # "-march=native" is not available to it; its build for AVX2 + FMA hosts (-march=x86-64-v3, oracle/Makefile) is, where the host has them
            try:
                if ref.v3_available():
                    r3 = ref.Reference(template, pr, release="v3")
                    r3.solve_batch(seeds[:4], params[:4], 512)
                    dts3 = []
                    for _ in range(3):
                        t1 = time.perf_counter()
                        _, _, rsuc3, _ = r3.solve_batch(seeds, params, 512)
                        dts3.append(time.perf_counter() - t1)
                    cb["march_x86_64_v3"] = {"value": float(rsuc3.sum()) / float(np.median(dts3)), "cores": 1, "success_rate": float(rsuc3.mean()),
                                             "flags": "the reference's Release flags + -march=x86-64-v3", "query_parallel": query_parallel("v3", ncpu)}
                else:
                    cb["march_x86_64_v3"] = {"note": "no AVX2 + FMA build of the reference library on this box, or the host lacks the flags"}
            except Exception as e:
                cb["march_x86_64_v3"] = {"error": repr(e)}
            cb["four_island_threads_note"] = ("reference default concurrency() = 4 island threads are identical clones (same RNG state): per-query "
                                              "rate = the 1-thread figure")
        else:
            cb = dict(port)
            cb["host_cpus"] = ncpu
        out["cpu_baseline"] = cb
        out["speedup_vs_cpu_1thread"] = out["value"] / cb["value"] if cb["value"] > 0 else None
        out["speedup_vs_port_same_parameters_1thread"] = out["value"] / port["value"] if port["value"] > 0 else None
        if "reference_parameters" in out and cb.get("kind") == "reference" and cb["value"] > 0:
            out["speedup_at_reference_parameters_vs_cpu_1thread"] = out["reference_parameters"]["value"] / cb["value"]

    if rank == 0 and world == 1 and not args.timed_only and os.environ.get("BIOIK_BENCH_SMALL", "1") != "0":
        out["small_batches"] = small_batches(h, template, dev, seeds, params, cpu=not args.no_cpu_baseline)
        out["small_batches_secondary_goals"] = small_batches_secondary(dev, cpu=not args.no_cpu_baseline)
    if one_pose is not None:
        out["one_pose_timeouts"] = one_pose

    if rank == 0:
        # the figures a reader looks for first, in one object at the END of the line (what a truncated tail still shows)
        c = out.get("configs", {})
        sb = {str(e["n"]): e for e in out.get("small_batches", {}).get("sizes", [])}

        def r3(x):
            return None if x is None else float("%.3g" % x)
        out["summary"] = {
            "value": r3(out["value"]), "chip_frac": r3(out["roofline"]["chip_level_frac"]),
            "steady_frac": r3(out.get("longer_run", {}).get("steady_frac")), "fixed_ms_per_run": r3(out.get("longer_run", {}).get("fixed_ms_per_run")),
            "lat3": r3(out.get("latency_schedule_three_in_flight", {}).get("value")),
            "one_at_a_time": r3(out.get("one_batch_at_a_time", {}).get("value")),
            "host_pipelined": r3(out.get("host_pointer_pipelined", {}).get("value")),
            "ref_params": r3(out.get("reference_parameters", {}).get("value")),
            "c3": [r3(c.get("c3", {}).get("value")), r3(c.get("c3", {}).get("roofline", {}).get("chip_level_frac"))],
            "c4": [r3(c.get("c4", {}).get("value")), r3(c.get("c4", {}).get("roofline", {}).get("chip_level_frac"))],
            "ms_per_call_n1_n16_n256": [r3(sb.get(k, {}).get("gpu_ms")) for k in ("1", "16", "256")],
            "ref_1thread_ms_n1_n16_n256": [r3(sb.get(k, {}).get("reference_1thread_ms")) for k in ("1", "16", "256")],
            "cpu_1thread": r3(out.get("cpu_baseline", {}).get("value")), "x_cpu": r3(out.get("speedup_vs_cpu_1thread")),
            # one pose per plugin call with timeout 1 / 5 / 20 ms, per problem (arm; arm + MinimalDisplacement; PR2 all; snake): [gpu success, gpu ms, reference success, reference ms]
            "one_pose_timeouts": [[[r3(c.get("gpu_success_rate")), r3(c.get("gpu_mean_ms")), r3(c.get("reference_success_rate")), r3(c.get("reference_mean_ms"))] for c in row["cells"]]
                                  for row in out.get("one_pose_timeouts", {}).get("rows", [])],
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
