"""Calls that cannot fill the chip: ms per call for n = 1 ... 1024 queries of the bench workload (C2: PR2-like right arm, PoseGoal, pop 128, <= 64 steps),
device arrays in and out, one call at a time, `reps` different batches per size.  Variants: environment switches of the library (lane mappings) and
islands.  usage: small_batches.py [variant ...]   variant = name:ENV=V,ENV=V;islands=I   (see main)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from bio_ik_amd import PoseGoal, ProblemTemplate, abi, pr2_like
from bio_ik_amd.solver import HipSolver
from bio_ik_amd.workload import make_queries

SIZES = [int(x) for x in os.environ.get("SMALL_SIZES", "1,16,64,256,1024").split(",")]
REPS = int(os.environ.get("SMALL_REPS", "24"))
POP, STEPS, FK = int(os.environ.get("SMALL_POP", "128")), int(os.environ.get("SMALL_STEPS", "64")), os.environ.get("SMALL_FK", "exact")  # (SMALL_POP=16 SMALL_FK=linear SMALL_STEPS=512: the reference's own parameters)


def run(h, t, name, env, islands):
    for k in [k for k in os.environ if k.startswith("BIOIK_SOLVE_")]:
        del os.environ[k]
    os.environ.update(env)
    dev = torch.device("cuda", 0)
    out = []
    for n in SIZES:
        p = abi.default_solve_params(population=POP, max_steps=STEPS, fk_mode=abi.FK_LINEAR if FK == "linear" else abi.FK_EXACT, random_seed=1, islands=islands if islands >= 0 else max(1, min(-islands, 4096 // max(n, 1))), island_sync=1 if islands not in (0, 1) else 0)  # (islands=0: BIOIK_ISLANDS_AUTO, the library's own rule)
        reps = REPS if n <= 256 else max(6, REPS // 4)
        sets = []
        for r in range(reps):
            seeds, params, _ = make_queries(t, h.active_variables, h.fk_genes, n, seed=1000 + r)
            sets.append((torch.from_numpy(seeds).to(dev), torch.from_numpy(params).to(dev)))
        o = (torch.empty((n, h.V), dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev))
        s = torch.cuda.Stream(dev)
        def call(ds, dp):
            h.solve_batch_device(p, n, ds.data_ptr(), dp.data_ptr(), o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), s.cuda_stream)
            s.synchronize()
        call(*sets[0]); call(*sets[0])
        ts, steps, suc = [], [], []
        for ds, dp in sets:
            t0 = time.perf_counter(); call(ds, dp); ts.append(time.perf_counter() - t0)
            st = o[3].cpu().numpy(); steps.append((st.mean(), st.max())); suc.append(o[2].cpu().numpy().mean())
        ms = 1e3 * np.mean(ts)
        out.append((n, ms))
        print("%-28s n %5d islands %2d (0: auto): %.3f ms per call (min %.3f max %.3f)  %.0f solves/s  steps mean %.1f, mean of max %.1f  success %.4f" % (name, n, p.islands, ms, 1e3 * min(ts), 1e3 * max(ts), n / np.mean(ts) * np.mean(suc), np.mean([a for a, b in steps]), np.mean([b for a, b in steps]), np.mean(suc)), flush=True)
    return out


def main():
    prob = os.environ.get("SMALL_PROBLEM", "c2")  # c2: the bench workload; c2sec: the same arm with a MinimalDisplacementGoal; c4: the 31-joint chain with AvoidJointLimitsGoal (SMALL_POP=512)
    if prob == "c2sec":
        from bio_ik_amd import MinimalDisplacementGoal
        t = ProblemTemplate(pr2_like(), "right_arm", [PoseGoal("r_wrist_roll_link"), MinimalDisplacementGoal()])
    elif prob == "c4":
        from bio_ik_amd import AvoidJointLimitsGoal, snake
        t = ProblemTemplate(snake(31), "snake", [PoseGoal("tip"), AvoidJointLimitsGoal()])
    else:
        t = ProblemTemplate(pr2_like(), "right_arm", [PoseGoal("r_wrist_roll_link")])
    h = HipSolver(t, device=0)
    variants = sys.argv[1:] or ["default:"]
    for v in variants:
        name, _, rest = v.partition(":")
        envs, _, isl = rest.partition(";")
        env = dict(e.split("=") for e in envs.split(",") if e)
        islands = int(isl.split("=")[1]) if isl else 1
        run(h, t, name, env, islands)


if __name__ == "__main__":
    main()
