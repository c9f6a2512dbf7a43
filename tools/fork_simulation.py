"""CPU only (the checker's restatement as an analysis vehicle, nothing of the product): what FORKED islands would do to the straggler tail of the bench workload.
A query still unsolved after K steps continues as I copies of its solver state -- both species' elites, the solution -- that draw from different random streams
from then on; the first copy that passes ends the query.  (Round 4 simulated islands started late FROM THE SEED: they repeat the descent and barely help.)"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, '.')
from bio_ik_amd import PoseGoal, ProblemTemplate, abi, pr2_like
from bio_ik_amd.workload import make_queries
from oracle import orc
t = ProblemTemplate(pr2_like(), "right_arm", [PoseGoal("r_wrist_roll_link")])
o = orc.Oracle(t, kind="ref")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
budget = int(sys.argv[2]) if len(sys.argv) > 2 else 128
seeds, params, _ = make_queries(t, o.active_variables, o.fk_genes, n, seed=0xB101C)
p = abi.default_solve_params(population=128, max_steps=budget, random_seed=1)
def run(K, I):
    a, b = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
    o._chk(o.L.orc_fork_simulation(o._p, C.byref(p), C.c_size_t(n), seeds.ctypes.data_as(C.POINTER(C.c_double)), params.ctypes.data_as(C.POINTER(C.c_double)), C.c_int(K), C.c_int(I),
                                   a.ctypes.data_as(C.POINTER(C.c_int32)), b.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int(64), C.c_uint64(0)))
    return a, b
def line(name, s, extra=""):
    print("%-22s mean %.2f  p99 %3d  p99.9 %3d  max %3d  beyond 64 steps: %.4f  %s" % (name, s.mean(), np.percentile(s, 99), np.percentile(s, 99.9), s.max(), (s > 64).mean(), extra), flush=True)
a, plain = run(10**9, 1)
line("one island", plain)
for K in (8, 12, 16, 24):
    for I in (2, 4, 8):
        s, _ = run(K, I)
        work = np.where(s > K, K + (s - K) * I, s).mean() / plain.mean()
        line("fork at %d into %d" % (K, I), s, "work x%.2f" % work)
