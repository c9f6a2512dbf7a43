"""Late islands, simulated before they were built (round 4): step counts of the CPU oracle (test infrastructure used as an analysis vehicle, nothing of the product)
for four independent random streams per query of the bench workload; what a query would need if islands 1 ... I - 1 started at step K and all stopped at the first success.
usage: python tools/late_islands_simulation.py [queries]  ->  profiles/r04_late_islands_simulation.log"""
import sys, numpy as np, time
sys.path.insert(0, '/root/repo')
from bio_ik_amd import PoseGoal, ProblemTemplate, abi, pr2_like
from bio_ik_amd.workload import make_queries
from oracle import orc
t = ProblemTemplate(pr2_like(), "right_arm", [PoseGoal("r_wrist_roll_link")])
o = orc.Oracle(t, kind="ref")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
seeds, params, _ = make_queries(t, o.active_variables, o.fk_genes, n, seed=0xB101C)
p = abi.default_solve_params(population=128, max_steps=256, random_seed=1)
S = []
t0 = time.time()
# one solve per "island": islands = 1 and another random_seed each (independent streams: statistically what distinct islands of a query are)
for isl in range(4):
    p.random_seed = 1 + 1000 * isl
    sol, fit, suc, steps = o.solve_batch(p, orc.RNG_COUNTER, seeds, params, n_threads=8)
    S.append(np.where(suc > 0, steps, 10**6))
print("oracle time", time.time() - t0)
S = np.array(S)
s0 = S[0]
print("island0: mean %.2f, cdf@8,16,24,32,48,64: " % s0.mean(), [float((s0 <= k).mean()) for k in (8, 16, 24, 32, 48, 64)])
print("corr(log steps) isl0 vs isl1: %.3f" % np.corrcoef(np.log(S[0]), np.log(S[1]))[0, 1])
for I in (2, 3, 4):
    for K in (0, 4, 8, 12, 16, 24):
        late = S[1:I].min(axis=0) + K
        T = np.where(s0 <= K, s0, np.minimum(s0, late))
        work = np.where(s0 <= K, s0, K + (T - K) * I)
        print("I=%d K=%2d: mean T %.2f  max T %d  p99.9 %d  p99 %d  work/query %.2f (x%.2f)" % (I, K, T.mean(), T.max(), np.percentile(T, 99.9), np.percentile(T, 99), work.mean(), work.mean() / s0.mean()))
