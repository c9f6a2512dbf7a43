#!/bin/bash
# does the order of the queries inside a batch matter?  (asc / desc = sorted by the steps each query needs, known from a first solve; the
# per-query RNG streams follow the position in the batch, so a reordered batch is a different random experiment: compare the mean steps)
for TP in 0 1; do for ORD in "" asc desc; do
BIOIK_SOLVE_TWO_PHASE=$TP BIOIK_BENCH_ORDER=$ORD python bench.py --no-cpu-baseline --steps 30 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('two_phase=$TP order=${ORD:-natural}: %.0f solves/s %.2f ms per batch | one at a time %.2f ms | mean steps %.2f success %.4f' % (d['value'], d['ms_per_step'], d['one_batch_at_a_time']['ms_per_step'], d['mean_steps_per_solve'], d['success_rate']))"
done; done
