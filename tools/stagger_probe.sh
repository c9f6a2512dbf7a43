for TP in 0 1; do for ST in 0 1 2 3 5; do
BIOIK_SOLVE_TWO_PHASE=$TP BIOIK_BENCH_STAGGER_MS=$ST python bench.py --no-cpu-baseline --timed-only --steps 60 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('two_phase=$TP stagger=$ST ms: %.0f solves/s %.2f ms per batch kernel_ms %.2f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"
done; done
