real() { python bench.py --no-cpu-baseline --steps 36 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1: %.0f solves/s %.2f ms | one at a time %.0f %.2f ms | tracking %.0f' % (d['value'], d['ms_per_step'], d['one_batch_at_a_time']['value'], d['one_batch_at_a_time']['ms_per_step'], d['tracking_seeds']['value']))"; }
real default
BIOIK_SOLVE_TWO_PHASE=0 real one_launch
BIOIK_SOLVE_THREADS=64 BIOIK_SOLVE_SPECIES_PARALLEL=1 BIOIK_SOLVE_COLUMNLESS=2 real halves_whole_solve
BIOIK_SOLVE_THREADS=64 BIOIK_SOLVE_SPECIES_PARALLEL=1 BIOIK_SOLVE_COLUMNLESS=2 BIOIK_SOLVE_TWO_PHASE=1 real halves_whole_solve_split1
BIOIK_SOLVE_TWO_PHASE=8 real handover8
BIOIK_SOLVE_TWO_PHASE=16 real handover16
