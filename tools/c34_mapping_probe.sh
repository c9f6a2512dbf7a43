run() { python bench.py --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1: configs', {k:(round(v['value']),round(v['ms_per_step'],1)) for k,v in d['configs'].items()})"; }
run auto
BIOIK_SOLVE_THREADS=64 BIOIK_SOLVE_COLUMNLESS=2 run t64_cl2
BIOIK_SOLVE_THREADS=64 BIOIK_SOLVE_COLUMNLESS=1 run t64_cl1
BIOIK_SOLVE_THREADS=128 BIOIK_SOLVE_COLUMNLESS=2 run t128_cl2
BIOIK_SOLVE_THREADS=128 BIOIK_SOLVE_COLUMNLESS=1 run t128_cl1
BIOIK_SOLVE_THREADS=64 BIOIK_SOLVE_SPECIES_PARALLEL=1 BIOIK_SOLVE_COLUMNLESS=2 run t64_sp_cl2
BIOIK_SOLVE_THREADS=256 BIOIK_SOLVE_COLUMNLESS=1 run t256_cl1
BIOIK_SOLVE_THREADS=128 run t128_columns
