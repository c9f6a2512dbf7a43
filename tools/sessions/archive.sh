#!/bin/bash
# The command lists of the GPU sessions of rounds 2 - 4 (one `gpurun` call each), kept as a record of how the figures in profiles/r0[234]_* were taken:
# one function per session, `bash tools/sessions/archive.sh <id>` runs one (ids: r02_10, r03_1 ... r03_61, r04_1 ... r04_42; no argument lists them).
# Round 5's sessions are the files tools/sessions/r05_NN.sh.

s_r02_10() {
# final session of the round: GPU suite, the round's profiles (bench line + rocprofv3 stats + PMC passes), C5 in miniature, a two-rank
# dry run of bench.py on one GPU (gloo), lone-workgroup / timeline probes are separate (tools/lone_probe.sh), randomised parity soak
O=gpurun_out; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/s10_pytest.log 2>&1; echo "pytest rc=$?" >> $O/s10_pytest.log
tail -3 $O/s10_pytest.log
bash tools/profile_round.sh r02 > $O/s10_profile_round.log 2>&1
tail -3 $O/s10_profile_round.log
python -c "
import json; d=json.load(open('$O/bench_r02.json'))
print('bench', d['value'], d['ms_per_step'], d['success_rate'], 'one-at-a-time', d['one_batch_at_a_time']['value'], 'roofline', d['roofline']['frac'], d['roofline']['chip_level_frac'])
print('configs', {k:(v['value'],v['success_rate'],v['ms_per_step'],v['roofline']['frac']) for k,v in d.get('configs',{}).items()})
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline'].get('query_parallel'), d['speedup_vs_cpu_1thread'])
"
BIOIK_BENCH_C5_BATCH=16384 python bench.py --config c5 --steps 3 --warmup 1 > $O/s10_c5.json 2> $O/s10_c5.err; head -c 600 $O/s10_c5.json; echo
BIOIK_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 1 --no-cpu-baseline > $O/s10_two_ranks.json 2> $O/s10_two_ranks.err; echo "two ranks rc=$?"; head -c 700 $O/s10_two_ranks.json; echo
bash tools/step_rate.sh bio_ik_amd/libbioik_hip.so | tee $O/s10_step_rate.log
python tools/graph_capture_check.py > $O/s10_graph.log 2>&1; python tools/graph_capture_check.py 4096 >> $O/s10_graph.log 2>&1; echo "graph capture rc=$?"; grep identical $O/s10_graph.log
timeout 900 python tools/fuzz_parity.py 300 20260926 > $O/s10_fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 $O/s10_fuzz.log

}

s_r03_1() {
# round 3, GPU session 1: GPU parity suite on the new library, then A/B of library builds (fixed-work step rate, bench line), then per-phase cycles
O=gpurun_out/s1; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gputests.log 2>&1
tail -3 $O/gputests.log
bash tools/step_rate.sh build/lib_r2.so build/lib_b.so build/lib_c_licm.so build/lib_d_nosparse.so build/lib_e_w4.so > $O/step_rate.log 2>&1
cat $O/step_rate.log
for lib in build/lib_r2.so build/lib_b.so build/lib_c_licm.so; do
  BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null > $O/bench_$(basename $lib .so).json
  python -c "import sys,json; d=json.load(open('$O/bench_$(basename $lib .so).json')); print('$lib bench: %.0f solves/s %.2f ms success %.4f one-at-a-time %.0f chip_frac %.3f | configs' % (d['value'], d['ms_per_step'], d['success_rate'], d['one_batch_at_a_time']['value'], d['roofline'].get('chip_level_frac', -1)), {k:(round(v['value']),round(v['success_rate'],3), round(v['roofline']['chip_level_frac'],3)) for k,v in d.get('configs',{}).items()})"
done 2>&1 | tee $O/bench_ab.log
for c in c2 c3 c4; do BIOIK_SOLVE_REPORT=1 BIOIK_HIP_LIBRARY=build/libphase.so python tools/phase_probe_config.py $c $([ $c = c2 ] && echo 1536 || echo 3072); done > $O/phases.log 2>&1
bash tools/lone_probe.sh build/libphase.so >> $O/phases.log 2>&1
cat $O/phases.log

}

s_r03_2() {
# round 3, GPU session 2: GPU parity suite on the new library (all of K1-K3), A/B of library builds (fixed-work step rate, bench line)
O=gpurun_out/s2; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gputests.log 2>&1
tail -3 $O/gputests.log
bash tools/step_rate.sh build/lib_b.so build/lib_f_all.so build/lib_g_k13.so build/lib_h_k2.so build/lib_i_nosink.so build/lib_j_licm.so build/lib_b.so build/lib_f_all.so > $O/step_rate.log 2>&1
cat $O/step_rate.log
for lib in build/lib_b.so build/lib_f_all.so build/lib_i_nosink.so; do
  BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null > $O/bench_$(basename $lib .so).json
  python -c "import sys,json; d=json.load(open('$O/bench_$(basename $lib .so).json')); print('$lib bench: %.0f solves/s %.2f ms success %.4f one-at-a-time %.0f chip_frac %.3f | configs' % (d['value'], d['ms_per_step'], d['success_rate'], d['one_batch_at_a_time']['value'], d['roofline'].get('chip_level_frac', -1)), {k:(round(v['value']),round(v['success_rate'],3), round(v['roofline']['chip_level_frac'],3)) for k,v in d.get('configs',{}).items()})"
done 2>&1 | tee $O/bench_ab.log

}

s_r03_3() {
# round 3, GPU session 3: GPU suite, then the round's bench line + rocprofv3 kernel stats + PMC passes (tools/profile_round.sh r03)
O=gpurun_out/s3; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gputests.log 2>&1
tail -4 $O/gputests.log
( time bash tools/profile_round.sh r03 ) > $O/profile_round.log 2>&1
tail -5 $O/profile_round.log
python -c "
import json; d=json.load(open('gpurun_out/bench_r03.json'))
print('bench: %.0f solves/s %.2f ms chip_frac %.3f one-at-a-time %.0f pipelined %.0f host %.0f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['one_batch_at_a_time']['value'], d['host_pointer_pipelined']['value'], d['host_pointer_entry']['solves_per_s']))
print('cpu', d['cpu_baseline']['value'], 'speedup', d['speedup_vs_cpu_1thread'])
print({k:(round(v['value']), round(v['success_rate'],3), round(v['roofline']['chip_level_frac'],3), v.get('cpu_baseline'), v.get('speedup_vs_cpu_1thread')) for k,v in d['configs'].items()})
print('stream', d['streamed_fitness'])
"
tail -3 gpurun_out/bench_r03.err

}

s_r03_4() {
# round 3, GPU session 4: mapping (A) vs (B) micro-benchmark; solves in flight x hardware queues; where the measured HBM bytes of a solve come from
O=gpurun_out/s4; mkdir -p $O
export TMPDIR=/tmp
./build/mapping_b > $O/mapping_b.log 2>&1; cat $O/mapping_b.log
for q in 4 8; do for f in 3 4 5 6; do
  GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline --timed-only --steps 36 --warmup 6 --in-flight $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hw queues $q in flight $f: %.0f solves/s %.2f ms per batch, per-solve kernels %.2f ms' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"
done; done 2>&1 | tee $O/inflight_hwq.log
cd /tmp
pmc() { d=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$d -o $d -- python $GRAFT_REPO_ROOT/bench.py --timed-only --steps 4 --warmup 1 --in-flight 1 > $GRAFT_REPO_ROOT/$O/pmc_$d.log 2>&1; }
BIOIK_SOLVE_TWO_PHASE=0 pmc one_f FETCH_SIZE
BIOIK_SOLVE_TWO_PHASE=0 pmc one_w WRITE_SIZE
BIOIK_SOLVE_TWO_PHASE=0 BIOIK_BENCH_MAX_STEPS=4 pmc one4_f FETCH_SIZE
BIOIK_SOLVE_TWO_PHASE=0 BIOIK_BENCH_MAX_STEPS=4 pmc one4_w WRITE_SIZE
BIOIK_SOLVE_TWO_PHASE=0 BIOIK_BENCH_BATCH=1024 pmc one1k_f FETCH_SIZE
BIOIK_SOLVE_TWO_PHASE=0 BIOIK_BENCH_BATCH=1024 pmc one1k_w WRITE_SIZE
cd $GRAFT_REPO_ROOT
python - <<'PY' | tee gpurun_out/s4/hbm_sources.log
import csv, glob, collections
for d in ("one_f","one_w","one4_f","one4_w","one1k_f","one1k_w"):
    for f in glob.glob("gpurun_out/s4/pmc_%s/*counter_collection.csv" % d):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_solve" in r["Kernel_Name"]: agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in agg.items(): print(d, k, "launches %d mean %.0f KiB" % (len(v), sum(v) / len(v)))
PY

}

s_r03_5() {
# round 3, GPU session 5: cooperative single-individual walks (COOP) A/B, GPU suite, bench, per-phase cycles, PMC traffic one solve at a time
O=gpurun_out/s5; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gputests.log 2>&1
tail -3 $O/gputests.log
ROUNDS=2 bash tools/step_rate.sh build/lib_f_all.so build/lib_k_nocoop.so build/lib_k_coop.so > $O/step_rate.log 2>&1
cat $O/step_rate.log
for lib in build/lib_k_nocoop.so build/lib_k_coop.so; do
  BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null > $O/bench_$(basename $lib .so).json
  python -c "import sys,json; d=json.load(open('$O/bench_$(basename $lib .so).json')); print('$lib bench: %.0f solves/s %.2f ms success %.4f one-at-a-time %.0f chip_frac %.3f pipelined %.0f | configs' % (d['value'], d['ms_per_step'], d['success_rate'], d['one_batch_at_a_time']['value'], d['roofline'].get('chip_level_frac', -1), d['host_pointer_pipelined']['value']), {k:(round(v['value']),round(v['success_rate'],3), round(v['roofline']['chip_level_frac'],3)) for k,v in d.get('configs',{}).items()})"
done 2>&1 | tee $O/bench_ab.log
for c in c2 c3 c4; do BIOIK_SOLVE_REPORT=1 BIOIK_HIP_LIBRARY=build/libphase.so python tools/phase_probe_config.py $c $([ $c = c2 ] && echo 1536 || echo 3072); done > $O/phases.log 2>&1
bash tools/lone_probe.sh build/libphase.so >> $O/phases.log 2>&1
grep -E "^==|fitness|reproduce|rank|approx|species|support_eval|top2" $O/phases.log
cd /tmp
pmc() { d=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$d -o $d -- python $GRAFT_REPO_ROOT/bench.py --timed-only --steps 4 --warmup 1 --in-flight 1 > $GRAFT_REPO_ROOT/$O/pmc_$d.log 2>&1; }
pmc two_f FETCH_SIZE
pmc two_w WRITE_SIZE
cd $GRAFT_REPO_ROOT
python - <<'PY' | tee gpurun_out/s5/hbm_two_launch.log
import csv, glob, collections
for d in ("two_f","two_w"):
    for f in glob.glob("gpurun_out/s5/pmc_%s/*counter_collection.csv" % d):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_solve" in r["Kernel_Name"]: agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in agg.items(): print(d, k, "launches %d mean %.0f KiB" % (len(v), sum(v) / len(v)))
PY

}

s_r03_6() {
# round 3, GPU session 6: four wavefronts per SIMD (128 VGPRs) now that the hot loops fit; one- vs two-launch solve on the new kernels
O=gpurun_out/s6; mkdir -p $O
export TMPDIR=/tmp
ROUNDS=2 bash tools/step_rate.sh build/lib_k_coop.so build/lib_l_w4.so build/lib_l_w4_nocoop.so > $O/step_rate.log 2>&1
cat $O/step_rate.log
for lib in build/lib_k_coop.so build/lib_l_w4.so; do
  BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null > $O/bench_$(basename $lib .so).json
  python -c "import sys,json; d=json.load(open('$O/bench_$(basename $lib .so).json')); print('$lib bench: %.0f solves/s %.2f ms success %.4f one-at-a-time %.0f chip_frac %.3f pipelined %.0f | configs' % (d['value'], d['ms_per_step'], d['success_rate'], d['one_batch_at_a_time']['value'], d['roofline'].get('chip_level_frac', -1), d['host_pointer_pipelined']['value']), {k:(round(v['value']),round(v['success_rate'],3), round(v['roofline']['chip_level_frac'],3)) for k,v in d.get('configs',{}).items()})"
done 2>&1 | tee $O/bench_ab.log
for k in 0 1 2; do
  BIOIK_SOLVE_TWO_PHASE=$k BIOIK_BENCH_CONFIGS=0 BIOIK_BENCH_STREAM=0 python bench.py --no-cpu-baseline --steps 36 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('two_phase=$k: %.0f solves/s %.2f ms | one at a time %.0f' % (d['value'], d['ms_per_step'], d['one_batch_at_a_time']['value']))"
done 2>&1 | tee $O/two_phase.log
BIOIK_BENCH_CONFIGS=0 BIOIK_BENCH_STREAM=0 python bench.py --no-cpu-baseline --steps 36 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('two_phase=auto: %.0f solves/s %.2f ms | one at a time %.0f' % (d['value'], d['ms_per_step'], d['one_batch_at_a_time']['value']))" | tee -a $O/two_phase.log

}

s_r03_7() {
# round 3, GPU session 7: the 128-register build of the computed-children kernel where a CU holds >= 16 wavefronts (launcher rule) against the 168-register build everywhere
O=gpurun_out/s7; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gputests.log 2>&1
tail -3 $O/gputests.log
for rep in 1 2; do for lib in build/lib_k_coop.so build/lib_m_cl4.so; do
  BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null > $O/bench_$(basename $lib .so).json
  python -c "import sys,json; d=json.load(open('$O/bench_$(basename $lib .so).json')); print('$lib bench: %.0f solves/s %.2f ms success %.4f one-at-a-time %.0f chip_frac %.3f pipelined %.0f | configs' % (d['value'], d['ms_per_step'], d['success_rate'], d['one_batch_at_a_time']['value'], d['roofline'].get('chip_level_frac', -1), d['host_pointer_pipelined']['value']), {k:(round(v['value']),round(v['success_rate'],3), round(v['roofline']['chip_level_frac'],3)) for k,v in d.get('configs',{}).items()}, 'ref-params %.0f tracking %.0f' % (d['reference_parameters']['value'], d['tracking_seeds']['value']))"
done; done 2>&1 | tee $O/bench_ab.log

}

s_r03_8() {
# round 3, GPU session 8: kernels restructured so that nothing but the lane's best two is live across the chain walk: three against four wavefronts per SIMD
O=gpurun_out/s8; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gputests.log 2>&1
grep -E "passed|failed" $O/gputests.log
ROUNDS=2 bash tools/step_rate.sh build/lib_k_coop.so build/lib_n_w3.so build/lib_n_w4.so > $O/step_rate.log 2>&1
cat $O/step_rate.log
for rep in 1 2; do for lib in build/lib_k_coop.so build/lib_n_w3.so build/lib_n_w4.so; do
  BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null > $O/bench_$(basename $lib .so).json
  python -c "import sys,json; d=json.load(open('$O/bench_$(basename $lib .so).json')); print('$lib bench: %.0f solves/s %.2f ms success %.4f one-at-a-time %.0f chip_frac %.3f pipelined %.0f | configs' % (d['value'], d['ms_per_step'], d['success_rate'], d['one_batch_at_a_time']['value'], d['roofline'].get('chip_level_frac', -1), d['host_pointer_pipelined']['value']), {k:(round(v['value']),round(v['success_rate'],3), round(v['roofline']['chip_level_frac'],3)) for k,v in d.get('configs',{}).items()}, 'ref-params %.0f tracking %.0f' % (d['reference_parameters']['value'], d['tracking_seeds']['value']))"
done; done 2>&1 | tee $O/bench_ab.log

}

s_r03_9() {
# round 3, GPU session 9: only k_solve_lean (the second launch of a C2 solve) under the four-wavefront budget
O=gpurun_out/s9; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do for lib in build/lib_k_coop.so build/lib_o_w3.so build/lib_o_lean4.so; do
  BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null > $O/bench_$(basename $lib .so).json
  python -c "import sys,json; d=json.load(open('$O/bench_$(basename $lib .so).json')); print('$lib bench: %.0f solves/s %.2f ms success %.4f one-at-a-time %.0f chip_frac %.3f pipelined %.0f | configs' % (d['value'], d['ms_per_step'], d['success_rate'], d['one_batch_at_a_time']['value'], d['roofline'].get('chip_level_frac', -1), d['host_pointer_pipelined']['value']), {k:(round(v['value']),round(v['success_rate'],3), round(v['roofline']['chip_level_frac'],3)) for k,v in d.get('configs',{}).items()}, 'ref-params %.0f tracking %.0f' % (d['reference_parameters']['value'], d['tracking_seeds']['value']))"
done; done 2>&1 | tee $O/bench_ab.log
for lib in build/lib_o_w3.so build/lib_o_lean4.so; do for n in 64 512 1024 2048 4096 16384; do
  BIOIK_HIP_LIBRARY=$lib BIOIK_BENCH_BATCH=$n BIOIK_BENCH_CONFIGS=0 BIOIK_BENCH_STREAM=0 python bench.py --no-cpu-baseline --timed-only --in-flight 1 --steps 12 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib one call of $n queries: %.2f ms, %.0f solves/s' % (d['ms_per_step'], d['value']))"
done; done 2>&1 | tee $O/batch_sizes.log

}

s_r03_10() {
# round 3, GPU session 10: GPU suite, the round's bench line + rocprofv3 kernel stats + PMC passes (one solve at a time), per-phase cycles, parity soak
O=gpurun_out/s10; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gputests.log 2>&1
grep -E "passed|failed" $O/gputests.log
( time bash tools/profile_round.sh r03 ) > $O/profile_round.log 2>&1
python -c "
import json; d=json.load(open('gpurun_out/bench_r03.json'))
print('bench: %.0f solves/s %.2f ms chip_frac %.3f one-at-a-time %.0f pipelined %.0f host %.0f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['one_batch_at_a_time']['value'], d['host_pointer_pipelined']['value'], d['host_pointer_entry']['solves_per_s']))
print('cpu', d['cpu_baseline']['value'], 'speedup', d['speedup_vs_cpu_1thread'])
print({k:(round(v['value']), round(v['success_rate'],3), round(v['roofline']['chip_level_frac'],3), round(v.get('speedup_vs_cpu_1thread',0))) for k,v in d['configs'].items()})
"
( time timeout 900 python tools/fuzz_parity.py 1500 3 ) > $O/fuzz.log 2>&1
tail -3 $O/fuzz.log

}

s_r03_11() {
# round 3, GPU session 11: the gradient terms of a generation tabulated once per species (lib_q_gt) against computing them per gene and child
# (lib_q_head): fixed work, bench line with C3 / C4; the new GPU tests
O=gpurun_out/s11; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_plugin.py tests/test_gpu_parity.py -m gpu -x -q -k "python_plugin or four_wavefront or submit_wait or selection_ties" 2>&1 | tail -5 | tee $O/new_tests.log
ROUNDS=2 bash tools/step_rate.sh build/lib_q_head.so build/lib_q_gt.so 2>&1 | tee $O/step_rate.log
for rep in 1 2; do for lib in build/lib_q_head.so build/lib_q_gt.so; do
  BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null > $O/bench_$(basename $lib .so).json
  python -c "import sys,json; d=json.load(open('$O/bench_$(basename $lib .so).json')); print('$lib bench: %.0f solves/s %.2f ms success %.4f one-at-a-time %.0f chip_frac %.3f pipelined %.0f | configs' % (d['value'], d['ms_per_step'], d['success_rate'], d['one_batch_at_a_time']['value'], d['roofline'].get('chip_level_frac', -1), d['host_pointer_pipelined']['value']), {k:(round(v['value']),round(v['success_rate'],3), round(v['roofline']['chip_level_frac'],3)) for k,v in d.get('configs',{}).items()}, 'ref-params %.0f tracking %.0f' % (d['reference_parameters']['value'], d['tracking_seeds']['value']))"
done; done 2>&1 | tee $O/bench_ab.log

}

s_r03_12() {
# round 3, GPU session 12: per-word gene records (one batch of scalar loads per reproduction trip) and the two-minima top-2 for half-wavefront groups (lib_r_words) against the previous build (lib_q_head)
# fixed work, bench line with C3 / C4, then the GPU suite on the new build
O=gpurun_out/s12; mkdir -p $O
export TMPDIR=/tmp
ROUNDS=2 bash tools/step_rate.sh build/lib_q_head.so build/lib_r_words.so 2>&1 | tee $O/step_rate.log
for rep in 1 2; do for lib in build/lib_q_head.so build/lib_r_words.so; do
  BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null > $O/bench_$(basename $lib .so).json
  python -c "import sys,json; d=json.load(open('$O/bench_$(basename $lib .so).json')); print('$lib bench: %.0f solves/s %.2f ms success %.4f one-at-a-time %.0f chip_frac %.3f pipelined %.0f | configs' % (d['value'], d['ms_per_step'], d['success_rate'], d['one_batch_at_a_time']['value'], d['roofline'].get('chip_level_frac', -1), d['host_pointer_pipelined']['value']), {k:(round(v['value']),round(v['success_rate'],3), round(v['roofline']['chip_level_frac'],3)) for k,v in d.get('configs',{}).items()}, 'ref-params %.0f tracking %.0f' % (d['reference_parameters']['value'], d['tracking_seeds']['value']))"
done; done 2>&1 | tee $O/bench_ab.log
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $O/gpu_suite.log

}

s_r03_13() {
# round 3, GPU session 13: only the two-minima top-2 for half-wavefront groups (lib_s_halves) against the previous build (lib_q_head)
# fixed work, bench line with C3 / C4, then the GPU suite on the new build
O=gpurun_out/s13; mkdir -p $O
export TMPDIR=/tmp
ROUNDS=2 bash tools/step_rate.sh build/lib_q_head.so build/lib_s_halves.so 2>&1 | tee $O/step_rate.log
for rep in 1 2 3; do for lib in build/lib_q_head.so build/lib_s_halves.so; do
  BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null > $O/bench_$(basename $lib .so).json
  python -c "import sys,json; d=json.load(open('$O/bench_$(basename $lib .so).json')); print('$lib bench: %.0f solves/s %.2f ms success %.4f one-at-a-time %.0f chip_frac %.3f pipelined %.0f | configs' % (d['value'], d['ms_per_step'], d['success_rate'], d['one_batch_at_a_time']['value'], d['roofline'].get('chip_level_frac', -1), d['host_pointer_pipelined']['value']), {k:(round(v['value']),round(v['success_rate'],3), round(v['roofline']['chip_level_frac'],3)) for k,v in d.get('configs',{}).items()}, 'ref-params %.0f tracking %.0f' % (d['reference_parameters']['value'], d['tracking_seeds']['value']))"
done; done 2>&1 | tee $O/bench_ab.log

}

s_r03_14() {
# round 3, GPU session 14: GPU suite, the bench line + rocprofv3 kernel stats + PMC passes (one solve at a time) on the final kernels, parity soak with the gradient family
O=gpurun_out/s14; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gputests.log 2>&1
grep -E "passed|failed" $O/gputests.log
( time bash tools/profile_round.sh r03 ) > $O/profile_round.log 2>&1
python -c "
import json; d=json.load(open('gpurun_out/bench_r03.json'))
print('bench: %.0f solves/s %.2f ms chip_frac %.3f one-at-a-time %.0f pipelined %.0f host %.0f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['one_batch_at_a_time']['value'], d['host_pointer_pipelined']['value'], d['host_pointer_entry']['solves_per_s']))
print('cpu', d['cpu_baseline']['value'], 'speedup', d['speedup_vs_cpu_1thread'])
print({k:(round(v['value']), round(v['success_rate'],3), round(v['roofline']['chip_level_frac'],3), round(v.get('speedup_vs_cpu_1thread',0))) for k,v in d['configs'].items()})
"
( time timeout 900 python tools/fuzz_parity.py 1200 5 ) > $O/fuzz.log 2>&1
tail -3 $O/fuzz.log

}

s_r03_15() {
# round 3, GPU session 15: the pose-only form of the memetic goal evaluation (lib_t_pose) against the build before (lib_t_base)
# fixed work, bench line with C3 / C4, then the GPU suite on the new build
O=gpurun_out/s15; mkdir -p $O
export TMPDIR=/tmp
ROUNDS=2 bash tools/step_rate.sh build/lib_t_base.so build/lib_t_pose.so 2>&1 | tee $O/step_rate.log
for rep in 1 2; do for lib in build/lib_t_base.so build/lib_t_pose.so; do
  BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null > $O/bench_$(basename $lib .so).json
  python -c "import sys,json; d=json.load(open('$O/bench_$(basename $lib .so).json')); print('$lib bench: %.0f solves/s %.2f ms success %.4f one-at-a-time %.0f chip_frac %.3f pipelined %.0f | configs' % (d['value'], d['ms_per_step'], d['success_rate'], d['one_batch_at_a_time']['value'], d['roofline'].get('chip_level_frac', -1), d['host_pointer_pipelined']['value']), {k:(round(v['value']),round(v['success_rate'],3), round(v['roofline']['chip_level_frac'],3)) for k,v in d.get('configs',{}).items()}, 'ref-params %.0f tracking %.0f' % (d['reference_parameters']['value'], d['tracking_seeds']['value']))"
done; done 2>&1 | tee $O/bench_ab.log

}

s_r03_16() {
# round 3, GPU session 16: pose-only forms -- memetic goal evaluation only (lib_t_pose), plus the per-tip form in every walk and the success test's
# early exit (lib_u_pose2) -- against the build before (lib_t_base)
O=gpurun_out/s16; mkdir -p $O
export TMPDIR=/tmp
ROUNDS=2 bash tools/step_rate.sh build/lib_t_base.so build/lib_t_pose.so build/lib_u_pose2.so 2>&1 | tee $O/step_rate.log
for rep in 1 2; do for lib in build/lib_t_base.so build/lib_t_pose.so build/lib_u_pose2.so; do
  BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null > $O/bench_$(basename $lib .so).json
  python -c "import sys,json; d=json.load(open('$O/bench_$(basename $lib .so).json')); print('$lib bench: %.0f solves/s %.2f ms success %.4f one-at-a-time %.0f chip_frac %.3f pipelined %.0f | configs' % (d['value'], d['ms_per_step'], d['success_rate'], d['one_batch_at_a_time']['value'], d['roofline'].get('chip_level_frac', -1), d['host_pointer_pipelined']['value']), {k:(round(v['value']),round(v['success_rate'],3), round(v['roofline']['chip_level_frac'],3)) for k,v in d.get('configs',{}).items()}, 'ref-params %.0f tracking %.0f' % (d['reference_parameters']['value'], d['tracking_seeds']['value']))"
done; done 2>&1 | tee $O/bench_ab.log

}

s_r03_17() {
# round 3, GPU session 17: the pose-only forms without the one in the pair walk (lib_v_pose3) against with it (lib_u_pose2) and the build before (lib_t_base);
# per-phase cycles of a lone workgroup and of a full chip on the new kernels
O=gpurun_out/s17; mkdir -p $O
export TMPDIR=/tmp
ROUNDS=2 bash tools/step_rate.sh build/lib_u_pose2.so build/lib_v_pose3.so 2>&1 | tee $O/step_rate.log
for rep in 1 2; do for lib in build/lib_t_base.so build/lib_u_pose2.so build/lib_v_pose3.so; do
  BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null > $O/bench_$(basename $lib .so).json
  python -c "import sys,json; d=json.load(open('$O/bench_$(basename $lib .so).json')); print('$lib bench: %.0f solves/s %.2f ms success %.4f one-at-a-time %.0f chip_frac %.3f pipelined %.0f | configs' % (d['value'], d['ms_per_step'], d['success_rate'], d['one_batch_at_a_time']['value'], d['roofline'].get('chip_level_frac', -1), d['host_pointer_pipelined']['value']), {k:(round(v['value']),round(v['success_rate'],3), round(v['roofline']['chip_level_frac'],3)) for k,v in d.get('configs',{}).items()}, 'ref-params %.0f tracking %.0f' % (d['reference_parameters']['value'], d['tracking_seeds']['value']))"
done; done 2>&1 | tee $O/bench_ab.log
for c in c2 c3 c4; do BIOIK_SOLVE_REPORT=1 BIOIK_HIP_LIBRARY=build/libphase.so python tools/phase_probe_config.py $c $([ $c = c2 ] && echo 1536 || echo 3072); done > $O/phases.log 2>&1
bash tools/lone_probe.sh build/libphase.so >> $O/phases.log 2>&1

}

s_r03_18() {
# round 3, GPU session 18: the goals over the joint values by gene in the memetic phase (lib_w_genes) against the build before (lib_v_pose3)
# fixed work, bench line with C3 / C4, then the GPU suite on the new build
O=gpurun_out/s18; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2 3; do for lib in build/lib_v_pose3.so build/lib_w_genes.so; do
  BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null > $O/bench_$(basename $lib .so).json
  python -c "import sys,json; d=json.load(open('$O/bench_$(basename $lib .so).json')); print('$lib bench: %.0f solves/s %.2f ms success %.4f one-at-a-time %.0f chip_frac %.3f pipelined %.0f | configs' % (d['value'], d['ms_per_step'], d['success_rate'], d['one_batch_at_a_time']['value'], d['roofline'].get('chip_level_frac', -1), d['host_pointer_pipelined']['value']), {k:(round(v['value']),round(v['success_rate'],3), round(v['roofline']['chip_level_frac'],3)) for k,v in d.get('configs',{}).items()}, 'ref-params %.0f tracking %.0f' % (d['reference_parameters']['value'], d['tracking_seeds']['value']))"
done; done 2>&1 | tee $O/bench_ab.log

}

s_r03_19() {
# round 3, GPU session 19: lane mappings of C3 / C4 on the round's final kernels, with the 128-register build of the computed-children kernel forced
O=gpurun_out/s19; mkdir -p $O
export TMPDIR=/tmp
run() { python bench.py --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1: configs', {k:(round(v['value']),round(v['ms_per_step'],1)) for k,v in d['configs'].items()})"; }
{
run auto
run auto_again
BIOIK_SOLVE_FOUR_WAVES=1 run auto_w4
BIOIK_SOLVE_THREADS=128 BIOIK_SOLVE_COLUMNLESS=2 run t128_cl2
BIOIK_SOLVE_THREADS=128 BIOIK_SOLVE_COLUMNLESS=2 BIOIK_SOLVE_FOUR_WAVES=1 run t128_cl2_w4
BIOIK_SOLVE_THREADS=128 BIOIK_SOLVE_COLUMNLESS=1 run t128_cl1
BIOIK_SOLVE_THREADS=128 BIOIK_SOLVE_COLUMNLESS=1 BIOIK_SOLVE_FOUR_WAVES=1 run t128_cl1_w4
BIOIK_SOLVE_THREADS=64 BIOIK_SOLVE_COLUMNLESS=2 run t64_cl2
BIOIK_SOLVE_THREADS=64 BIOIK_SOLVE_COLUMNLESS=2 BIOIK_SOLVE_FOUR_WAVES=1 run t64_cl2_w4
BIOIK_SOLVE_THREADS=64 BIOIK_SOLVE_SPECIES_PARALLEL=1 BIOIK_SOLVE_COLUMNLESS=2 run t64_sp_cl2
BIOIK_SOLVE_THREADS=64 BIOIK_SOLVE_SPECIES_PARALLEL=1 BIOIK_SOLVE_COLUMNLESS=2 BIOIK_SOLVE_FOUR_WAVES=1 run t64_sp_cl2_w4
} 2>&1 | tee $O/mappings.log

}

s_r03_20() {
# round 3, GPU session 20: the joint walk of both species' children (k_solve_lean_clj; C3) against one half of the wavefront per species (BIOIK_SOLVE_NO_JOINT=1)
O=gpurun_out/s20; mkdir -p $O
export TMPDIR=/tmp
run() { python bench.py --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1: configs', {k:(round(v['value']),round(v['ms_per_step'],1),round(v['success_rate'],3),round(v['roofline']['chip_level_frac'],3)) for k,v in d['configs'].items()})"; }
{
for rep in 1 2 3; do
BIOIK_SOLVE_NO_JOINT=1 run halves
run joint
done
} 2>&1 | tee $O/joint.log
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/gpu_suite.log
( time timeout 900 python tools/fuzz_parity.py 600 9 ) > $O/fuzz.log 2>&1; tail -2 $O/fuzz.log

}

s_r03_21() {
# round 3, GPU session 21: the joint walk with a wavefront per species (128 lanes; C4, and C3 under that mapping) against the launcher's choices
O=gpurun_out/s21; mkdir -p $O
export TMPDIR=/tmp
run() { python bench.py --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1: configs', {k:(round(v['value']),round(v['ms_per_step'],1),round(v['success_rate'],3),round(v['roofline']['chip_level_frac'],3)) for k,v in d['configs'].items()})"; }
{
for rep in 1 2 3; do
run auto
BIOIK_SOLVE_JOINT_128=1 run auto_joint128
BIOIK_SOLVE_JOINT_128=1 BIOIK_SOLVE_THREE_WAVES=1 run auto_joint128_w3
done
BIOIK_SOLVE_THREADS=128 BIOIK_SOLVE_COLUMNLESS=2 BIOIK_SOLVE_JOINT_128=1 run t128_cl2_joint
BIOIK_SOLVE_THREADS=128 BIOIK_SOLVE_COLUMNLESS=2 BIOIK_SOLVE_JOINT_128=1 BIOIK_SOLVE_FOUR_WAVES=1 run t128_cl2_joint_w4
} 2>&1 | tee $O/joint128.log

}

s_r03_22() {
# round 3, GPU session 22: GPU suite, the bench line + rocprofv3 kernel stats + PMC passes (one solve at a time) on the final kernels, parity soak
O=gpurun_out/s22; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gputests.log 2>&1
grep -E "passed|failed" $O/gputests.log
( time bash tools/profile_round.sh r03 ) > $O/profile_round.log 2>&1
python -c "
import json; d=json.load(open('gpurun_out/bench_r03.json'))
print('bench: %.0f solves/s %.2f ms chip_frac %.3f one-at-a-time %.0f pipelined %.0f host %.0f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['one_batch_at_a_time']['value'], d['host_pointer_pipelined']['value'], d['host_pointer_entry']['solves_per_s']))
print('cpu', d['cpu_baseline']['value'], 'speedup', d['speedup_vs_cpu_1thread'])
print({k:(round(v['value']), round(v['success_rate'],3), round(v['roofline']['chip_level_frac'],3), round(v.get('speedup_vs_cpu_1thread',0))) for k,v in d['configs'].items()})
"
( time timeout 900 python tools/fuzz_parity.py 1500 11 ) > $O/fuzz.log 2>&1
tail -3 $O/fuzz.log

}

s_r03_23() {
# round 3, GPU session 23: the winner copy skipped when no child made it (lib_y_stay) against the build before (lib_x_base)
# fixed work, bench line with C3 / C4, then the GPU suite on the new build
O=gpurun_out/s23; mkdir -p $O
export TMPDIR=/tmp
ROUNDS=2 bash tools/step_rate.sh build/lib_x_base.so build/lib_y_stay.so 2>&1 | tee $O/step_rate.log
for rep in 1 2; do for lib in build/lib_x_base.so build/lib_y_stay.so; do
  BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null > $O/bench_$(basename $lib .so).json
  python -c "import sys,json; d=json.load(open('$O/bench_$(basename $lib .so).json')); print('$lib bench: %.0f solves/s %.2f ms success %.4f one-at-a-time %.0f chip_frac %.3f pipelined %.0f | configs' % (d['value'], d['ms_per_step'], d['success_rate'], d['one_batch_at_a_time']['value'], d['roofline'].get('chip_level_frac', -1), d['host_pointer_pipelined']['value']), {k:(round(v['value']),round(v['success_rate'],3), round(v['roofline']['chip_level_frac'],3)) for k,v in d.get('configs',{}).items()}, 'ref-params %.0f tracking %.0f' % (d['reference_parameters']['value'], d['tracking_seeds']['value']))"
done; done 2>&1 | tee $O/bench_ab.log

}

s_r03_24() {
# round 3, GPU session 24: C2 under the computed-children mappings on the final kernels (fixed work and bench line)
O=gpurun_out/s24; mkdir -p $O
export TMPDIR=/tmp
lib=bio_ik_amd/libbioik_hip.so
{
for r in 1 2; do
echo "== default"; bash tools/step_rate.sh $lib | grep 3072
echo "== columnless 2 (128 lanes)"; BIOIK_SOLVE_THREADS=128 BIOIK_SOLVE_COLUMNLESS=2 ROUNDS=1 bash tools/step_rate.sh $lib | grep 3072
echo "== columnless 2, four wavefronts"; BIOIK_SOLVE_THREADS=128 BIOIK_SOLVE_COLUMNLESS=2 BIOIK_SOLVE_FOUR_WAVES=1 ROUNDS=1 bash tools/step_rate.sh $lib | grep 3072
echo "== halves, columnless 2"; BIOIK_SOLVE_THREADS=64 BIOIK_SOLVE_SPECIES_PARALLEL=1 BIOIK_SOLVE_COLUMNLESS=2 ROUNDS=1 bash tools/step_rate.sh $lib | grep 3072
echo "== halves, columnless 2, four wavefronts"; BIOIK_SOLVE_THREADS=64 BIOIK_SOLVE_SPECIES_PARALLEL=1 BIOIK_SOLVE_COLUMNLESS=2 BIOIK_SOLVE_FOUR_WAVES=1 ROUNDS=1 bash tools/step_rate.sh $lib | grep 3072
done
} 2>&1 | tee $O/c2_mappings.log

}

s_r03_25() {
# round 3, GPU session 25: hand-over of the two-launch solve after K steps on the final kernels (first launch: both species on one wavefront, computed children),
# whose throughput at fixed work is now 30 % above the second launch's mapping (profiles/r03_c2_mappings.log)
O=gpurun_out/s25; mkdir -p $O
export TMPDIR=/tmp
KS="1 2 3 4 6 8 12 16 24" bash tools/two_phase_sweep.sh 2>&1 | tee $O/two_phase.log

}

s_r03_26() {
# round 3, GPU session 26: why the half-wavefront mapping's +30 % at fixed work does not show on a stream of real solves: fixed work at 4 / 8 / 16 / 32 steps
# per query, and the whole solve under that mapping with 3 and 6 solves in flight
O=gpurun_out/s26; mkdir -p $O
export TMPDIR=/tmp
H="BIOIK_SOLVE_THREADS=64 BIOIK_SOLVE_SPECIES_PARALLEL=1 BIOIK_SOLVE_COLUMNLESS=2"
fixed() { for st in 4 8 16 32; do
  v=$(env $1 BIOIK_SOLVE_TWO_PHASE=0 BIOIK_BENCH_IN_FLIGHT=1 BIOIK_BENCH_STREAM=0 BIOIK_BENCH_DTWIST=1e-300 BIOIK_BENCH_MAX_STEPS=$st BIOIK_BENCH_BATCH=${2:-3072} python bench.py --no-cpu-baseline --timed-only --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms -> %.0f steps/ms chip-wide' % (d['ms_per_step'], ${2:-3072}*$st/d['ms_per_step']))")
  echo "$3 batch=${2:-3072} steps=$st : $v"; done; }
{
fixed "X=1" 3072 default_one_launch
fixed "$H" 3072 halves
fixed "X=1" 12288 default_one_launch
fixed "$H" 12288 halves
for nf in 3 6; do
  python bench.py --no-cpu-baseline --timed-only --in-flight $nf --steps 30 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default, $nf in flight: %.0f solves/s %.2f ms' % (d['value'], d['ms_per_step']))"
  env $H python bench.py --no-cpu-baseline --timed-only --in-flight $nf --steps 30 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('halves whole solve, $nf in flight: %.0f solves/s %.2f ms' % (d['value'], d['ms_per_step']))"
done
} 2>&1 | tee $O/halves.log

}

s_r03_27() {
# round 3, GPU session 27: solves in flight x hardware queues for the default two-launch solve, a late hand-over (K=12) and the whole solve under the
# half-wavefront mapping -- does the +27 % of that mapping at fixed work need more work in flight to show on real solves?
O=gpurun_out/s27; mkdir -p $O
export TMPDIR=/tmp
H="BIOIK_SOLVE_THREADS=64 BIOIK_SOLVE_SPECIES_PARALLEL=1 BIOIK_SOLVE_COLUMNLESS=2"
one() { env $2 python bench.py --no-cpu-baseline --timed-only --in-flight $3 --steps 48 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1, $3 in flight: %.0f solves/s %.2f ms' % (d['value'], d['ms_per_step']))"; }
{
for q in 4 8; do
export GPU_MAX_HW_QUEUES=$q
echo "== GPU_MAX_HW_QUEUES=$q"
for nf in 3 6 8 12; do
  one "default (hand-over after 1)" "X=1" $nf
  one "hand-over after 12" "BIOIK_SOLVE_TWO_PHASE=12" $nf
  one "halves whole solve" "$H" $nf
done
done
} 2>&1 | tee $O/inflight.log

}

s_r03_28() {
# round 3, GPU session 28: GPU suite and the bench line with BIOIK_SCHEDULE_THROUGHPUT, six solves in flight on eight hardware queues; profile passes
O=gpurun_out/s28; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gputests.log 2>&1
grep -E "passed|failed" $O/gputests.log
( time bash tools/profile_round.sh r03 ) > $O/profile_round.log 2>&1
python -c "
import json; d=json.load(open('gpurun_out/bench_r03.json'))
print('bench: %.0f solves/s %.2f ms chip_frac %.3f | latency schedule, three in flight %.0f | one-at-a-time %.0f | pipelined %.0f host %.0f | tracking %.0f ref-params %.0f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['latency_schedule_three_in_flight']['value'], d['one_batch_at_a_time']['value'], d['host_pointer_pipelined']['value'], d['host_pointer_entry']['solves_per_s'], d['tracking_seeds']['value'], d['reference_parameters']['value']))
print('cpu', d['cpu_baseline']['value'], 'speedup', d['speedup_vs_cpu_1thread'])
print({k:(round(v['value']), round(v['success_rate'],3), round(v['roofline']['chip_level_frac'],3), round(v.get('speedup_vs_cpu_1thread',0))) for k,v in d['configs'].items()})
"
tail -3 gpurun_out/bench_r03.err
bash tools/step_rate.sh bio_ik_amd/libbioik_hip.so | tail -2
( time timeout 900 python tools/fuzz_parity.py 1500 13 ) > $O/fuzz.log 2>&1
tail -3 $O/fuzz.log

}

s_r03_29() {
# round 3, GPU session 29: GPU suite, bench line (BIOIK_SCHEDULE_THROUGHPUT, six in flight, eight hardware queues), profile passes, parity soak -- final library of the round
O=gpurun_out/s29; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gputests.log 2>&1
grep -E "passed|failed" $O/gputests.log
( time bash tools/profile_round.sh r03 ) > $O/profile_round.log 2>&1
python -c "
import json; d=json.load(open('gpurun_out/bench_r03.json'))
print('bench: %.0f solves/s %.2f ms chip_frac %.3f | latency schedule, three in flight %.0f | one-at-a-time %.0f | pipelined %.0f host %.0f | tracking %.0f ref-params %.0f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['latency_schedule_three_in_flight']['value'], d['one_batch_at_a_time']['value'], d['host_pointer_pipelined']['value'], d['host_pointer_entry']['solves_per_s'], d['tracking_seeds']['value'], d['reference_parameters']['value']))
print('cpu', d['cpu_baseline']['value'], 'speedup', d['speedup_vs_cpu_1thread'])
print({k:(round(v['value']), round(v['success_rate'],3), round(v['roofline']['chip_level_frac'],3), round(v.get('speedup_vs_cpu_1thread',0))) for k,v in d['configs'].items()})
"
tail -3 gpurun_out/bench_r03.err
bash tools/step_rate.sh bio_ik_amd/libbioik_hip.so | tail -2
( time timeout 900 python tools/fuzz_parity.py 1500 17 ) > $O/fuzz.log 2>&1
tail -3 $O/fuzz.log
python tools/pipeline_probe.py 2>&1 | grep -v Warning | tail -6 | tee gpurun_out/s29/pipeline.log

}

s_r03_30() {
# round 3, GPU session 30: the bench line at the driver's command (--steps 20 --warmup 5) by solves in flight: end effects of a short timed region
O=gpurun_out/s30; mkdir -p $O
export TMPDIR=/tmp
{
for nf in 4 5 6 8 10; do for rep in 1 2; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --timed-only --in-flight $nf 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps 20, $nf in flight: %.0f solves/s %.2f ms chip %.3f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac']))"
done; done
for rep in 1 2; do python bench.py --gpus 1 --steps 60 --warmup 5 --no-cpu-baseline --timed-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps 60, 6 in flight: %.0f solves/s %.2f ms' % (d['value'], d['ms_per_step']))"; done
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json ) 2>&1 | grep real
python -c "import json; d=json.load(open('$O/bench_driver_cmd.json')); print('driver command: %.0f solves/s %.2f ms' % (d['value'], d['ms_per_step']))"
} 2>&1 | tee $O/short_runs.log

}

s_r03_31() {
# round 3, GPU session 31: hardware queues beyond eight (bench line, host pipeline inside bench.py)
O=gpurun_out/s31; mkdir -p $O
export TMPDIR=/tmp
{
for q in 8 12 16; do
  GPU_MAX_HW_QUEUES=$q python bench.py --gpus 1 --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('GPU_MAX_HW_QUEUES=$q: %.0f solves/s %.2f ms | pipelined from host arrays %.0f | tracking %.0f | c3 %.0f c4 %.0f' % (d['value'], d['ms_per_step'], d['host_pointer_pipelined']['value'], d['tracking_seeds']['value'], d['configs']['c3']['value'], d['configs']['c4']['value']))"
done
} 2>&1 | tee $O/queues.log

}

s_r03_32() {
# round 3, GPU session 32: per-phase cycles of a C2 step under the dense mapping of the throughput schedule (both species on one wavefront), full chip and lone
O=gpurun_out/s32; mkdir -p $O
export TMPDIR=/tmp
H="BIOIK_SOLVE_THREADS=64 BIOIK_SOLVE_SPECIES_PARALLEL=1 BIOIK_SOLVE_COLUMNLESS=2"
{
env $H BIOIK_SOLVE_REPORT=1 BIOIK_HIP_LIBRARY=build/libphase.so python tools/phase_probe_config.py c2 3072
env $H BIOIK_HIP_LIBRARY=build/libphase.so python tools/phase_probe_config.py c2 1
BIOIK_HIP_LIBRARY=build/libphase.so python tools/phase_probe_config.py c2 1536
} 2>&1 | grep -v "amdgpu.ids\|Warning\|getlimits\|machar" | tee $O/phases_dense.log

}

s_r03_33() {
# round 3, GPU session 33: the dense kernel of the throughput schedule -- two-minima top-2 for its 32-lane groups (z1), winner copy skipped when the elites stay (z2),
# both (z3) -- against the build before (z0): fixed work under the dense mapping and the bench line (six in flight)
O=gpurun_out/s33; mkdir -p $O
export TMPDIR=/tmp
{
for rep in 1 2; do for lib in build/lib_z0.so build/lib_z1.so build/lib_z2.so build/lib_z3.so; do
  v=$(BIOIK_BENCH_SCHEDULE=throughput BIOIK_BENCH_IN_FLIGHT=1 BIOIK_BENCH_STREAM=0 BIOIK_HIP_LIBRARY=$lib BIOIK_BENCH_DTWIST=1e-300 BIOIK_BENCH_MAX_STEPS=32 BIOIK_BENCH_BATCH=3072 python bench.py --no-cpu-baseline --timed-only --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms -> %.0f steps/ms chip-wide' % (d['ms_per_step'], 3072*32/d['ms_per_step']))")
  echo "$lib fixed work, dense mapping: $v"
done; done
for rep in 1 2 3; do for lib in build/lib_z0.so build/lib_z1.so build/lib_z2.so build/lib_z3.so; do
  BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --timed-only --steps 60 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib bench: %.0f solves/s %.2f ms chip %.3f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac']))"
done; done
} 2>&1 | tee $O/dense_ab.log

}

s_r03_34() {
# round 3, GPU session 34: GPU suite, bench line, profile passes, parity soak (schedule drawn at random too) -- final tree of the round
O=gpurun_out/s34; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gputests.log 2>&1
grep -E "passed|failed" $O/gputests.log
( time bash tools/profile_round.sh r03 ) > $O/profile_round.log 2>&1
python -c "
import json; d=json.load(open('gpurun_out/bench_r03.json'))
print('bench: %.0f solves/s %.2f ms chip_frac %.3f | latency schedule, three in flight %.0f | one-at-a-time %.0f | pipelined %.0f host %.0f | tracking %.0f ref-params %.0f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['latency_schedule_three_in_flight']['value'], d['one_batch_at_a_time']['value'], d['host_pointer_pipelined']['value'], d['host_pointer_entry']['solves_per_s'], d['tracking_seeds']['value'], d['reference_parameters']['value']))
print('cpu', d['cpu_baseline']['value'], 'speedup', d['speedup_vs_cpu_1thread'])
print({k:(round(v['value']), round(v['success_rate'],3), round(v['roofline']['chip_level_frac'],3), round(v.get('speedup_vs_cpu_1thread',0))) for k,v in d['configs'].items()})
"
tail -3 gpurun_out/bench_r03.err
bash tools/step_rate.sh bio_ik_amd/libbioik_hip.so | tail -2
( time timeout 900 python tools/fuzz_parity.py 1500 19 ) > $O/fuzz.log 2>&1
tail -3 $O/fuzz.log
python tools/pipeline_probe.py 2>&1 | grep -v Warning | tail -6 | tee gpurun_out/s34/pipeline.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s34/bench_driver_cmd.json ) 2>&1 | grep real; python -c "import json; d=json.load(open(\"gpurun_out/s34/bench_driver_cmd.json\")); print(\"driver command: %.0f solves/s %.2f ms, %d in flight\" % (d[\"value\"], d[\"ms_per_step\"], d[\"config\"][\"batches_in_flight\"]))"

}

s_r03_35() {
# round 3, GPU session 35: the bench line of record (default command) after the last harness change
O=gpurun_out/s35; mkdir -p $O
export TMPDIR=/tmp
python bench.py > gpurun_out/bench_r03.json 2> $O/bench.err
python -c "
import json; d=json.load(open('gpurun_out/bench_r03.json'))
print('bench: %.0f solves/s %.2f ms chip_frac %.3f | latency schedule, three in flight %.0f | one-at-a-time %.0f | pipelined %.0f host %.0f | tracking %.0f ref-params %.0f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['latency_schedule_three_in_flight']['value'], d['one_batch_at_a_time']['value'], d['host_pointer_pipelined']['value'], d['host_pointer_entry']['solves_per_s'], d['tracking_seeds']['value'], d['reference_parameters']['value']))
"

}

s_r03_36() {
# round 3, GPU session 36: GPU suite, bench line, profile passes, parity soak -- final tree of the round (BIOIK_SCHEDULE_AUTO added)
O=gpurun_out/s36; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gputests.log 2>&1
grep -E "passed|failed" $O/gputests.log
( time bash tools/profile_round.sh r03 ) > $O/profile_round.log 2>&1
python -c "
import json; d=json.load(open('gpurun_out/bench_r03.json'))
print('bench: %.0f solves/s %.2f ms chip_frac %.3f | latency schedule, three in flight %.0f | one-at-a-time %.0f | pipelined %.0f host %.0f | tracking %.0f ref-params %.0f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['latency_schedule_three_in_flight']['value'], d['one_batch_at_a_time']['value'], d['host_pointer_pipelined']['value'], d['host_pointer_entry']['solves_per_s'], d['tracking_seeds']['value'], d['reference_parameters']['value']))
print('cpu', d['cpu_baseline']['value'], 'speedup', d['speedup_vs_cpu_1thread'])
print({k:(round(v['value']), round(v['success_rate'],3), round(v['roofline']['chip_level_frac'],3), round(v.get('speedup_vs_cpu_1thread',0))) for k,v in d['configs'].items()})
"
tail -3 gpurun_out/bench_r03.err
bash tools/step_rate.sh bio_ik_amd/libbioik_hip.so | tail -2
( time timeout 900 python tools/fuzz_parity.py 1500 23 ) > $O/fuzz.log 2>&1
tail -3 $O/fuzz.log
python tools/pipeline_probe.py 2>&1 | grep -v Warning | tail -6 | tee gpurun_out/s36/pipeline.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s36/bench_driver_cmd.json ) 2>&1 | grep real; python -c "import json; d=json.load(open(\"gpurun_out/s36/bench_driver_cmd.json\")); print(\"driver command: %.0f solves/s %.2f ms, %d in flight\" % (d[\"value\"], d[\"ms_per_step\"], d[\"config\"][\"batches_in_flight\"]))"

}

s_r03_37() {
# round 3, GPU session 37: dry run of the N = 2 control flow of bench.py on a one-GPU box (two ranks sharing the device, gloo for the barrier and the reductions)
O=gpurun_out/s37; mkdir -p $O
export TMPDIR=/tmp
BIOIK_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_two_ranks.json 2> $O/bench_two_ranks.err
echo rc=$?
python -c "
import json; d=json.loads([l for l in open('$O/bench_two_ranks.json') if l.startswith('{')][-1])
print('two ranks on one GPU (gloo): %.0f solves/s %.2f ms n_gpus %d in flight %d' % (d['value'], d['ms_per_step'], d['n_gpus'], d['config']['batches_in_flight']))"
tail -3 $O/bench_two_ranks.err

}

s_r03_38() {
# round 3, GPU session 38: the dense kernel compiled for single-wavefront workgroups (__launch_bounds__(64, 3): BIOIK_SOLVE_CL64; (64, 4): BIOIK_SOLVE_CL64W4)
O=gpurun_out/s38; mkdir -p $O
export TMPDIR=/tmp
{
for rep in 1 2 3; do for v in "X=1" "BIOIK_SOLVE_CL64=1" "BIOIK_SOLVE_CL64W4=1"; do
  env $v python bench.py --no-cpu-baseline --timed-only --steps 60 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v bench: %.0f solves/s %.2f ms chip %.3f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac']))"
done; done
for v in "X=1" "BIOIK_SOLVE_CL64=1" "BIOIK_SOLVE_CL64W4=1"; do
  r=$(env $v BIOIK_BENCH_SCHEDULE=throughput BIOIK_BENCH_IN_FLIGHT=1 BIOIK_BENCH_STREAM=0 BIOIK_BENCH_DTWIST=1e-300 BIOIK_BENCH_MAX_STEPS=32 BIOIK_BENCH_BATCH=3072 python bench.py --no-cpu-baseline --timed-only --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms -> %.0f steps/ms chip-wide' % (d['ms_per_step'], 3072*32/d['ms_per_step']))")
  echo "$v fixed work, dense mapping: $r"
done
} 2>&1 | tee $O/cl64.log

}

s_r03_39() {
# round 3, GPU session 39: the throughput schedule on the four-wavefront dense kernel: bench line, solves in flight, host pipeline, GPU suite, parity soak
O=gpurun_out/s39; mkdir -p $O
export TMPDIR=/tmp
{
for nf in 5 6 8 12; do
  python bench.py --no-cpu-baseline --timed-only --in-flight $nf --steps 60 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$nf in flight: %.0f solves/s %.2f ms chip %.3f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac']))"
done
BIOIK_SOLVE_THREE_WAVES=1 python bench.py --no-cpu-baseline --timed-only --steps 60 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('three wavefronts per SIMD, 6 in flight: %.0f solves/s %.2f ms' % (d['value'], d['ms_per_step']))"
python bench.py --no-cpu-baseline --steps 60 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench: %.0f solves/s %.2f ms chip %.3f | latency schedule, three in flight %.0f | one-at-a-time %.0f | pipelined %.0f host %.0f | tracking %.0f ref-params %.0f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['latency_schedule_three_in_flight']['value'], d['one_batch_at_a_time']['value'], d['host_pointer_pipelined']['value'], d['host_pointer_entry']['solves_per_s'], d['tracking_seeds']['value'], d['reference_parameters']['value']))"
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --timed-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('driver command: %.0f solves/s %.2f ms' % (d['value'], d['ms_per_step']))"
python tools/pipeline_probe.py 2>&1 | grep -v Warning | tail -9
} 2>&1 | tee $O/w4.log
( time timeout 1500 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -4 | tee $O/gputests.log
( time timeout 900 python tools/fuzz_parity.py 1000 29 ) > $O/fuzz.log 2>&1; tail -3 $O/fuzz.log

}

s_r03_40() {
# round 3, GPU session 40: GPU suite, bench line, profile passes, parity soak -- final tree (four-wavefront dense kernel for the throughput schedule)
O=gpurun_out/s40; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gputests.log 2>&1
grep -E "passed|failed" $O/gputests.log
( time bash tools/profile_round.sh r03 ) > $O/profile_round.log 2>&1
python -c "
import json; d=json.load(open('gpurun_out/bench_r03.json'))
print('bench: %.0f solves/s %.2f ms chip_frac %.3f | latency schedule, three in flight %.0f | one-at-a-time %.0f | pipelined %.0f host %.0f | tracking %.0f ref-params %.0f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['latency_schedule_three_in_flight']['value'], d['one_batch_at_a_time']['value'], d['host_pointer_pipelined']['value'], d['host_pointer_entry']['solves_per_s'], d['tracking_seeds']['value'], d['reference_parameters']['value']))
print('cpu', d['cpu_baseline']['value'], 'speedup', d['speedup_vs_cpu_1thread'])
print({k:(round(v['value']), round(v['success_rate'],3), round(v['roofline']['chip_level_frac'],3), round(v.get('speedup_vs_cpu_1thread',0))) for k,v in d['configs'].items()})
"
tail -3 gpurun_out/bench_r03.err
bash tools/step_rate.sh bio_ik_amd/libbioik_hip.so | tail -2
( time timeout 900 python tools/fuzz_parity.py 1500 31 ) > $O/fuzz.log 2>&1
tail -3 $O/fuzz.log
python tools/pipeline_probe.py 2>&1 | grep -v Warning | tail -6 | tee gpurun_out/s40/pipeline.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s40/bench_driver_cmd.json ) 2>&1 | grep real; python -c "import json; d=json.load(open(\"gpurun_out/s40/bench_driver_cmd.json\")); print(\"driver command: %.0f solves/s %.2f ms, %d in flight\" % (d[\"value\"], d[\"ms_per_step\"], d[\"config\"][\"batches_in_flight\"]))"

}

s_r03_41() {
# round 3, GPU session 41: the four-wavefront dense kernel with the species record read per generation (SLIM: 66 -> 30 spilled values): bench line, HBM traffic
O=gpurun_out/s41; mkdir -p $O
export TMPDIR=/tmp
R=$(pwd)
{
for rep in 1 2 3; do
  python bench.py --no-cpu-baseline --timed-only --steps 60 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('slim w4: %.0f solves/s %.2f ms chip %.3f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac']))"
  BIOIK_SOLVE_THREE_WAVES=1 python bench.py --no-cpu-baseline --timed-only --steps 60 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('w3: %.0f solves/s %.2f ms' % (d['value'], d['ms_per_step']))"
done
python bench.py --no-cpu-baseline --steps 60 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench: %.0f solves/s | tracking %.0f | pipelined %.0f' % (d['value'], d['tracking_seeds']['value'], d['host_pointer_pipelined']['value']))"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  BIOIK_BENCH_STREAM=0 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$O/pmc_$c -o $c -- python $R/bench.py --timed-only --steps 5 --warmup 1 --in-flight 1 > $R/$O/pmc_$c.log 2>&1
  python - <<PY
import csv,glob
f=glob.glob("$R/$O/pmc_$c/**/*counter_collection.csv", recursive=True)[0]
tot=0; n=0
for r in csv.DictReader(open(f)):
    if r["Kernel_Name"].startswith("k_solve") and r["Counter_Name"]=="$c": tot+=float(r["Counter_Value"]); n+=1
print("$c per solve launch: %.1f KiB over %d dispatches" % (tot/max(n,1), n))
PY
done
} 2>&1 | tee $R/$O/slim.log

}

s_r03_42() {
# round 3, GPU session 42: GPU suite, bench line, profile passes, parity soak -- final tree (four-wavefront dense kernel, species record per generation)
O=gpurun_out/s42; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gputests.log 2>&1
grep -E "passed|failed" $O/gputests.log
( time bash tools/profile_round.sh r03 ) > $O/profile_round.log 2>&1
python -c "
import json; d=json.load(open('gpurun_out/bench_r03.json'))
print('bench: %.0f solves/s %.2f ms chip_frac %.3f | latency schedule, three in flight %.0f | one-at-a-time %.0f | pipelined %.0f host %.0f | tracking %.0f ref-params %.0f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['latency_schedule_three_in_flight']['value'], d['one_batch_at_a_time']['value'], d['host_pointer_pipelined']['value'], d['host_pointer_entry']['solves_per_s'], d['tracking_seeds']['value'], d['reference_parameters']['value']))
print('cpu', d['cpu_baseline']['value'], 'speedup', d['speedup_vs_cpu_1thread'])
print({k:(round(v['value']), round(v['success_rate'],3), round(v['roofline']['chip_level_frac'],3), round(v.get('speedup_vs_cpu_1thread',0))) for k,v in d['configs'].items()})
"
tail -3 gpurun_out/bench_r03.err
bash tools/step_rate.sh bio_ik_amd/libbioik_hip.so | tail -2
( time timeout 900 python tools/fuzz_parity.py 1500 37 ) > $O/fuzz.log 2>&1
tail -3 $O/fuzz.log
python tools/pipeline_probe.py 2>&1 | grep -v Warning | tail -6 | tee gpurun_out/s42/pipeline.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s42/bench_driver_cmd.json ) 2>&1 | grep real; python -c "import json; d=json.load(open(\"gpurun_out/s42/bench_driver_cmd.json\")); print(\"driver command: %.0f solves/s %.2f ms, %d in flight\" % (d[\"value\"], d[\"ms_per_step\"], d[\"config\"][\"batches_in_flight\"]))"

}

s_r03_43() {
# round 3, GPU session 43: GPU suite, bench line, profile passes, parity soak -- the final sources of the round
O=gpurun_out/s43; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gputests.log 2>&1
grep -E "passed|failed" $O/gputests.log
( time bash tools/profile_round.sh r03 ) > $O/profile_round.log 2>&1
python -c "
import json; d=json.load(open('gpurun_out/bench_r03.json'))
print('bench: %.0f solves/s %.2f ms chip_frac %.3f | latency schedule, three in flight %.0f | one-at-a-time %.0f | pipelined %.0f host %.0f | tracking %.0f ref-params %.0f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['latency_schedule_three_in_flight']['value'], d['one_batch_at_a_time']['value'], d['host_pointer_pipelined']['value'], d['host_pointer_entry']['solves_per_s'], d['tracking_seeds']['value'], d['reference_parameters']['value']))
print('cpu', d['cpu_baseline']['value'], 'speedup', d['speedup_vs_cpu_1thread'])
print({k:(round(v['value']), round(v['success_rate'],3), round(v['roofline']['chip_level_frac'],3), round(v.get('speedup_vs_cpu_1thread',0))) for k,v in d['configs'].items()})
"
tail -3 gpurun_out/bench_r03.err
bash tools/step_rate.sh bio_ik_amd/libbioik_hip.so | tail -2
( time timeout 900 python tools/fuzz_parity.py 1500 41 ) > $O/fuzz.log 2>&1
tail -3 $O/fuzz.log
python tools/pipeline_probe.py 2>&1 | grep -v Warning | tail -6 | tee gpurun_out/s43/pipeline.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s43/bench_driver_cmd.json ) 2>&1 | grep real; python -c "import json; d=json.load(open(\"gpurun_out/s43/bench_driver_cmd.json\")); print(\"driver command: %.0f solves/s %.2f ms, %d in flight\" % (d[\"value\"], d[\"ms_per_step\"], d[\"config\"][\"batches_in_flight\"]))"

}

s_r03_44() {
# round 3, GPU session 44: C3's joint-walk kernel under the four-wavefront budget with the species record per generation (BIOIK_SOLVE_CLJ4=1)
O=gpurun_out/s44; mkdir -p $O
export TMPDIR=/tmp
run() { python bench.py --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1: configs', {k:(round(v['value']),round(v['ms_per_step'],1),round(v['roofline']['chip_level_frac'],3)) for k,v in d['configs'].items()})"; }
{
for rep in 1 2 3; do
run clj
BIOIK_SOLVE_CLJ4=1 run clj4
done
} 2>&1 | tee $O/clj4.log

}

s_r03_45() {
# round 3, GPU session 45: C4's 128-register kernel with the species record read per generation (BIOIK_SOLVE_CL4S=1: 67 -> 31 spilled values)
O=gpurun_out/s45; mkdir -p $O
export TMPDIR=/tmp
run() { python bench.py --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1: configs', {k:(round(v['value']),round(v['ms_per_step'],1),round(v['success_rate'],3),round(v['roofline']['chip_level_frac'],3)) for k,v in d['configs'].items()})"; }
{
for rep in 1 2 3; do
run cl4
BIOIK_SOLVE_CL4S=1 run cl4s
done
} 2>&1 | tee $O/cl4s.log

}

s_r03_46() {
# round 3, GPU session 46: GPU suite, bench line, profile passes, parity soak -- the final sources of the round
O=gpurun_out/s46; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gputests.log 2>&1
grep -E "passed|failed" $O/gputests.log
( time bash tools/profile_round.sh r03 ) > $O/profile_round.log 2>&1
python -c "
import json; d=json.load(open('gpurun_out/bench_r03.json'))
print('bench: %.0f solves/s %.2f ms chip_frac %.3f | latency schedule, three in flight %.0f | one-at-a-time %.0f | pipelined %.0f host %.0f | tracking %.0f ref-params %.0f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['latency_schedule_three_in_flight']['value'], d['one_batch_at_a_time']['value'], d['host_pointer_pipelined']['value'], d['host_pointer_entry']['solves_per_s'], d['tracking_seeds']['value'], d['reference_parameters']['value']))
print('cpu', d['cpu_baseline']['value'], 'speedup', d['speedup_vs_cpu_1thread'])
print({k:(round(v['value']), round(v['success_rate'],3), round(v['roofline']['chip_level_frac'],3), round(v.get('speedup_vs_cpu_1thread',0))) for k,v in d['configs'].items()})
"
tail -3 gpurun_out/bench_r03.err
bash tools/step_rate.sh bio_ik_amd/libbioik_hip.so | tail -2
( time timeout 900 python tools/fuzz_parity.py 1500 43 ) > $O/fuzz.log 2>&1
tail -3 $O/fuzz.log
python tools/pipeline_probe.py 2>&1 | grep -v Warning | tail -6 | tee gpurun_out/s46/pipeline.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s46/bench_driver_cmd.json ) 2>&1 | grep real; python -c "import json; d=json.load(open(\"gpurun_out/s46/bench_driver_cmd.json\")); print(\"driver command: %.0f solves/s %.2f ms, %d in flight\" % (d[\"value\"], d[\"ms_per_step\"], d[\"config\"][\"batches_in_flight\"]))"

}

s_r03_47() {
# round 3, GPU session 47: solves in flight at the driver's command (--steps 20) and at the default (--steps 60) on the final kernels
O=gpurun_out/s47; mkdir -p $O
export TMPDIR=/tmp
{
for st in 20 60; do for nf in 4 5 6 10 12; do for rep in 1 2; do
  python bench.py --gpus 1 --steps $st --warmup 5 --no-cpu-baseline --timed-only --in-flight $nf 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps $st, $nf in flight: %.0f solves/s %.2f ms chip %.3f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac']))"
done; done; done
} 2>&1 | tee $O/inflight.log

}

s_r03_48() {
# the driver's own command with the automatic number of solves in flight (ten at K = 20)
mkdir -p gpurun_out
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_command.json 2> gpurun_out/bench_driver_command.err
tail -c 600 gpurun_out/bench_driver_command.json

}

s_r03_49() {
# a longer parity soak on the final sources: 20000 further cases (another seed), whole solves bit for bit against the oracle
mkdir -p gpurun_out
( time timeout 1200 python tools/fuzz_parity.py 20000 4711 ) > gpurun_out/fuzz_long.log 2>&1
tail -4 gpurun_out/fuzz_long.log

}

s_r03_50() {
# round 3, GPU session 50 (the same as 37, with the automatic ten solves in flight): dry run of the N = 2 control flow of bench.py on a one-GPU box (two ranks sharing the device, gloo for the barrier and the reductions)
O=gpurun_out/s50; mkdir -p $O
export TMPDIR=/tmp
BIOIK_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_two_ranks.json 2> $O/bench_two_ranks.err
echo rc=$?
python -c "
import json; d=json.loads([l for l in open('$O/bench_two_ranks.json') if l.startswith('{')][-1])
print('two ranks on one GPU (gloo): %.0f solves/s %.2f ms n_gpus %d in flight %d' % (d['value'], d['ms_per_step'], d['n_gpus'], d['config']['batches_in_flight']))"
tail -3 $O/bench_two_ranks.err

}

s_r03_51() {
# round 3, GPU session 51: an isolated call (one solve after the other) and three in flight under BIOIK_SCHEDULE_LATENCY:
# hand-over after K steps x the first launch's kernel (168-register computed-children kernel / its 128-register build: 4096 wavefronts = the whole
# batch resident at once)
O=gpurun_out/s51; mkdir -p $O
export TMPDIR=/tmp
for inf in 1 3; do for cl in 0 1; do for k in 1 4 8 12 16 24; do
  if [ $cl = 1 ]; then export BIOIK_SOLVE_CL64W4=1; else unset BIOIK_SOLVE_CL64W4; fi
  v=$(BIOIK_SOLVE_TWO_PHASE=$k timeout 120 python bench.py --timed-only --no-cpu-baseline --schedule latency --in-flight $inf --steps 24 --warmup 3 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s %.2f ms' % (d['value'], d['ms_per_step']))")
  echo "in flight $inf, first launch $( [ $cl = 1 ] && echo '128-register' || echo '168-register' ), hand-over after $k: $v"
done; done; done 2>&1 | tee $O/handover_sweep.log

}

s_r03_52() {
# round 3, GPU session 52: the candidates of session 51 on the tracking seeds (seed = target + N(0, 0.1 rad)) as well: BIOIK_SCHEDULE_LATENCY, three in flight
O=gpurun_out/s52; mkdir -p $O
export TMPDIR=/tmp BIOIK_BENCH_CONFIGS=0
for v in "1 0" "4 1" "8 1" "12 1" "16 1"; do set -- $v
  if [ $2 = 1 ]; then export BIOIK_SOLVE_CL64W4=1; else unset BIOIK_SOLVE_CL64W4; fi
  r=$(BIOIK_SOLVE_TWO_PHASE=$1 timeout 300 python bench.py --no-cpu-baseline --schedule latency --in-flight 3 --steps 24 --warmup 3 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print('three in flight %.0f  one at a time %.0f  tracking seeds %.0f (mean steps %.2f)' % (d['value'], d['one_batch_at_a_time']['value'], d['tracking_seeds']['value'], d['tracking_seeds']['mean_steps_per_solve']))")
  echo "hand-over after $1, first launch $( [ $2 = 1 ] && echo '128-register' || echo '168-register' ): $r"
done 2>&1 | tee $O/handover_tracking.log

}

s_r03_53() {
# round 3, GPU session 53: the GPU suite and smoke() on the final tree (what the driver runs at round end)
O=gpurun_out/s53; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/ -x -q -m gpu ) > $O/gpu_suite.log 2>&1; echo "suite rc=$?"
tail -4 $O/gpu_suite.log
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" ) > $O/smoke.log 2>&1; echo "smoke rc=$?"
tail -4 $O/smoke.log

}

s_r03_54() {
# round 3, GPU session 54: the C3 / C4 legs of bench.py time ONE launch per stream (all in flight at once: start of the first to the end of the last,
# no steady state).  The same legs with 1 / 2 / 3 / 6 launches per stream, six and ten streams.
O=gpurun_out/s54; mkdir -p $O
export TMPDIR=/tmp
for inf in 6 10; do for rounds in 1 2 3 6; do
  r=$(BIOIK_BENCH_CONFIG_ROUNDS=$rounds timeout 600 python bench.py --no-cpu-baseline --in-flight $inf --steps $((inf * 2)) --warmup 2 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); c=d['configs']
print('C3 %.0f solves/s (chip-level %.3f, %d launches)  C4 %.0f solves/s (chip-level %.3f)' % (c['c3']['value'], c['c3']['roofline']['chip_level_frac'], c['c3']['batches_timed'], c['c4']['value'], c['c4']['roofline']['chip_level_frac']))")
  echo "$inf in flight, $rounds per stream: $r"
done; done 2>&1 | tee $O/config_rounds.log

}

s_r03_55() {
# round 3, GPU session 55: the profile of record again (bench line with six timed launches per stream in the C3 / C4 legs, kernel statistics, PMC passes)
# and the driver's own bench command
O=gpurun_out/s55; mkdir -p $O
export TMPDIR=/tmp
( time bash tools/profile_round.sh r03 ) > $O/profile_round.log 2>&1
python -c "
import json; d=json.load(open('gpurun_out/bench_r03.json'))
print('bench: %.0f solves/s %.2f ms chip_frac %.3f | latency schedule, three in flight %.0f | one-at-a-time %.0f | pipelined %.0f host %.0f | tracking %.0f ref-params %.0f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['latency_schedule_three_in_flight']['value'], d['one_batch_at_a_time']['value'], d['host_pointer_pipelined']['value'], d['host_pointer_entry']['solves_per_s'], d['tracking_seeds']['value'], d['reference_parameters']['value']))
print('cpu', d['cpu_baseline']['value'], 'speedup', d['speedup_vs_cpu_1thread'])
print({k:(round(v['value']), v['batches_timed'], round(v['success_rate'],3), round(v['roofline']['chip_level_frac'],3), round(v.get('speedup_vs_cpu_1thread',0))) for k,v in d['configs'].items()})
"
tail -3 gpurun_out/bench_r03.err
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json ) 2>&1 | grep real
python -c "
import json; d=json.loads([l for l in open('$O/bench_driver_cmd.json') if l.startswith('{')][-1]); c=d['configs']
print('driver command: %.0f solves/s %.2f ms, %d in flight; C3 %.0f (%.3f) C4 %.0f (%.3f)' % (d['value'], d['ms_per_step'], d['config']['batches_in_flight'], c['c3']['value'], c['c3']['roofline']['chip_level_frac'], c['c4']['value'], c['c4']['roofline']['chip_level_frac']))"

}

s_r03_56() {
# round 3, GPU session 56: the driver's bench command three times with every stream opened before the warm-up (session 55 measured 6.9e5 with five of the
# ten streams first used inside the timed region), then the default command
O=gpurun_out/s56; mkdir -p $O
export TMPDIR=/tmp BIOIK_BENCH_CONFIGS=0
for i in 1 2 3; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('driver command: %.0f solves/s %.2f ms, %d in flight' % (d['value'], d['ms_per_step'], d['config']['batches_in_flight']))"
done 2>&1 | tee $O/driver_command.log
python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('default command: %.0f solves/s %.2f ms, %d in flight, chip-level %.3f' % (d['value'], d['ms_per_step'], d['config']['batches_in_flight'], d['roofline']['chip_level_frac']))" 2>&1 | tee -a $O/driver_command.log

}

s_r03_57() {
# round 3, GPU session 57 (= 55 after bench.py opens every stream before the warm-up): the profile of record again (bench line with six timed launches per stream in the C3 / C4 legs, kernel statistics, PMC passes)
# and the driver's own bench command
O=gpurun_out/s57; mkdir -p $O
export TMPDIR=/tmp
( time bash tools/profile_round.sh r03 ) > $O/profile_round.log 2>&1
python -c "
import json; d=json.load(open('gpurun_out/bench_r03.json'))
print('bench: %.0f solves/s %.2f ms chip_frac %.3f | latency schedule, three in flight %.0f | one-at-a-time %.0f | pipelined %.0f host %.0f | tracking %.0f ref-params %.0f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['latency_schedule_three_in_flight']['value'], d['one_batch_at_a_time']['value'], d['host_pointer_pipelined']['value'], d['host_pointer_entry']['solves_per_s'], d['tracking_seeds']['value'], d['reference_parameters']['value']))
print('cpu', d['cpu_baseline']['value'], 'speedup', d['speedup_vs_cpu_1thread'])
print({k:(round(v['value']), v['batches_timed'], round(v['success_rate'],3), round(v['roofline']['chip_level_frac'],3), round(v.get('speedup_vs_cpu_1thread',0))) for k,v in d['configs'].items()})
"
tail -3 gpurun_out/bench_r03.err
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json ) 2>&1 | grep real
python -c "
import json; d=json.loads([l for l in open('$O/bench_driver_cmd.json') if l.startswith('{')][-1]); c=d['configs']
print('driver command: %.0f solves/s %.2f ms, %d in flight; C3 %.0f (%.3f) C4 %.0f (%.3f)' % (d['value'], d['ms_per_step'], d['config']['batches_in_flight'], c['c3']['value'], c['c3']['roofline']['chip_level_frac'], c['c4']['value'], c['c4']['roofline']['chip_level_frac']))"

}

s_r03_58() {
# round 3, GPU session 58: the host-pointer pipeline leg of bench.py fell from 9.2e5 to 7.4e5 when the device-pointer legs went from six to ten streams
# (ten + six I/O streams + the default stream on 16 hardware queues).  GPU_MAX_HW_QUEUES 16 / 24 / 32, and the stand-alone probe.
O=gpurun_out/s58; mkdir -p $O
export TMPDIR=/tmp BIOIK_BENCH_CONFIGS=0
for q in 16 24 32; do
  GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('hardware queues %s: %.0f solves/s, %d in flight | host-pointer pipeline %.0f | host-pointer entry %.0f | tracking %.0f' % (d['config']['hardware_queues'], d['value'], d['config']['batches_in_flight'], d['host_pointer_pipelined']['value'], d['host_pointer_entry']['solves_per_s'], d['tracking_seeds']['value']))"
done 2>&1 | tee $O/hwq.log
python tools/pipeline_probe.py 2>&1 | grep -v Warning | tail -6 | tee -a $O/hwq.log

}

s_r03_59() {
# round 3, GPU session 59: BASELINE.json configs[4] (mixed PR2 / snake batch, sorted by model, end to end through solve_mixed) on one GPU:
# the round-2 size (16384) and the full 262144
O=gpurun_out/s59; mkdir -p $O
export TMPDIR=/tmp
BIOIK_BENCH_C5_BATCH=16384 timeout 600 python bench.py --config c5 --steps 5 --warmup 2 2>/dev/null | grep '^{' | tail -1 > $O/bench_c5_16384.json
timeout 900 python bench.py --config c5 --steps 3 --warmup 1 2>/dev/null | grep '^{' | tail -1 > $O/bench_c5_262144.json
for f in $O/bench_c5_16384.json $O/bench_c5_262144.json; do python -c "
import json; d=json.load(open('$f')); print('$f: %.0f solves/s, %.1f ms per batch' % (d['value'], d['ms_per_step']), d['config'])"; done

}

s_r03_60() {
# a third parity soak on the final sources: 20000 cases, another seed
mkdir -p gpurun_out
( time timeout 1200 python tools/fuzz_parity.py 20000 90210 ) > gpurun_out/fuzz_long2.log 2>&1
tail -4 gpurun_out/fuzz_long2.log

}

s_r03_61() {
# round 3, GPU session 61 (run on two boxes): the default bench command and the driver's, final bench.py
mkdir -p gpurun_out/s61; export TMPDIR=/tmp
for a in "" "--gpus 1 --steps 20 --warmup 5"; do
python bench.py $a 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); c=d['configs']
print('bench.py $a: %.0f solves/s %.2f ms chip-level %.3f, %d in flight | latency schedule three in flight %.0f | one at a time %.0f | host-pointer pipeline %.0f | tracking %.0f | C3 %.0f (%.3f) C4 %.0f (%.3f) | cpu %.0f -> %.0fx' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['config']['batches_in_flight'], d['latency_schedule_three_in_flight']['value'], d['one_batch_at_a_time']['value'], d['host_pointer_pipelined']['value'], d['tracking_seeds']['value'], c['c3']['value'], c['c3']['roofline']['chip_level_frac'], c['c4']['value'], c['c4']['roofline']['chip_level_frac'], d['cpu_baseline']['value'], d['speedup_vs_cpu_1thread']))"
done | tee -a gpurun_out/s61/bench_lines.log

}

s_r04_1() {
# round 4, GPU session 1: the GPU suite on the kernels compiled for one mapping (solve_body<.., FIXED>), then A/B of library builds on fixed work (throughput
# schedule) and on the driver's bench command: lib_r03 = round 3's final sources, lib_a1 = FIXED kernels without register spills, lib_a2 = + the clamp as
# two instructions; BIOIK_SOLVE_DENSE_HANDOVER = a throughput solve hands its stragglers to the latency mapping after K steps
O=gpurun_out/r04s1; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/gpu_suite.log 2>&1
tail -4 $O/gpu_suite.log
SCHEDULE=throughput ROUNDS=1 bash tools/step_rate.sh build/ab/lib_r03.so build/ab/lib_a1.so build/ab/lib_a2.so 2>&1 | tee $O/step_rate_throughput.log
line() { python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d.get('configs',{})
print('$1: %.0f solves/s %.2f ms chip %.3f success %.4f | one-at-a-time %.0f | lat3 %.0f | pipelined %.0f | tracking %.0f ref-params %.0f |' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['success_rate'], d['one_batch_at_a_time']['value'], d['latency_schedule_three_in_flight']['value'], d['host_pointer_pipelined']['value'], d['tracking_seeds']['value'], d['reference_parameters']['value']), {k:(round(v['value']), round(v['roofline']['chip_level_frac'],3)) for k,v in c.items()})"; }
for lib in build/ab/lib_r03.so build/ab/lib_a2.so; do
  BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | line "$lib driver-cmd" | tee -a $O/bench_ab.log
done
for K in 12 16 24; do
  BIOIK_SOLVE_DENSE_HANDOVER=$K BIOIK_BENCH_CONFIGS=0 BIOIK_HIP_LIBRARY=build/ab/lib_a2.so python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | line "lib_a2 handover=$K driver-cmd" | tee -a $O/bench_ab.log
done
for K in 0 16; do
  BIOIK_SOLVE_DENSE_HANDOVER=$K BIOIK_BENCH_CONFIGS=0 BIOIK_HIP_LIBRARY=build/ab/lib_a2.so python bench.py --no-cpu-baseline --timed-only --steps 60 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('lib_a2 handover=$K 60 steps timed-only: %.0f solves/s %.2f ms chip %.3f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac']))" | tee -a $O/bench_ab.log
done

}

s_r04_2() {
# round 4, GPU session 2: lib_a3 = lib_a2 + both winners of a generation copied at once (half-wavefront groups, <= 16 ops) + the serial-chain pair walk in the
# kernels compiled for one mapping (no frame copies per joint); GPU suite on it, fixed work, the driver's bench command
O=gpurun_out/r04s2; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/gpu_suite.log 2>&1
tail -4 $O/gpu_suite.log
SCHEDULE=throughput ROUNDS=2 bash tools/step_rate.sh build/ab/lib_a2.so build/ab/lib_a3.so 2>&1 | tee $O/step_rate_throughput.log
line() { python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d.get('configs',{})
print('$1: %.0f solves/s %.2f ms chip %.3f success %.4f | one-at-a-time %.0f | lat3 %.0f | pipelined %.0f | tracking %.0f ref-params %.0f |' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['success_rate'], d['one_batch_at_a_time']['value'], d['latency_schedule_three_in_flight']['value'], d['host_pointer_pipelined']['value'], d['tracking_seeds']['value'], d['reference_parameters']['value']), {k:(round(v['value']), round(v['roofline']['chip_level_frac'],3)) for k,v in c.items()})"; }
for lib in build/ab/lib_a2.so build/ab/lib_a3.so; do
  BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | line "$lib driver-cmd" | tee -a $O/bench_ab.log
done
BIOIK_BENCH_CONFIGS=0 BIOIK_HIP_LIBRARY=build/ab/lib_a3.so python bench.py --no-cpu-baseline --timed-only --steps 60 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('lib_a3 60 steps timed-only: %.0f solves/s %.2f ms chip %.3f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac']))" | tee -a $O/bench_ab.log

}

s_r04_3() {
# round 4, GPU session 3: GPU suite on the tree with island_sync, the hybrid callback-goal path, bioik_eval_arith; the full default bench line (new CPU legs,
# oracle pose check); `python bench.py --gpus 2` by itself on a one-GPU box (re-executes under torch.distributed.run, gloo since ranks share the device)
O=gpurun_out/r04s3; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/gpu_suite.log 2>&1
tail -5 $O/gpu_suite.log
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
tail -3 $O/bench_default.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04s3/bench_default.json') if l.startswith('{')][-1])
print('bench: %.0f solves/s %.2f ms chip %.3f | lat3 %.0f | one-at-a-time %.0f | pipelined %.0f | tracking %.0f ref-params %.0f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['latency_schedule_three_in_flight']['value'], d['one_batch_at_a_time']['value'], d['host_pointer_pipelined']['value'], d['tracking_seeds']['value'], d['reference_parameters']['value']))
cb=d['cpu_baseline']; print('cpu 1 thread', cb['value'], 'query_parallel', cb.get('query_parallel'), 'v3', cb.get('march_x86_64_v3'))
print('pose check:', d['pose_check'], d['max_pos_err_m_of_successes'], d['max_rot_err_rad_of_successes'])
print({k:(round(v['value']), round(v['roofline']['chip_level_frac'],3)) for k,v in d['configs'].items()})
PY
( time python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_gpus2_plain.json 2> $O/bench_gpus2_plain.err ) 2>&1 | grep real
tail -2 $O/bench_gpus2_plain.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04s3/bench_gpus2_plain.json') if l.startswith('{')][-1])
print('--gpus 2 plain: n_gpus', d['n_gpus'], 'value %.0f' % d['value'], d['process_group'])
PY

}

s_r04_4() {
# round 4, GPU session 4: the parents' mixed momentum from a table.  lib_a3: computed per gene and child; lib_a4: six-row table of the finished momentum term
# (dense kernel only, +768 B of LDS per query); lib_a5: two-row table of the mixed momentum on the species' other elite buffer (no LDS growth), both fixed-mapping kernels
# incl. the pre-selection pass of C4
O=gpurun_out/r04s4; mkdir -p $O
export TMPDIR=/tmp
SCHEDULE=throughput ROUNDS=2 bash tools/step_rate.sh build/ab/lib_a3.so build/ab/lib_a4.so build/ab/lib_a5.so 2>&1 | tee $O/step_rate_throughput.log
line() { python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d.get('configs',{})
print('$1: %.0f solves/s %.2f ms chip %.3f success %.4f | one-at-a-time %.0f | lat3 %.0f | pipelined %.0f | tracking %.0f ref-params %.0f |' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['success_rate'], d['one_batch_at_a_time']['value'], d['latency_schedule_three_in_flight']['value'], d['host_pointer_pipelined']['value'], d['tracking_seeds']['value'], d['reference_parameters']['value']), {k:(round(v['value']), round(v['roofline']['chip_level_frac'],3)) for k,v in c.items()})"; }
for lib in build/ab/lib_a3.so build/ab/lib_a5.so; do
  BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | line "$lib driver-cmd" | tee -a $O/bench_ab.log
done
BIOIK_HIP_LIBRARY=build/ab/lib_a5.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2

}

s_r04_5() {
# round 4, GPU session 5: PMC passes of the dense kernel on FIXED work (no query may succeed, 32 steps, 4096 queries, throughput schedule, one launch at a time):
# instruction classes, VALU busy, waits -- what bounds k_solve_lean_cl64w4 after the register spills are gone
O=$(pwd)/gpurun_out/r04s5; mkdir -p $O
R=$(pwd)
export TMPDIR=/tmp
export BIOIK_BENCH_SCHEDULE=throughput BIOIK_BENCH_IN_FLIGHT=1 BIOIK_BENCH_STREAM=0 BIOIK_BENCH_DTWIST=1e-300 BIOIK_BENCH_MAX_STEPS=32 BIOIK_BENCH_BATCH=4096
python bench.py --no-cpu-baseline --timed-only --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fixed work, 4096 x 32 steps: %.3f ms per launch -> %.0f steps/ms' % (d['ms_per_step'], 4096*32/d['ms_per_step']))" | tee $O/fixed_work.log
cd /tmp
pmc() { d=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$d -o $d -- python $R/bench.py --no-cpu-baseline --timed-only --steps 3 --warmup 1 > $O/pmc_$d.log 2>&1; }
pmc sq SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_WAIT_ANY
pmc mem SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_SALU
pmc mix SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F64
pmc busy SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
cd $R
python - <<'PY'
import csv, glob, collections, os
O='gpurun_out/r04s5'
for f in sorted(glob.glob(O+'/pmc_*/*counter_collection.csv')):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'k_solve' in r['Kernel_Name']:
            agg[(r['Kernel_Name'].split('(')[0], r['Counter_Name'])].append(float(r['Counter_Value']))
            disp={k:r[k] for k in ('Grid_Size','Workgroup_Size','LDS_Block_Size','Scratch_Size','VGPR_Count','SGPR_Count')}
    for (k,c),v in sorted(agg.items()):
        print('%-28s %-28s launches %d mean %.5g' % (k,c,len(v),sum(v)/len(v)))
    print(os.path.basename(f), disp)
PY

}

s_r04_6() {
# round 4, GPU session 6: per-phase shader cycles (-DBIOIK_PHASE_TIMING build of the present sources) of C2 under the throughput schedule (dense kernel), full chip and lone,
# of the reference's own parameters (pop 16, linear), of C3 and C4
O=gpurun_out/r04s6; mkdir -p $O
export BIOIK_HIP_LIBRARY=build/ab/libphase.so BIOIK_SOLVE_REPORT=1
( python tools/phase_probe_config.py c2 4096 throughput; python tools/phase_probe_config.py c2 1 throughput; python tools/phase_probe_config.py ref 4096; python tools/phase_probe_config.py ref 1; python tools/phase_probe_config.py c3 3072; python tools/phase_probe_config.py c4 2048 ) 2>&1 | grep -v "amdgpu.ids" | tee $O/phases.log

}

s_r04_7() {
# round 4, GPU session 7: the reference's own parameters (pop 16, linearised FK) under other lane mappings (diagnostic switches), fixed work:
# what a dense variant for small populations could give
O=gpurun_out/r04s7; mkdir -p $O
probe() { python - "$@" <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from bio_ik_amd import PoseGoal, ProblemTemplate, abi, pr2_like
from bio_ik_amd.solver import HipSolver
from bio_ik_amd.workload import make_queries
t = ProblemTemplate(pr2_like(), "right_arm", [PoseGoal("r_wrist_roll_link")])
h = HipSolver(t, device=0)
n = int(sys.argv[1]); pop = int(sys.argv[2])
seeds, params, _ = make_queries(t, h.active_variables, h.fk_genes, n, seed=5)
dev = torch.device("cuda", 0)
ds, dp = torch.from_numpy(seeds).to(dev), torch.from_numpy(params).to(dev)
o = (torch.empty((n, h.V), dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev))
p = abi.default_solve_params(population=pop, max_steps=32, random_seed=1, fk_mode=abi.FK_LINEAR)
p.dtwist = 1e-300
st = torch.cuda.Stream(dev)
def go():
    h.solve_batch_device(p, n, ds.data_ptr(), dp.data_ptr(), o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), st.cuda_stream)
for _ in range(3): go()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): go()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print("pop %d linear, %d queries x 32 steps: %.3f ms -> %.0f steps/ms   env %s" % (pop, n, dt * 1e3, n * 32 / dt / 1e3, {k: v for k, v in os.environ.items() if k.startswith("BIOIK_SOLVE")}))
PY
}
for n in 4096 8192; do
probe $n 16
BIOIK_SOLVE_THREADS=64 BIOIK_SOLVE_SPECIES_PARALLEL=1 BIOIK_SOLVE_COLUMNLESS=1 probe $n 16
BIOIK_SOLVE_THREADS=64 BIOIK_SOLVE_SPECIES_PARALLEL=1 BIOIK_SOLVE_STORE_CHILDREN=0 probe $n 16
BIOIK_SOLVE_THREADS=64 BIOIK_SOLVE_SPECIES_PARALLEL=0 probe $n 16
probe $n 32
BIOIK_SOLVE_THREADS=64 BIOIK_SOLVE_SPECIES_PARALLEL=1 BIOIK_SOLVE_COLUMNLESS=1 probe $n 32
done 2>&1 | grep -v amdgpu.ids | tee $O/ref_params_mappings.log

}

s_r04_8() {
# round 4, GPU session 8: k_solve_lean_lin (small populations, linearised phenotypes: the reference's own parameters) -- lib_a5 (k_solve_lean, 146 registers, twelve
# queries per CU) against lib_a6 (the kernel compiled for that mapping, 121 registers, sixteen per CU): fixed work and the bench line's reference_parameters leg; GPU suite
O=gpurun_out/r04s8; mkdir -p $O
sed -n '/^probe()/,/^}/p' tools/session_r04_7.sh > /tmp/probe.sh; source /tmp/probe.sh
for lib in build/ab/lib_a5.so build/ab/lib_a6.so; do for n in 4096 8192; do BIOIK_HIP_LIBRARY=$lib probe $n 16 2>&1 | grep -v amdgpu.ids | sed "s|^|$lib |"; done; BIOIK_HIP_LIBRARY=$lib probe 4096 32 2>&1 | grep -v amdgpu.ids | sed "s|^|$lib |"; done | tee $O/fixed_work.log
for lib in build/ab/lib_a5.so build/ab/lib_a6.so; do
BIOIK_BENCH_CONFIGS=0 BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['reference_parameters']
print('$lib: value %.0f | reference_parameters %.0f solves/s (%.2f ms, success %.4f, %.1f steps) | tracking %.0f' % (d['value'], r['value'], r['ms_per_step'], r['success_rate'], r['mean_steps_per_solve'], d['tracking_seeds']['value']))" | tee -a $O/bench_ab.log
done
( time timeout 900 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -5 | tee $O/gpu_suite.log

}

s_r04_9() {
# round 4, GPU session 9: the 128-lane mapping of the latency schedule through k_solve_lean_cl4 (the kernel compiled for that mapping, four wavefronts per SIMD, children
# computed where they are read) against its present kernel k_solve_lean (children kept in columns, 146 registers): fixed work, lone step, and the latency legs of the bench line
O=gpurun_out/r04s9; mkdir -p $O
run() { echo "== $1"; shift; env "$@" SCHEDULE=latency ROUNDS=1 bash tools/step_rate.sh bio_ik_amd/libbioik_hip.so; env "$@" BIOIK_BENCH_BATCH=4096 BIOIK_BENCH_DTWIST=1e-300 BIOIK_BENCH_MAX_STEPS=32 BIOIK_BENCH_SCHEDULE=latency BIOIK_BENCH_IN_FLIGHT=1 BIOIK_BENCH_STREAM=0 python bench.py --no-cpu-baseline --timed-only --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  batch 4096: %.3f ms -> %.0f steps/ms' % (d['ms_per_step'], 4096*32/d['ms_per_step']))"
env "$@" BIOIK_BENCH_CONFIGS=0 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('  bench: value %.0f | lat3 %.0f | one-at-a-time %.0f (%.2f ms) | host entry %.0f | tracking %.0f' % (d['value'], d['latency_schedule_three_in_flight']['value'], d['one_batch_at_a_time']['value'], d['one_batch_at_a_time']['ms_per_step'], d['host_pointer_entry']['solves_per_s'], d['tracking_seeds']['value']))"; }
( run "default (k_solve_lean_cl first step, k_solve_lean the rest)" A=1
  run "k_solve_lean_cl4 for the 128-lane launches" BIOIK_SOLVE_FOUR_WAVES=1 BIOIK_SOLVE_THREADS=128 BIOIK_SOLVE_COLUMNLESS=2
  run "k_solve_lean_cl (168 registers) for the 128-lane launches" BIOIK_SOLVE_THREADS=128 BIOIK_SOLVE_COLUMNLESS=2 ) 2>&1 | tee $O/latency_kernels.log

}

s_r04_10() {
# round 4, GPU session 10: lib_a7 = the latency schedule's 128-lane launches through k_solve_lean_cl4; full bench legs, the dense hand-over to it
# (BIOIK_SOLVE_DENSE_HANDOVER), the two-launch hand-over step (BIOIK_SOLVE_TWO_PHASE), GPU suite
O=gpurun_out/r04s10; mkdir -p $O
line() { python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$1: value %.0f (%.2f ms, chip %.3f) | lat3 %.0f | one-at-a-time %.0f (%.2f ms) | host entry %.0f | pipelined %.0f | tracking %.0f | ref-params %.0f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['latency_schedule_three_in_flight']['value'], d['one_batch_at_a_time']['value'], d['one_batch_at_a_time']['ms_per_step'], d['host_pointer_entry']['solves_per_s'], d['host_pointer_pipelined']['value'], d['tracking_seeds']['value'], d['reference_parameters']['value']))"; }
( for lib in build/ab/lib_a6.so build/ab/lib_a7.so; do BIOIK_BENCH_CONFIGS=0 BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | line "$lib driver-cmd"; done
for K in 16 24 32; do BIOIK_SOLVE_DENSE_HANDOVER=$K BIOIK_BENCH_CONFIGS=0 BIOIK_HIP_LIBRARY=build/ab/lib_a7.so python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | line "lib_a7 dense handover=$K"; done
for TP in 0 2 4 8; do BIOIK_SOLVE_TWO_PHASE=$TP BIOIK_BENCH_CONFIGS=0 BIOIK_HIP_LIBRARY=build/ab/lib_a7.so python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | line "lib_a7 two-phase=$TP"; done ) 2>&1 | tee $O/bench_ab.log
( time timeout 900 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -5 | tee $O/gpu_suite.log

}

s_r04_11() {
# round 4, GPU session 11: lib_a8 = AvoidJointLimitsGoal's free zone in C4's pre-selection (genes that no child of a generation can take out of it are not generated);
# C4 on the bench line against lib_a7, per-phase cycles, a 3000-case parity soak against the oracle
O=gpurun_out/r04s11; mkdir -p $O
for lib in build/ab/lib_a7.so build/ab/lib_a8.so; do
BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['configs']
print('$lib: value %.0f | C3 %.0f (%.3f, success %.3f) | C4 %.0f (%.3f, success %.4f, %.1f ms)' % (d['value'], c['c3']['value'], c['c3']['roofline']['chip_level_frac'], c['c3']['success_rate'], c['c4']['value'], c['c4']['roofline']['chip_level_frac'], c['c4']['success_rate'], c['c4']['ms_per_step']))" | tee -a $O/bench_ab.log
done
( time python tools/fuzz_parity.py 3000 8844 ) 2>&1 | grep -v " ok$" | tail -6 | tee $O/fuzz.log

}

s_r04_12() {
# round 4, GPU session 12: the profile of record (bench line, rocprofv3 kernel statistics of the same command, PMC passes) + the driver's own bench command
O=gpurun_out/r04s12; mkdir -p $O
export TMPDIR=/tmp
( time bash tools/profile_round.sh r04 ) > $O/profile_round.log 2>&1
tail -3 gpurun_out/bench_r04.err
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json ) 2>&1 | grep real
python - <<'PY'
import json
for f in ('gpurun_out/bench_r04.json', 'gpurun_out/r04s12/bench_driver_cmd.json'):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1]); c=d['configs']
    print('%s: %.0f solves/s %.2f ms chip %.3f | lat3 %.0f | one-at-a-time %.0f | pipelined %.0f | tracking %.0f | ref-params %.0f | cpu %.0f -> %.0fx | C3 %.0f (%.3f) C4 %.0f (%.3f)' % (f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['latency_schedule_three_in_flight']['value'], d['one_batch_at_a_time']['value'], d['host_pointer_pipelined']['value'], d['tracking_seeds']['value'], d['reference_parameters']['value'], d['cpu_baseline']['value'], d['speedup_vs_cpu_1thread'], c['c3']['value'], c['c3']['roofline']['chip_level_frac'], c['c4']['value'], c['c4']['roofline']['chip_level_frac']))
PY

}

s_r04_13() {
# round 4, GPU session 13: k_solve_lean_clj4 (the joint walk of both species' children under 128 registers, 4 wavefronts per SIMD) on C3 against the
# three-wavefront kernel of the same mapping (BIOIK_SOLVE_THREE_WAVES), the GPU parity suite on this tree
O=gpurun_out/r04s13; mkdir -p $O
( time python -m pytest tests -m gpu -x -q ) > $O/gpu_suite.log 2>&1; tail -3 $O/gpu_suite.log
for tw in 0 1 0 1; do
E=""; [ $tw = 1 ] && E="BIOIK_SOLVE_THREE_WAVES=1"
env $E python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['configs']
print('three_waves=$tw: value %.0f | C3 %.0f (%.3f, success %.3f, %.1f ms) | C4 %.0f (%.3f, success %.4f, %.1f ms)' % (d['value'], c['c3']['value'], c['c3']['roofline']['chip_level_frac'], c['c3']['success_rate'], c['c3']['ms_per_step'], c['c4']['value'], c['c4']['roofline']['chip_level_frac'], c['c4']['success_rate'], c['c4']['ms_per_step']))" | tee -a $O/bench_ab.log
done
BIOIK_SOLVE_REPORT=1 python bench.py --no-cpu-baseline --steps 1 --warmup 0 2>&1 | grep 'launch:' | sort | uniq -c | sort -rn | head -12 | tee -a $O/bench_ab.log

}

s_r04_14() {
# round 4, GPU session 14: per-phase shader cycles of C3 under k_solve_lean_clj4 with the pre-selection split into scoring | sort | draw
O=gpurun_out/r04s14; mkdir -p $O
export BIOIK_HIP_LIBRARY=build/ab/libphase.so BIOIK_SOLVE_REPORT=1
( python tools/phase_probe_config.py c3 3072; python tools/phase_probe_config.py c4 2048 ) 2>&1 | grep -v "amdgpu.ids" | tee $O/phases.log

}

s_r04_15() {
# round 4, GPU session 15: the pre-selection's sort in registers (sort_pairs_in_registers) on C3 / C4 against session 13's numbers (C3 44.6e3, C4 72.8e3), GPU suite, phases
O=gpurun_out/r04s15; mkdir -p $O
( time python -m pytest tests -m gpu -x -q ) > $O/gpu_suite.log 2>&1; grep -E "passed|failed" $O/gpu_suite.log
for i in 1 2; do
python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['configs']
print('value %.0f | C3 %.0f (%.3f, success %.3f, %.1f ms) | C4 %.0f (%.3f, success %.4f, %.1f ms)' % (d['value'], c['c3']['value'], c['c3']['roofline']['chip_level_frac'], c['c3']['success_rate'], c['c3']['ms_per_step'], c['c4']['value'], c['c4']['roofline']['chip_level_frac'], c['c4']['success_rate'], c['c4']['ms_per_step']))" | tee -a $O/bench_ab.log
done
export BIOIK_HIP_LIBRARY=build/ab/libphase.so BIOIK_SOLVE_REPORT=1
( python tools/phase_probe_config.py c3 3072; python tools/phase_probe_config.py c4 2048 ) 2>&1 | grep -v "amdgpu.ids" | tee $O/phases.log | grep -E "==|presel|fitness"

}

s_r04_16() {
# round 4, GPU session 16: the critical path of ONE query (nobody else on the chip) per phase, under the latency schedule's kernel and under 256 lanes
O=gpurun_out/r04s16; mkdir -p $O
export BIOIK_HIP_LIBRARY=build/ab/libphase.so BIOIK_SOLVE_REPORT=1
( python tools/phase_probe_config.py c2 1 latency; BIOIK_SOLVE_THREADS=256 python tools/phase_probe_config.py c2 1 latency; python tools/phase_probe_config.py c2 1 throughput; python tools/phase_probe_config.py c2 64 latency ) 2>&1 | grep -v "amdgpu.ids" | tee $O/phases.log

}

s_r04_17() {
# round 4, GPU session 17: the critical path of ONE query under k_solve_lean_cl4 (forced: 128 lanes, computed children in pairs), per phase
O=gpurun_out/r04s17; mkdir -p $O
export BIOIK_HIP_LIBRARY=build/ab/libphase.so BIOIK_SOLVE_REPORT=1
( BIOIK_SOLVE_THREADS=128 BIOIK_SOLVE_COLUMNLESS=2 BIOIK_SOLVE_FOUR_WAVES=1 python tools/phase_probe_config.py c2 1 latency; BIOIK_SOLVE_THREADS=128 BIOIK_SOLVE_COLUMNLESS=2 BIOIK_SOLVE_FOUR_WAVES=1 python tools/phase_probe_config.py c2 256 latency; python tools/phase_probe_config.py c2 4096 latency ) 2>&1 | grep -v "amdgpu.ids" | tee $O/phases.log

}

s_r04_18() {
# round 4, GPU session 18: the throughput schedule's dense kernel with its stragglers handed to the latency mapping after K steps (BIOIK_SOLVE_DENSE_HANDOVER=K), on an
# isolated call (one solve after the other), three and six in flight at the driver's step count; against the latency schedule's own isolated call
O=gpurun_out/r04s18; mkdir -p $O
export TMPDIR=/tmp
run() { timeout 120 python bench.py --timed-only --no-cpu-baseline --schedule $1 --in-flight $2 --steps $3 --warmup 5 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s %.2f ms success %.4f' % (d['value'], d['ms_per_step'], d.get('success_rate', -1)))"; }
echo "latency schedule, in flight 1: $(run latency 1 24)" | tee -a $O/handover_sweep.log
for inf in 1 3 10; do for k in 0 4 6 8 10 12 16 24; do
  echo "throughput schedule, in flight $inf, hand-over after $k: $(BIOIK_SOLVE_DENSE_HANDOVER=$k run throughput $inf $([ $inf = 1 ] && echo 24 || echo 20))"
done; done 2>&1 | tee -a $O/handover_sweep.log

}

s_r04_19() {
# round 4, GPU session 19: the driver's command (20 timed steps) against the number of solves in flight
O=gpurun_out/r04s19; mkdir -p $O
export TMPDIR=/tmp
run() { timeout 120 python bench.py --timed-only --no-cpu-baseline --schedule throughput --in-flight $1 --steps $2 --warmup 5 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s %.2f ms chip %.3f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac']))"; }
for rep in 1 2; do for inf in 4 5 6 8 10 12 16 20; do
  echo "20 steps, in flight $inf: $(run $inf 20)"
done; done 2>&1 | tee -a $O/inflight_sweep.log
for inf in 6 10 20; do echo "60 steps, in flight $inf: $(run $inf 60)"; done 2>&1 | tee -a $O/inflight_sweep.log

}

s_r04_20() {
# round 4, GPU session 20: hand-over of the throughput schedule's stragglers when the chip runs empty (BIOIK_SOLVE_DRAIN_BELOW=N resident workgroups):
# an isolated call, the driver's command (20 steps, 10 and 20 in flight), the steady state (60 steps)
O=gpurun_out/r04s20; mkdir -p $O
export TMPDIR=/tmp

run() { timeout 120 python bench.py --timed-only --no-cpu-baseline --schedule throughput --in-flight $1 --steps $2 --warmup 5 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s %.2f ms chip %.3f success %.4f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d.get('success_rate', -1)))"; }
for n in 0 512 1024 1536 2048; do
  export BIOIK_SOLVE_DRAIN_BELOW=$n
  echo "drain below $n: isolated $(run 1 24) | 20 steps, 10 in flight $(run 10 20) | 20 steps, 20 in flight $(run 20 20) | 60 steps, 10 in flight $(run 10 60)"
done 2>&1 | tee -a $O/drain_sweep.log
for m in 8; do
  export BIOIK_SOLVE_DRAIN_BELOW=1536 BIOIK_SOLVE_DRAIN_MIN_STEPS=$m
  echo "drain below 1536, min steps $m: isolated $(run 1 24) | 20 steps, 10 in flight $(run 10 20) | 20 steps, 20 in flight $(run 20 20)"
done 2>&1 | tee -a $O/drain_sweep.log

}

s_r04_21() {
# round 4, GPU session 21: what the hand-over machinery costs when it never triggers (BIOIK_SOLVE_DRAIN_BELOW=1), steady state
O=gpurun_out/r04s21; mkdir -p $O
export TMPDIR=/tmp
run() { timeout 120 python bench.py --timed-only --no-cpu-baseline --schedule throughput --in-flight $1 --steps $2 --warmup 5 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s %.2f ms chip %.3f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac']))"; }
for rep in 1 2; do for n in 0 1 256; do
  export BIOIK_SOLVE_DRAIN_BELOW=$n
  echo "drain below $n: 60 steps, 10 in flight $(run 10 60) | 60 steps, 3 in flight $(run 3 60)"
done; done 2>&1 | tee -a $O/drain_cost.log

}

s_r04_22() {
# round 4, GPU session 22: the hand-over machinery with one resident word per XCD, read with device scope: never triggering (cost), and the sweep of the threshold
O=gpurun_out/r04s22; mkdir -p $O
export TMPDIR=/tmp
run() { timeout 120 python bench.py --timed-only --no-cpu-baseline --schedule throughput --in-flight $1 --steps $2 --warmup 5 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s %.2f ms chip %.3f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac']))"; }
for rep in 1 2; do for n in 0 1 1024; do
  export BIOIK_SOLVE_DRAIN_BELOW=$n
  echo "drain below $n: isolated $(run 1 24) | 20 steps, 10 in flight $(run 10 20) | 20 steps, 20 in flight $(run 20 20) | 60 steps, 10 in flight $(run 10 60)"
done; done 2>&1 | tee -a $O/drain_sweep.log

}

s_r04_23() {
# round 4, GPU session 23 (experiment): what about the per-step read of the resident word costs a stream of solves its throughput
export TMPDIR=/tmp
run() { timeout 120 python bench.py --timed-only --no-cpu-baseline --schedule throughput --in-flight $1 --steps $2 --warmup 5 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s %.2f ms chip %.3f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac']))"; }
for rep in 1 2; do
  echo "off: $(BIOIK_SOLVE_DRAIN_BELOW=0 run 10 60)"
  echo "never triggers, every step: $(BIOIK_SOLVE_DRAIN_BELOW=1 run 10 60)"
  echo "never triggers, one workgroup in eight reads: $(BIOIK_SOLVE_DRAIN_BELOW=1 BIOIK_SOLVE_DRAIN_MIN_STEPS=-1 run 10 60)"
  echo "never triggers, reads another (quiet) word: $(BIOIK_SOLVE_DRAIN_BELOW=1 BIOIK_SOLVE_DRAIN_MIN_STEPS=-2 run 10 60)"
  echo "never triggers, every fourth step: $(BIOIK_SOLVE_DRAIN_BELOW=1 BIOIK_SOLVE_DRAIN_MIN_STEPS=-3 run 10 60)"
done

}

s_r04_24() {
# round 4, GPU session 24: the latency schedule with the dense kernel first and its stragglers handed to k_solve_lean_cl4 when the chip runs empty (the default now)
# against k_solve_lean_cl4 alone (BIOIK_SOLVE_DRAIN_BELOW=0): an isolated call, three in flight; GPU suite; the bench line
O=gpurun_out/r04s24; mkdir -p $O
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -x -q ) > $O/gpu_suite.log 2>&1; grep -E "passed|failed" $O/gpu_suite.log
run() { timeout 120 python bench.py --timed-only --no-cpu-baseline --schedule latency --in-flight $1 --steps $2 --warmup 5 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s %.2f ms' % (d['value'], d['ms_per_step']))"; }
for rep in 1 2; do for n in 0 512 1024 1536; do
  export BIOIK_SOLVE_DRAIN_BELOW=$n
  echo "latency schedule, drain below $n: isolated $(run 1 24) | three in flight $(run 3 24)"
done; done 2>&1 | tee -a $O/latency_drain.log
unset BIOIK_SOLVE_DRAIN_BELOW
python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['configs']
print('driver cmd: %.0f solves/s %.2f ms chip %.3f | lat3 %.0f | one-at-a-time %.0f | pipelined %.0f | tracking %.0f | ref-params %.0f | C3 %.0f (%.3f) C4 %.0f (%.3f)' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['latency_schedule_three_in_flight']['value'], d['one_batch_at_a_time']['value'], d['host_pointer_pipelined']['value'], d['tracking_seeds']['value'], d['reference_parameters']['value'], c['c3']['value'], c['c3']['roofline']['chip_level_frac'], c['c4']['value'], c['c4']['roofline']['chip_level_frac']))" | tee -a $O/latency_drain.log

}

s_r04_25() {
# round 4, GPU session 25: small batches under the latency schedule, one call at a time: the launcher's choice (<= 768 units: 256 lanes, children in columns)
# against k_solve_lean_cl4 forced (128 lanes, computed children in pairs) and the dense kernel (throughput schedule)
O=gpurun_out/r04s25; mkdir -p $O
export TMPDIR=/tmp
run() { BIOIK_BENCH_BATCH=$1 timeout 120 python bench.py --timed-only --in-flight 1 --no-cpu-baseline --schedule $2 --steps 12 --warmup 3 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s %.3f ms' % (d['value'], d['ms_per_step']))"; }
for b in 1 16 64 256 512 768 1024 1536 2048; do
  echo "batch $b: launcher $(run $b latency) | cl4 forced $(BIOIK_SOLVE_THREADS=128 BIOIK_SOLVE_COLUMNLESS=2 BIOIK_SOLVE_FOUR_WAVES=1 run $b latency) | dense $(run $b throughput)"
done 2>&1 | tee $O/small_batches.log

}

s_r04_26() {
# round 4, GPU session 26: the resident word prefetched into LDS at the start of a step: the hand-over's bookkeeping on a stream of solves (never triggering: threshold 1),
# the throughput schedule with the hand-over (BIOIK_SOLVE_DRAIN_THROUGHPUT=1), GPU suite
O=gpurun_out/r04s26; mkdir -p $O
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -x -q ) > $O/gpu_suite.log 2>&1; grep -E "passed|failed" $O/gpu_suite.log
run() { timeout 120 python bench.py --timed-only --no-cpu-baseline --schedule $3 --in-flight $1 --steps $2 --warmup 5 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s %.2f ms chip %.3f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac']))"; }
for rep in 1 2 3; do
  echo "throughput, off: 60/10 $(run 10 60 throughput) | 20/10 $(run 10 20 throughput) | 20/20 $(run 20 20 throughput) | isolated $(run 1 24 throughput)"
  echo "throughput, on, never triggers: 60/10 $(BIOIK_SOLVE_DRAIN_THROUGHPUT=1 BIOIK_SOLVE_DRAIN_BELOW=1 run 10 60 throughput)"
  echo "throughput, on, below 1024: 60/10 $(BIOIK_SOLVE_DRAIN_THROUGHPUT=1 run 10 60 throughput) | 20/10 $(BIOIK_SOLVE_DRAIN_THROUGHPUT=1 run 10 20 throughput) | 20/20 $(BIOIK_SOLVE_DRAIN_THROUGHPUT=1 run 20 20 throughput) | isolated $(BIOIK_SOLVE_DRAIN_THROUGHPUT=1 run 1 24 throughput)"
done 2>&1 | tee $O/drain_prefetch.log
echo "latency: isolated $(run 1 24 latency) | three in flight $(run 3 24 latency)" | tee -a $O/drain_prefetch.log

}

s_r04_29() {
# round 4, GPU session 29: the profile of record on the final kernels (bench line, rocprofv3 kernel statistics of the same command, PMC passes) + the driver's own bench command
O=gpurun_out/r04s29; mkdir -p $O
export TMPDIR=/tmp
( time bash tools/profile_round.sh r04 ) > $O/profile_round.log 2>&1
tail -3 gpurun_out/bench_r04.err
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json ) 2>&1 | grep real
python - <<'PY'
import json
for f in ('gpurun_out/bench_r04.json', 'gpurun_out/r04s29/bench_driver_cmd.json'):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1]); c=d['configs']
    print('%s: %.0f solves/s %.2f ms chip %.3f | lat3 %.0f | one-at-a-time %.0f | pipelined %.0f | tracking %.0f | ref-params %.0f | cpu %.0f -> %.0fx | C3 %.0f (%.3f) C4 %.0f (%.3f)' % (f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['latency_schedule_three_in_flight']['value'], d['one_batch_at_a_time']['value'], d['host_pointer_pipelined']['value'], d['tracking_seeds']['value'], d['reference_parameters']['value'], d['cpu_baseline']['value'], d['speedup_vs_cpu_1thread'], c['c3']['value'], c['c3']['roofline']['chip_level_frac'], c['c4']['value'], c['c4']['roofline']['chip_level_frac']))
PY

}

s_r04_30() {
# round 4, GPU session 30: the bench line with 32 hardware queues for its 20 + 6 streams (24 left the host-pointer leg and C3 / C4 sharing queues), 24 beside it
O=gpurun_out/r04s30; mkdir -p $O
export TMPDIR=/tmp
for q in 32 24 32; do
GPU_MAX_HW_QUEUES=$q python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['configs']
print('queues $q: %.0f solves/s %.2f ms chip %.3f | lat3 %.0f | one-at-a-time %.0f | pipelined %.0f | tracking %.0f | ref-params %.0f | C3 %.0f (%.3f) C4 %.0f (%.3f)' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['latency_schedule_three_in_flight']['value'], d['one_batch_at_a_time']['value'], d['host_pointer_pipelined']['value'], d['tracking_seeds']['value'], d['reference_parameters']['value'], c['c3']['value'], c['c3']['roofline']['chip_level_frac'], c['c4']['value'], c['c4']['roofline']['chip_level_frac']))" | tee -a $O/queues.log
done

}

s_r04_31() {
# round 4, GPU session 31: C3 / C4 on the bench line in ONE session: the library of commit 11d6e3b (before the register sort and the hand-over bookkeeping) against the
# present one, with ten streams in the process and with twenty (of which C3 / C4 use ten)
O=gpurun_out/r04s31; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do for lib in build/ab/lib_clj4.so bio_ik_amd/libbioik_hip.so; do for inf in 10 20; do
BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 20 --warmup 5 --in-flight $inf 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['configs']
print('$lib, $inf in flight: value %.0f (%.3f) | tracking %.0f | ref-params %.0f | C3 %.0f (%.3f) | C4 %.0f (%.3f)' % (d['value'], d['roofline']['chip_level_frac'], d['tracking_seeds']['value'], d['reference_parameters']['value'], c['c3']['value'], c['c3']['roofline']['chip_level_frac'], c['c4']['value'], c['c4']['roofline']['chip_level_frac']))" | tee -a $O/ab.log
done; done; done

}

s_r04_32() {
# round 4, GPU session 32: the line search's frame sets on the idle elite buffer (C3: 10520 -> 9496 B of LDS per query, sixteen instead of fifteen per CU) against the
# library before it (build/ab/lib_b1.so), same session; GPU suite
O=gpurun_out/r04s32; mkdir -p $O
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -x -q ) > $O/gpu_suite.log 2>&1; grep -E "passed|failed" $O/gpu_suite.log
for rep in 1 2; do for lib in build/ab/lib_b1.so bio_ik_amd/libbioik_hip.so; do
BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['configs']
print('$lib: value %.0f (%.3f) | lat3 %.0f | one-at-a-time %.0f | tracking %.0f | ref-params %.0f | C3 %.0f (%.3f) | C4 %.0f (%.3f)' % (d['value'], d['roofline']['chip_level_frac'], d['latency_schedule_three_in_flight']['value'], d['one_batch_at_a_time']['value'], d['tracking_seeds']['value'], d['reference_parameters']['value'], c['c3']['value'], c['c3']['roofline']['chip_level_frac'], c['c4']['value'], c['c4']['roofline']['chip_level_frac']))" | tee -a $O/ab.log
done; done
BIOIK_SOLVE_REPORT=1 python bench.py --no-cpu-baseline --steps 1 --warmup 0 2>&1 | grep 'solve:' | sort | uniq -c | sort -rn | head -8 | cut -c1-330 | tee -a $O/ab.log

}

s_r04_33() {
# round 4, GPU session 34: the joint-walk kernel with the walk for serial segments (lib_b3: the general walk), same session

O=gpurun_out/r04s34; mkdir -p $O
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -x -q ) > $O/gpu_suite.log 2>&1; grep -E "passed|failed" $O/gpu_suite.log
for rep in 1 2; do for lib in build/ab/lib_b3.so bio_ik_amd/libbioik_hip.so; do
BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['configs']
print('$lib: value %.0f (%.3f) | lat3 %.0f | one-at-a-time %.0f | tracking %.0f | ref-params %.0f | C3 %.0f (%.3f) | C4 %.0f (%.3f)' % (d['value'], d['roofline']['chip_level_frac'], d['latency_schedule_three_in_flight']['value'], d['one_batch_at_a_time']['value'], d['tracking_seeds']['value'], d['reference_parameters']['value'], c['c3']['value'], c['c3']['roofline']['chip_level_frac'], c['c4']['value'], c['c4']['roofline']['chip_level_frac']))" | tee -a $O/ab.log
done; done
BIOIK_SOLVE_REPORT=1 python bench.py --no-cpu-baseline --steps 1 --warmup 0 2>&1 | grep 'solve:' | sort | uniq -c | sort -rn | head -8 | cut -c1-330 | tee -a $O/ab.log

}

s_r04_35() {
# round 4, GPU session 35: the final tree: GPU suite, smoke(), a parity soak, then the profile of record (bench line, rocprofv3 kernel statistics of the same command,
# PMC passes) and the driver's own bench command
O=gpurun_out/r04s35; mkdir -p $O
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q ) > $O/gpu_suite.log 2>&1; grep -E "passed|failed" $O/gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee $O/smoke.log
( time python tools/fuzz_parity.py 1500 9911 ) 2>&1 | grep -v " ok$" | tail -4 | tee $O/fuzz.log
( time bash tools/profile_round.sh r04 ) > $O/profile_round.log 2>&1
tail -3 gpurun_out/bench_r04.err
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json ) 2>&1 | grep real
python - <<'PY'
import json
for f in ('gpurun_out/bench_r04.json', 'gpurun_out/r04s35/bench_driver_cmd.json'):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1]); c=d['configs']
    print('%s: %.0f solves/s %.2f ms chip %.3f | lat3 %.0f | one-at-a-time %.0f | pipelined %.0f | tracking %.0f | ref-params %.0f | cpu %.0f -> %.0fx | C3 %.0f (%.3f) C4 %.0f (%.3f)' % (f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['latency_schedule_three_in_flight']['value'], d['one_batch_at_a_time']['value'], d['host_pointer_pipelined']['value'], d['tracking_seeds']['value'], d['reference_parameters']['value'], d['cpu_baseline']['value'], d['speedup_vs_cpu_1thread'], c['c3']['value'], c['c3']['roofline']['chip_level_frac'], c['c4']['value'], c['c4']['roofline']['chip_level_frac']))
PY

}

s_r04_36() {
# round 4, GPU session 36: the final tree (after the exact evaluation counts of the C3 / C4 legs and the M0-preserving prefetch): GPU suite, smoke(), a parity soak, then the profile of record (bench line, rocprofv3 kernel statistics of the same command,
# PMC passes) and the driver's own bench command
O=gpurun_out/r04s36; mkdir -p $O
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q ) > $O/gpu_suite.log 2>&1; grep -E "passed|failed" $O/gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee $O/smoke.log
( time python tools/fuzz_parity.py 3000 4477 ) 2>&1 | grep -v " ok$" | tail -6 | tee $O/fuzz.log
( time bash tools/profile_round.sh r04 ) > $O/profile_round.log 2>&1
tail -3 gpurun_out/bench_r04.err
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json ) 2>&1 | grep real
python - <<'PY'
import json
for f in ('gpurun_out/bench_r04.json', 'gpurun_out/r04s36/bench_driver_cmd.json'):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1]); c=d['configs']
    print('%s: %.0f solves/s %.2f ms chip %.3f | lat3 %.0f | one-at-a-time %.0f | pipelined %.0f | tracking %.0f | ref-params %.0f | cpu %.0f -> %.0fx | C3 %.0f (%.3f) C4 %.0f (%.3f)' % (f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['latency_schedule_three_in_flight']['value'], d['one_batch_at_a_time']['value'], d['host_pointer_pipelined']['value'], d['tracking_seeds']['value'], d['reference_parameters']['value'], d['cpu_baseline']['value'], d['speedup_vs_cpu_1thread'], c['c3']['value'], c['c3']['roofline']['chip_level_frac'], c['c4']['value'], c['c4']['roofline']['chip_level_frac']))
PY

}

s_r04_37() {
# round 4, GPU session 37: the pre-selection sorted by 64-bit keys (one compare, two dwords per exchange; exact sort behind a check) against the (fitness, index) pairs
# (build/ab/lib_b4.so), same session; GPU suite
O=gpurun_out/r04s37; mkdir -p $O
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -x -q ) > $O/gpu_suite.log 2>&1; grep -E "passed|failed" $O/gpu_suite.log
for rep in 1 2; do for lib in build/ab/lib_b4.so bio_ik_amd/libbioik_hip.so; do
BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['configs']
print('$lib: value %.0f (%.3f) | C3 %.0f (%.3f) | C4 %.0f (%.3f)' % (d['value'], d['roofline']['chip_level_frac'], c['c3']['value'], c['c3']['roofline']['chip_level_frac'], c['c4']['value'], c['c4']['roofline']['chip_level_frac']))" | tee -a $O/ab.log
done; done

}

s_r04_38() {
# round 4, GPU session 38: the two best children of a generation from keys in the half-wavefront kernels (dense, joint walk) against the merging butterfly (lib_b5)

O=gpurun_out/r04s38; mkdir -p $O
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -x -q ) > $O/gpu_suite.log 2>&1; grep -E "passed|failed" $O/gpu_suite.log
for rep in 1 2; do for lib in build/ab/lib_b5.so bio_ik_amd/libbioik_hip.so; do
BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['configs']
print('$lib: value %.0f (%.3f) | C3 %.0f (%.3f) | C4 %.0f (%.3f)' % (d['value'], d['roofline']['chip_level_frac'], c['c3']['value'], c['c3']['roofline']['chip_level_frac'], c['c4']['value'], c['c4']['roofline']['chip_level_frac']))" | tee -a $O/ab.log
done; done

}

s_r04_39() {
# round 4, GPU session 39: keyed top-2 (present library) against the merging butterfly (build/ab/lib_b5.so) on the headline: 60 timed steps, ten in flight, alternating
O=gpurun_out/r04s39; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2 3 4; do for lib in build/ab/lib_b5.so bio_ik_amd/libbioik_hip.so; do
echo "$lib: $(BIOIK_HIP_LIBRARY=$lib python bench.py --timed-only --no-cpu-baseline --steps 60 --warmup 5 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s %.3f ms chip %.4f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac']))")" | tee -a $O/ab.log
done; done

}

s_r04_40() {
# round 4, GPU session 40: the final tree (64-bit keys in the pre-selection sort and the top-2 selection): GPU suite, smoke(), a parity soak, then the profile of record (bench line, rocprofv3 kernel statistics of the same command,
# PMC passes) and the driver's own bench command
O=gpurun_out/r04s40; mkdir -p $O
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q ) > $O/gpu_suite.log 2>&1; grep -E "passed|failed" $O/gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee $O/smoke.log
( time python tools/fuzz_parity.py 4000 7731 ) 2>&1 | grep -v " ok$" | tail -6 | tee $O/fuzz.log
( time bash tools/profile_round.sh r04 ) > $O/profile_round.log 2>&1
tail -3 gpurun_out/bench_r04.err
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json ) 2>&1 | grep real
python - <<'PY'
import json
for f in ('gpurun_out/bench_r04.json', 'gpurun_out/r04s40/bench_driver_cmd.json'):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1]); c=d['configs']
    print('%s: %.0f solves/s %.2f ms chip %.3f | lat3 %.0f | one-at-a-time %.0f | pipelined %.0f | tracking %.0f | ref-params %.0f | cpu %.0f -> %.0fx | C3 %.0f (%.3f) C4 %.0f (%.3f)' % (f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['latency_schedule_three_in_flight']['value'], d['one_batch_at_a_time']['value'], d['host_pointer_pipelined']['value'], d['tracking_seeds']['value'], d['reference_parameters']['value'], d['cpu_baseline']['value'], d['speedup_vs_cpu_1thread'], c['c3']['value'], c['c3']['roofline']['chip_level_frac'], c['c4']['value'], c['c4']['roofline']['chip_level_frac']))
PY

}

s_r04_41() {
# round 4, GPU session 41: the final tree (persistent scratch, one-launch mapping on capturing streams): GPU suite, smoke(), a parity soak, then the profile of record (bench line, rocprofv3 kernel statistics of the same command,
# PMC passes) and the driver's own bench command
O=gpurun_out/r04s41; mkdir -p $O
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q ) > $O/gpu_suite.log 2>&1; grep -E "passed|failed" $O/gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee $O/smoke.log
( time python tools/fuzz_parity.py 2000 1357 ) 2>&1 | grep -v " ok$" | tail -6 | tee $O/fuzz.log
( time bash tools/profile_round.sh r04 ) > $O/profile_round.log 2>&1
tail -3 gpurun_out/bench_r04.err
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json ) 2>&1 | grep real
python - <<'PY'
import json
for f in ('gpurun_out/bench_r04.json', 'gpurun_out/r04s41/bench_driver_cmd.json'):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1]); c=d['configs']
    print('%s: %.0f solves/s %.2f ms chip %.3f | lat3 %.0f | one-at-a-time %.0f | pipelined %.0f | tracking %.0f | ref-params %.0f | cpu %.0f -> %.0fx | C3 %.0f (%.3f) C4 %.0f (%.3f)' % (f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['latency_schedule_three_in_flight']['value'], d['one_batch_at_a_time']['value'], d['host_pointer_pipelined']['value'], d['tracking_seeds']['value'], d['reference_parameters']['value'], d['cpu_baseline']['value'], d['speedup_vs_cpu_1thread'], c['c3']['value'], c['c3']['roofline']['chip_level_frac'], c['c4']['value'], c['c4']['roofline']['chip_level_frac']))
PY

}

s_r04_42() {
# round 4, GPU session 42: after the host-side change of the scratch (stream-ordered, persistent): the two GPU tests it touches, and the HBM-traffic passes again so that
# profiles/traffic.json carries the hash of the final sources (device code unchanged since session 41, whose other passes stay)
R=$(pwd); O=$R/gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "pipelining or hipgraph or stragglers or islands" 2>&1 | tail -2
cd /tmp
pmc() { d=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$d -o $d -- python $R/bench.py --timed-only --steps 5 --warmup 1 --in-flight 1 > $O/pmc_$d.log 2>&1; }
rm -rf $O/pmc_fetch $O/pmc_write
BIOIK_BENCH_STREAM=0 pmc fetch FETCH_SIZE
BIOIK_BENCH_STREAM=0 pmc write WRITE_SIZE
ls $O/pmc_fetch $O/pmc_write | head

}

if [ -z "$1" ]; then declare -F | sed -n "s/^declare -f s_//p"; exit 0; fi
"s_$1"
