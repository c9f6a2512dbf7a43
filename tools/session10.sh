#!/bin/bash
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
