#!/bin/bash
# lease 24: trigger scan from registers (stage 1 of the eager SEG flow), row -> taxon table for indexes with 64-bit rows
O=$GRAFT_REPO_ROOT/gpurun_out/r06_l24; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "wide or text_positions or stage1 or fragments or seg or greedy or long" ) > $O/gpu_tests_subset.log 2>&1; echo "[l24] subset rc=$?"; tail -3 $O/gpu_tests_subset.log
timeout 1500 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "[l24] bench rc=$?"
python - <<'PY'
import json,re
t=open('gpurun_out/r06_l24/bench_n1.err').read()
m=re.search(r'\[bench\] detail: (\{.*\})',t)
d=json.loads(m.group(1)); json.dump(d,open('gpurun_out/r06_l24/bench_detail.json','w'),indent=1)
print('headline',d['value']/1e6,d['stage_ms_per_step_exclusive'])
for k in ('greedy','paired','hard','hard_greedy','wide','wide_greedy','long','protein','host_buffers'):
    if k in d: print(k, round(d[k]['value']/1e6,1), d[k].get('stage_ms_per_step_exclusive'))
print(d.get('parity_checked_reads'), d.get('mismatches'))
PY
# the family-structured database at 4.5 G rows again (lease 23), now with the table
W=/dev/shm/kaiju_hardwide; mkdir -p $W
ARGS="--work $W --reads 500000 --steps 3 --warmup 1 --legs hard --hard-nseq 15900001 --hard-reads 2000000 --leg-steps 3 --cpu-sample 100000 --cpu-sample-legs 200000"
KAIJU_GPU_LOAD_TIMES=1 timeout 2100 python bench.py $ARGS > $O/bench_hardwide.json 2> $O/bench_hardwide.err
echo "[l24] hardwide rc=$?"; grep "leg hard\|HBM:\|row -> taxon\|hard database" $O/bench_hardwide.err | cut -c1-400
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $GRAFT_REPO_ROOT/bench.py $ARGS --no-cpu-baseline --leg-steps 2 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err )
cp $O/stats/s_kernel_stats.csv $O/kernel_stats_hardwide.csv 2>/dev/null; rm -rf $O/stats; head -8 $O/kernel_stats_hardwide.csv | cut -c1-200
rm -rf $W
