#!/bin/bash
# lease 27: s_Trim from prefix counts (k_seg) - the driver's line, the SEG / golden / Greedy tests, a fuzz run
O=$GRAFT_REPO_ROOT/gpurun_out/r06_l27; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "seg or golden or greedy or fullsize" ) > $O/gpu_tests_subset.log 2>&1; echo "[l27] subset rc=$?"; tail -3 $O/gpu_tests_subset.log
timeout 1500 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "[l27] bench rc=$?"
python - <<'PY'
import json,re
t=open('gpurun_out/r06_l27/bench_n1.err').read()
m=re.search(r'\[bench\] detail: (\{.*\})',t)
d=json.loads(m.group(1)); json.dump(d,open('gpurun_out/r06_l27/bench_detail_n1.json','w'),indent=1)
print('headline',round(d['value']/1e6,1),{k:round(v,2) for k,v in d['roofline']['stage_ms_per_step_exclusive'].items()})
for k in ('greedy','paired','hard','hard_greedy','wide','wide_greedy','long','protein','host_buffers'):
    if k in d and 'roofline' in d[k]: print(k, round(d[k]['value']/1e6,1), {a:round(b,2) for a,b in d[k]['roofline']['stage_ms_per_step_exclusive'].items()})
    elif k in d: print(k, round(d[k]['value']/1e6,1))
print(d.get('parity_checked_reads'), d.get('mismatches'))
PY
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --legs "" --steps 5 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err )
cp $O/stats/s_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null; rm -rf $O/stats; grep "k_seg\|k_mem_post1\|k_mem(" $O/kernel_stats.csv | cut -c1-140
timeout 900 python tests/tools/fuzz_gpu.py 30 91 > $O/fuzz_gpu.log 2>&1; echo "[l27] fuzz rc=$?"; tail -2 $O/fuzz_gpu.log
KAIJU_GPU_FORCE_WIDE=16 timeout 900 python tests/tools/fuzz_gpu.py 20 92 > $O/fuzz_gpu_wide.log 2>&1; echo "[l27] fuzz wide rc=$?"; tail -2 $O/fuzz_gpu_wide.log
