#!/bin/bash
# lease 34: the chain test for variants on up to four rows (kChainRows)
O=$GRAFT_REPO_ROOT/gpurun_out/r06_l34; mkdir -p $O
timeout 1500 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "[l34] bench rc=$?"
python - <<'PY'
import json,re
t=open('gpurun_out/r06_l34/bench_n1.err').read()
d=json.loads(re.search(r'\[bench\] detail: (\{.*\})',t).group(1)); json.dump(d,open('gpurun_out/r06_l34/bench_detail_n1.json','w'),indent=1)
print('headline',round(d['value']/1e6,1))
for k in ('greedy','hard_greedy','wide_greedy'):
    r=d[k]['roofline']; o=r['ops_per_unit']
    print(k, round(d[k]['value']/1e6,2), {a:round(b,2) for a,b in r['stage_ms_per_step_exclusive'].items()}, 'pruned', round(o.get('pruned_chains',0),2), 'items', round(o.get('items_read',0),2), 'lane_iters', round(o.get('lane_iterations',0),1), 'frac', round(r['frac'],3))
print('parity', d.get('parity_checked_reads'), d.get('mismatches'))
PY
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "greedy or golden or wide or fullsize" ) > $O/gpu_tests_subset.log 2>&1; echo "[l34] subset rc=$?"; tail -2 $O/gpu_tests_subset.log
timeout 900 python tests/tools/fuzz_gpu.py 40 111 > $O/fuzz_gpu.log 2>&1; echo "[l34] fuzz rc=$?"; tail -1 $O/fuzz_gpu.log
KAIJU_GPU_FORCE_WIDE=16 timeout 900 python tests/tools/fuzz_gpu.py 25 112 > $O/fuzz_gpu_wide.log 2>&1; echo "[l34] fuzz wide rc=$?"; tail -1 $O/fuzz_gpu_wide.log
