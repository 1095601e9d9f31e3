#!/bin/bash
# lease 37: stage 1 queues its SEG work with one atomic per wavefront and step (append_slot_wave)
O=$GRAFT_REPO_ROOT/gpurun_out/r06_l37; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "seg or golden or greedy or fullsize or long or paired" ) > $O/gpu_tests_subset.log 2>&1; echo "[l37] subset rc=$?"; tail -2 $O/gpu_tests_subset.log
timeout 1500 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "[l37] bench rc=$?"
python - <<'PY'
import json,re
t=open('gpurun_out/r06_l37/bench_n1.err').read()
d=json.loads(re.search(r'\[bench\] detail: (\{.*\})',t).group(1)); json.dump(d,open('gpurun_out/r06_l37/bench_detail_n1.json','w'),indent=1)
print('headline',round(d['value']/1e6,1), {k:round(v,2) for k,v in d['roofline']['stage_ms_per_step_exclusive'].items()})
for k in ('greedy','hard_greedy','wide_greedy','protein'):
    r=d[k]['roofline']; print(k, round(d[k]['value']/1e6,2), {a:round(b,2) for a,b in r['stage_ms_per_step_exclusive'].items()})
print('parity', d.get('parity_checked_reads'), d.get('mismatches'))
PY
