#!/bin/bash
# lease 30: the lazy SEG check with its residues in registers (k_mem_post1, k_trigcheck); two contexts in flight once more
O=$GRAFT_REPO_ROOT/gpurun_out/r06_l30; mkdir -p $O
show() { python - $1 $2 <<'PY'
import json,re,sys
t=open(sys.argv[1]).read()
m=re.search(r'\[bench\] detail: (\{.*\})',t)
if not m: print(sys.argv[2],'no detail line'); sys.exit(0)
d=json.loads(m.group(1))
print(sys.argv[2], 'headline', round(d['value']/1e6,2), {k:round(v,2) for k,v in d['roofline']['stage_ms_per_step_exclusive'].items()}, 'ctx', d['config'].get('contexts_in_flight'), 'chunk', d['config'].get('chunk'))
for k in ('greedy','paired','hard','hard_greedy','wide','wide_greedy','long','protein','host_buffers'):
    if k in d and 'roofline' in d[k]: print(' ', k, round(d[k]['value']/1e6,2), {a:round(b,2) for a,b in d[k]['roofline']['stage_ms_per_step_exclusive'].items()})
    elif k in d: print(' ', k, round(d[k]['value']/1e6,2))
print('  parity', d.get('parity_checked_reads'), d.get('mismatches'))
PY
}
timeout 1500 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "[l30] bench rc=$?"; show $O/bench_n1.err line
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --legs "" --steps 5 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err )
cp $O/stats/s_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null; rm -rf $O/stats; grep "k_mem_post\|k_seg\|k_mem_second\|k_mem(" $O/kernel_stats.csv | cut -d, -f1-4 | sed 's/(.*)"/"/' | cut -c1-120
( KAIJU_GPU_FUSED_POST=0 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or fused or fullsize or long" ) > $O/gpu_tests_unfused.log 2>&1; echo "[l30] unfused tests rc=$?"; tail -2 $O/gpu_tests_unfused.log
timeout 600 python bench.py --legs "" --steps 5 --no-cpu-baseline --contexts 2 --chunk 5000000 > $O/bench_mem_c2.json 2> $O/bench_mem_c2.err; show $O/bench_mem_c2.err mem_ctx2
timeout 600 python bench.py --mode greedy --legs "" --steps 3 --no-cpu-baseline > $O/bench_greedy_c1.json 2> $O/bench_greedy_c1.err; show $O/bench_greedy_c1.err greedy_ctx1
timeout 600 python bench.py --mode greedy --legs "" --steps 3 --no-cpu-baseline --contexts 2 --chunk 5000000 > $O/bench_greedy_c2.json 2> $O/bench_greedy_c2.err; show $O/bench_greedy_c2.err greedy_ctx2
timeout 600 python bench.py --mode greedy --legs "" --steps 3 --no-cpu-baseline --contexts 2 --chunk 2500000 > $O/bench_greedy_c2b.json 2> $O/bench_greedy_c2b.err; show $O/bench_greedy_c2b.err greedy_ctx2_2.5M
