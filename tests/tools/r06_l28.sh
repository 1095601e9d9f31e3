#!/bin/bash
# lease 28: k_seg without 64-bit products; the command line's batch size; the chain test in the wide Greedy lane (variant widecp)
O=$GRAFT_REPO_ROOT/gpurun_out/r06_l28; mkdir -p $O
V=$GRAFT_REPO_ROOT/kaiju_amd/variants
show() { python - $1 $2 <<'PY'
import json,re,sys
t=open(sys.argv[1]).read()
m=re.search(r'\[bench\] detail: (\{.*\})',t)
if not m: print(sys.argv[2],'no detail line'); sys.exit(0)
d=json.loads(m.group(1))
print(sys.argv[2], 'headline', round(d['value']/1e6,2), {k:round(v,2) for k,v in d['roofline']['stage_ms_per_step_exclusive'].items()})
for k in ('greedy','wide','wide_greedy','hard','hard_greedy'):
    if k in d and 'roofline' in d[k]: print(' ', k, round(d[k]['value']/1e6,2), {a:round(b,2) for a,b in d[k]['roofline']['stage_ms_per_step_exclusive'].items()}, 'pruned', round(d[k]['roofline']['ops_per_unit'].get('pruned_chains',0),2), 'items', round(d[k]['roofline']['ops_per_unit'].get('items_read',0),2))
print('  parity', d.get('parity_checked_reads'), d.get('mismatches'))
PY
}
timeout 900 python bench.py --legs greedy --steps 5 --no-cpu-baseline > $O/bench_seg.json 2> $O/bench_seg.err; show $O/bench_seg.err seg
# (a batch-size experiment of the command line ran here: inconclusive - runs of 96 M reads into a file in /dev/shm scatter between 17 and 38 M reads/s whatever the batch size - and removed)
timeout 900 python bench.py --legs wide --steps 3 --reads 2000000 > $O/bench_wide_cur.json 2> $O/bench_wide_cur.err; show $O/bench_wide_cur.err wide_cur
KAIJU_GPU_LIB=$V/libkaiju_gpu_widecp.so timeout 900 python bench.py --legs wide --steps 3 --reads 2000000 > $O/bench_wide_widecp.json 2> $O/bench_wide_widecp.err; show $O/bench_wide_widecp.err wide_widecp
W=/dev/shm/kaiju_hardwide; mkdir -p $W
ARGS="--work $W --reads 500000 --steps 3 --warmup 1 --legs hard --hard-nseq 15900001 --hard-reads 2000000 --leg-steps 3 --cpu-sample 100000 --cpu-sample-legs 200000"
KAIJU_GPU_LIB=$V/libkaiju_gpu_widecp.so timeout 2100 python bench.py $ARGS > $O/bench_hardwide_widecp.json 2> $O/bench_hardwide_widecp.err; show $O/bench_hardwide_widecp.err hardwide_widecp
timeout 900 python bench.py $ARGS --no-cpu-baseline > $O/bench_hardwide_cur.json 2> $O/bench_hardwide_cur.err; show $O/bench_hardwide_cur.err hardwide_cur
rm -rf $W
