#!/bin/bash
# lease 35: PMC traffic of the search kernels of the final library (passes 1 and 2, the launch sizes of the default line);
# the family-structured 4.5 G-row index once more (chain test on up to four rows in the wide lane)
O=$GRAFT_REPO_ROOT/gpurun_out/r06_l35; mkdir -p $O
PMC_PASSES="1 2" PMC_BENCH_ARGS=" " bash tests/tools/pmc_bench.sh $O/pmc > $O/pmc.log 2>&1; tail -2 $O/pmc.log
PMC_MEM_LAUNCH=5000000 PMC_PAIR_LAUNCH=2500000 PMC_MERGE=profiles/traffic.json python tests/tools/pmc_bench_collect.py $O/pmc $O/traffic.json profiles/r06_pmc_halves > $O/pmc_collect.log 2>&1; tail -4 $O/pmc_collect.log
W=/dev/shm/kaiju_hardwide; mkdir -p $W
ARGS="--work $W --reads 500000 --steps 3 --warmup 1 --legs hard --hard-nseq 15900001 --hard-reads 2000000 --leg-steps 3 --cpu-sample 100000 --cpu-sample-legs 200000"
KAIJU_GPU_LOAD_TIMES=1 timeout 2100 python bench.py $ARGS > $O/bench_hardwide.json 2> $O/bench_hardwide.err; echo "[l35] hardwide rc=$?"
python - <<'PY'
import json,re
t=open('gpurun_out/r06_l35/bench_hardwide.err').read()
d=json.loads(re.search(r'\[bench\] detail: (\{.*\})',t).group(1)); json.dump(d,open('gpurun_out/r06_l35/bench_hardwide_detail.json','w'),indent=1)
for k in ('hard','hard_greedy'):
    r=d[k]['roofline']; o=r['ops_per_unit']
    print(k, round(d[k]['value']/1e6,2), {a:round(b,2) for a,b in r['stage_ms_per_step_exclusive'].items()}, 'pruned', round(o.get('pruned_chains',0),2), 'items', round(o.get('items_read',0),2))
print('parity', d.get('parity_checked_reads'), d.get('mismatches'))
PY
rm -rf $W
