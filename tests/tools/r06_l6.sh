#!/bin/bash
# round 6, lease 6: (a) the locate of wide indexes (k_mem_locate_wide) - team size and rows per lane, on the `wide` leg (4.39 G rows,
# 23 rows a match); (b) PMC traffic of the legs hard / wide / long / protein (the run of lease 5 lost its output files)
O=$1
V=kaiju_amd/variants
for v in cur loc16 loc32 ilp2 ilp2_16 ilp2_4 cur; do
  tag=${v}_$RANDOM
  KAIJU_GPU_LIB=$PWD/$V/libkaiju_gpu_$v.so timeout 900 python bench.py --reads 100000 --contexts 1 --steps 1 --warmup 0 --leg-steps 4 --no-cpu-baseline --no-ref-ops --legs wide > $O/wide_$tag.json 2> $O/wide_$tag.err
  python - <<P
import json
try:
    d=json.load(open("gpurun_out/bench_detail_n1.json"))
    for nm in ("wide","wide_greedy"):
        r=d[nm]; print("$v", nm, round(r["value"]/1e6,1), "M reads/s", {k:round(x,2) for k,x in r["roofline"]["stage_ms_per_step_exclusive"].items()})
except Exception as e: print("$v failed", e)
P
done
# kernel times of the winner candidates under rocprof
for v in cur ilp2; do
( cd /tmp && export TMPDIR=/tmp && KAIJU_GPU_LIB=$GRAFT_REPO_ROOT/$V/libkaiju_gpu_$v.so timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/st_$v -o s -- python $GRAFT_REPO_ROOT/bench.py --reads 100000 --contexts 1 --steps 1 --warmup 0 --leg-steps 4 --no-cpu-baseline --no-ref-ops --legs wide > /dev/null 2> $GRAFT_REPO_ROOT/$O/st_$v.err )
cp $O/st_$v/s_kernel_stats.csv $O/kernel_stats_wide_$v.csv 2>/dev/null; rm -rf $O/st_$v; echo "== $v"; grep "locate_wide\|k_mem_wide2\|k_lca\|k_seg" $O/kernel_stats_wide_$v.csv | cut -c1-200
done
bash tests/tools/pmc_legs.sh $O/pmc_legs hard wide long protein > $O/pmc_legs.log 2>&1; tail -8 $O/pmc_legs.log
cp profiles/traffic.json $O/traffic.json
python tests/tools/pmc_legs_collect.py $O/pmc_legs $O/traffic.json profiles/r06_pmc_legs
find $O/pmc_legs -name "*.csv" -size +20M -delete
find $O/pmc_legs -name "*kernel_trace*" -delete
du -sh $O/pmc_legs
