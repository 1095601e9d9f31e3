#!/bin/bash
# HBM traffic of the search kernels at the benchmark's launch sizes: bench.py itself under rocprofv3 --pmc, one counter
# group per run (only --kernel-trace next to --pmc); raw CSVs stay in <outdir>, pmc_bench_collect.py turns them into
# profiles/traffic.json.      usage: pmc_bench.sh <outdir>
# PMC_BENCH_ARGS=" " (instead of the default "--contexts 1": one launch per step): the launch sizes of the default line - a MEM step of
# 4 M reads and more as two halves -, one after the other (KAIJU_BENCH_SERIAL); pmc_bench_collect.py then wants PMC_MEM_LAUNCH /
# PMC_PAIR_LAUNCH (reads / pairs per launch) and, to keep the records that are there, PMC_MERGE=<traffic.json>
OUT=$1
R=$(cd "$(dirname "$0")/../.." && pwd)
export TMPDIR=/tmp
mkdir -p $OUT; OUT=$(cd $OUT && pwd)
cd /tmp
i=0
for ctrs in \
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" \
  "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" ; do
  i=$((i+1))
  # (PMC_PASSES="1 2": only those counter groups - the request counts the traffic figure is made of)
  if [ -n "$PMC_PASSES" ]; then case " $PMC_PASSES " in *" $i "*) ;; *) continue ;; esac; fi
  timeout 600 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/pass$i -o p -- \
    env KAIJU_BENCH_SERIAL=1 python $R/bench.py ${PMC_BENCH_ARGS:---contexts 1} --steps 1 --warmup 0 --no-cpu-baseline --legs greedy,paired > $OUT/pass$i.json 2> $OUT/pass$i.log
  echo "pass $i rc=$? : $ctrs"
  rm -f $OUT/pass$i/p_agent_info.csv
done
