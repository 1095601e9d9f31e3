#!/bin/bash
# HBM traffic of the search kernels of the legs hard / hard_greedy / wide / wide_greedy / long / protein (round 6: every leg
# that carries a roofline gets its `traffic`): bench.py under rocprofv3 --pmc with ONE leg group per run (wide / long / protein take the first reads of the headline's
# workload, so the headline has the leg's size: 2 M reads; the collector tells the launches apart by kernel name and order), the two request-count groups only.
#   usage: pmc_legs.sh <outdir> [legs...]      then: pmc_legs_collect.py <outdir> profiles/traffic.json
OUT=$1; shift
LEGS=${@:-hard wide long protein}
R=$(cd "$(dirname "$0")/../.." && pwd)
export TMPDIR=/tmp
mkdir -p $OUT; OUT=$(cd $OUT && pwd)
cd /tmp
for leg in $LEGS; do
  i=0
  for ctrs in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
    i=$((i+1))
    mkdir -p $OUT/$leg
    timeout 900 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/$leg/pass$i -o p -- \
      python $R/bench.py --reads 2000000 --contexts 1 --steps 1 --warmup 0 --leg-steps 1 --no-cpu-baseline --no-ref-ops --parity-sample 0 --legs $leg \
      > $OUT/$leg/pass$i.json 2> $OUT/$leg/pass$i.log
    echo "$leg pass $i rc=$? : $ctrs"
    rm -f $OUT/$leg/pass$i/p_agent_info.csv
  done
done
