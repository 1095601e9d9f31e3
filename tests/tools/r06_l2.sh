#!/bin/bash
# round 6, lease 2: first run of the row-pool Greedy lane (k_greedy3) - parity tests, then A/B against k_greedy2
O=$1
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "greedy or Greedy" ) > $O/gpu_greedy_tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/gpu_greedy_tests.log
for v in "KAIJU_GPU_G3_SPLIT=1" "KAIJU_GPU_G3_SPLIT=0" "KAIJU_GPU_GREEDY_LANE=v2"; do
  echo "== $v"
  env $v timeout 600 python bench.py --mode greedy --legs "" --steps 3 --warmup 1 --no-cpu-baseline --no-ref-ops > $O/bench_${v##*=}_$( echo $v | cut -c11-16 ).json 2> $O/bench_${v##*=}_$( echo $v | cut -c11-16 ).err
  echo "rc=$?"; tail -3 $O/bench_${v##*=}_$( echo $v | cut -c11-16 ).err | cut -c1-600
done
