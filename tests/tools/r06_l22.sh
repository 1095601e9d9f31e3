#!/bin/bash
# round 6, lease 22: the chain test against the read's best score instead of min_score - Greedy parity tests, the Greedy legs
O=$1
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "greedy or Greedy or fullsize or randomised" ) > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
( timeout 600 python tests/tools/fuzz_gpu.py 40 101 ) > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz.log
timeout 1200 python bench.py --mode greedy --legs hard --steps 3 --warmup 1 --leg-steps 2 --cpu-sample 400000 --cpu-sample-legs 200000 > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; grep "leg " $O/bench.err | cut -c1-200; tail -c 900 $O/bench.json
