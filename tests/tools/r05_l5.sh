#!/bin/bash
# round 5, lease 5: the Greedy lanes without their in-lane locate, k_greedy2_wide at three wavefronts per SIMD: GPU suite + the
# default bench line (legs long / protein new)
O=$1
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/gpu_tests.log 2>&1; echo "[l5] suite rc=$?"; tail -4 $O/gpu_tests.log
( time timeout 1500 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err ); echo "[l5] bench rc=$?"; grep "leg \|wide index\|failed" $O/bench_n1.err
