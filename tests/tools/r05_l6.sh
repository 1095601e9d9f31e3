#!/bin/bash
# round 5, lease 6: the fast stage 1 for mates up to 287 nt, the five-letter resident table: GPU suite + default bench line
O=$1
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/gpu_tests.log 2>&1; echo "[l6] suite rc=$?"; tail -4 $O/gpu_tests.log
( time KAIJU_GPU_LOAD_TIMES=1 timeout 1500 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err ); echo "[l6] bench rc=$?"; grep "leg \|wide index\|failed\|index in HBM" $O/bench_n1.err
