#!/bin/bash
# lease 21: the default bench line once more on another box (run-to-run spread of the round's headline)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_l21; mkdir -p $O
export TMPDIR=/tmp
timeout 175 python bench.py > $O/bench_n1_repeat.json 2> $O/bench_n1_repeat.err; echo "bench rc=$?"; tail -n 5 $O/bench_n1_repeat.err
