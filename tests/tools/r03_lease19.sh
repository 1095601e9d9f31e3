#!/bin/bash
# lease 19: the rebuilt in-tree library at the round's last commit - smoke() and a short bench line without the CPU legs
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_l19; mkdir -p $O
export TMPDIR=/tmp
( time python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke.log 2>&1; tail -n 5 $O/smoke.log | head -2
timeout 150 python bench.py --no-cpu-baseline --legs greedy --steps 2 --leg-steps 1 > $O/bench_short.json 2> $O/bench_short.err; echo "bench rc=$?"; tail -n 3 $O/bench_short.err
