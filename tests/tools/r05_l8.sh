#!/bin/bash
# round 5, lease 8: the heavy-iteration gate of the Greedy lane once more (the lane lost its locate: other optimum?), and the wide
# legs with the walking locate reading one table per row instead of two
O=$1
python tests/tools/prof_prepare.py /tmp/kjw 680001 2000000 > $O/iid_prepare.log 2>&1
SWEEP=1:32,1:0,1:16,3:32,3:0,0:0,7:32,1:48 python tests/tools/greedy_sweep.py /tmp/kjw 2000000 > $O/greedy_sweep.txt 2>&1; cat $O/greedy_sweep.txt | tail -9
timeout 900 python bench.py --legs wide --no-cpu-baseline --steps 2 > $O/bench_wide.json 2> $O/bench_wide.err; grep "leg \|wide index" $O/bench_wide.err
