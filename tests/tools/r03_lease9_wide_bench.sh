#!/bin/bash
# BASELINE configs[3] class through the normal bench.py: a database of 15.5 M proteins / 4.33 G residues, index of 2^32 rows and more
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_wide; mkdir -p $O
export TMPDIR=/tmp
( time python bench.py --nseq 15500001 --reads 3000000 --steps 3 --warmup 1 --legs greedy --leg-steps 2 --cpu-sample 200000 --cpu-sample-legs 100000 ) > $O/bench_n1_wide.json 2> $O/bench_n1_wide.err
echo "rc=$?"; tail -12 $O/bench_n1_wide.err
