#!/bin/bash
# lease 1: A/B of the prepared lane variants, footprint sweep of the random-line rate, then the bench with the new parity leg
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_l1; mkdir -p $O
export TMPDIR=/tmp
( time python tests/tools/prof_prepare.py /tmp/kjw 680001 4000000 ) > $O/prepare.log 2>&1
VARIANTS="base gate1 gate3 roll roll_gate1 locp all3" timeout 600 bash tests/tools/mem_variants.sh run $O/mem mem 4000000 > $O/mem_variants.txt 2>&1
VARIANTS="base gdefer gdefer_locp g_occ3" timeout 600 bash tests/tools/mem_variants.sh run $O/greedy greedy 4000000 > $O/greedy_variants.txt 2>&1
timeout 300 tests/tools/randreach 160 256 1 8 0 1 4 11 40 110 160 > $O/randreach.txt 2>&1
timeout 300 tests/tools/randreach 160 256 4 8 0 1 4 11 40 110 160 >> $O/randreach.txt 2>&1
timeout 300 tests/tools/randreach 160 256 14 8 0.5 4.5 20 55 >> $O/randreach.txt 2>&1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
echo bench rc=$?
tail -2 $O/mem_variants.txt; tail -2 $O/greedy_variants.txt; tail -3 $O/randreach.txt
