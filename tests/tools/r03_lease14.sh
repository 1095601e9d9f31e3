#!/bin/bash
# lease 14: Greedy probe (kj_core.h kGreedyProbe): noprobe / cur, heavy-iteration gate 1 and 3
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_l14; mkdir -p $O
export TMPDIR=/tmp
[ -f /tmp/kjw/reads.npy ] || python tests/tools/prof_prepare.py /tmp/kjw 680001 4000000 > $O/prepare.log 2>&1
for v in noprobe cur; do
  for g in 1; do
  PROF_RUN_COUNTS=1 KAIJU_GPU_GREEDY_GATE=$g KAIJU_GPU_LIB=$PWD/kaiju_amd/variants/libkaiju_gpu_$v.so timeout 600 python tests/tools/prof_run.py /tmp/kjw greedy 1 3 4000000 > $O/greedy_${v}_g$g.txt 2>&1
  echo "== greedy $v gate $g"; grep -E "search|checksum|ops per" $O/greedy_${v}_g$g.txt | tail -3
  done
done
