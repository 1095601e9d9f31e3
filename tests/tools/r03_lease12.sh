#!/bin/bash
# lease 12: the span rule (kj_core.h kSpanRule) - norule / cur (k-mer lookups) / step (+ intervals that shrink to one row later)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_l12; mkdir -p $O
export TMPDIR=/tmp
[ -f /tmp/kjw/reads.npy ] || python tests/tools/prof_prepare.py /tmp/kjw 680001 4000000 > $O/prepare.log 2>&1
for mode in mem greedy; do
  for v in norule cur step; do
    [ $mode = mem ] && [ $v = step ] && continue
    PROF_RUN_COUNTS=1 KAIJU_GPU_LIB=$PWD/kaiju_amd/variants/libkaiju_gpu_$v.so timeout 600 python tests/tools/prof_run.py /tmp/kjw $mode 1 3 4000000 > $O/${mode}_$v.txt 2>&1
    echo "== $mode $v"; grep -E "search|checksum|ops per" $O/${mode}_$v.txt | tail -3
  done
done
for gw in "1 32" "3 16" "3 48"; do
  set -- $gw
  KAIJU_GPU_GREEDY_GATE=$1 KAIJU_GPU_GREEDY_WAITERS=$2 KAIJU_GPU_LIB=$PWD/kaiju_amd/variants/libkaiju_gpu_step.so timeout 600 python tests/tools/prof_run.py /tmp/kjw greedy 1 2 4000000 > $O/sweep_step_$1_$2.txt 2>&1
  echo "== sweep step gate $1 waiters $2: $(grep search $O/sweep_step_$1_$2.txt | tail -1)"
done
