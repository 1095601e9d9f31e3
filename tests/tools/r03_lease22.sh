#!/bin/bash
# lease 22: k_mem with the selectivity guard of the narrow probes (variants/cur) against the library of the closing measurements;
# a small workload (150 k proteins, 2 M reads) prepared on the spot - what is left of the round's GPU minutes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_l22; mkdir -p $O
W=/tmp/kjs
[ -f $W/reads.npy ] || python tests/tools/prof_prepare.py $W 150001 2000000 > $O/prepare.log 2>&1
for v in new base new base; do
  lib=$PWD/kaiju_amd/libkaiju_gpu.so; [ $v = new ] && lib=$PWD/kaiju_amd/variants/libkaiju_gpu_cur.so
  KAIJU_GPU_LIB=$lib timeout 20 python tests/tools/prof_run.py $W mem 1 4 2000000 > $O/mem_$v.txt 2>&1
  echo "== $v: $(grep search $O/mem_$v.txt | tail -n 2 | sed 's/.*search \([0-9.]*\).*/\1/' | tr '\n' ' ') $(grep checksum $O/mem_$v.txt)"
done
