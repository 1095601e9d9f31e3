#!/bin/bash
# lease 13: MEM probes (kj_core.h kMemProbe): noprobe / cur
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_l13; mkdir -p $O
export TMPDIR=/tmp
[ -f /tmp/kjw/reads.npy ] || python tests/tools/prof_prepare.py /tmp/kjw 680001 4000000 > $O/prepare.log 2>&1
for v in noprobe cur; do
  PROF_RUN_COUNTS=1 KAIJU_GPU_LIB=$PWD/kaiju_amd/variants/libkaiju_gpu_$v.so timeout 600 python tests/tools/prof_run.py /tmp/kjw mem 1 4 4000000 > $O/mem_$v.txt 2>&1
  echo "== mem $v"; grep -E "search|checksum|ops per" $O/mem_$v.txt | tail -4
done
