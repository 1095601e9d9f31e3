#!/bin/bash
# lease 17: k_mem variants behind the probes (fragment-switch gate 0/1/3, the probe's look at the line's presence bits, blocks per CU)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_l17; mkdir -p $O
export TMPDIR=/tmp
[ -f /tmp/kjw/reads.npy ] || python tests/tools/prof_prepare.py /tmp/kjw 680001 4000000 > $O/prepare.log 2>&1
for v in cur nopa gate0 gate3; do   # (nopa / gate0 / gate3: -DKJ_NO_PREV_ABSENT, -DKJ_MEM_GATE=0|3 - switches that were removed with the result)
  KAIJU_GPU_LIB=$PWD/kaiju_amd/variants/libkaiju_gpu_$v.so timeout 600 python tests/tools/prof_run.py /tmp/kjw mem 1 4 4000000 > $O/mem_$v.txt 2>&1
  echo "== mem $v"; grep -E "search|checksum" $O/mem_$v.txt | tail -3
done
for occ in 1 3; do
  KAIJU_GPU_BLOCKS_PER_CU=$occ KAIJU_GPU_LIB=$PWD/kaiju_amd/variants/libkaiju_gpu_cur.so timeout 600 python tests/tools/prof_run.py /tmp/kjw mem 1 3 4000000 > $O/mem_cur_occ$occ.txt 2>&1
  echo "== mem cur blocks/CU $occ"; grep -E "search" $O/mem_cur_occ$occ.txt | tail -1
done
