#!/bin/bash
# lease 15: the wide MEM lane with and without probes on the benchmark index forced into the wide layout (KAIJU_GPU_FORCE_WIDE)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_l15; mkdir -p $O
export TMPDIR=/tmp
[ -f /tmp/kjw/reads.npy ] || python tests/tools/prof_prepare.py /tmp/kjw 680001 4000000 > $O/prepare.log 2>&1
for v in noprobe cur; do
  PROF_RUN_COUNTS=1 KAIJU_GPU_FORCE_WIDE=20 KAIJU_GPU_LIB=$PWD/kaiju_amd/variants/libkaiju_gpu_$v.so timeout 600 python tests/tools/prof_run.py /tmp/kjw mem 1 3 4000000 > $O/widemem_$v.txt 2>&1
  echo "== forced wide mem $v"; grep -E "search|checksum|ops per" $O/widemem_$v.txt | tail -3
done
PROF_RUN_COUNTS=1 KAIJU_GPU_LIB=$PWD/kaiju_amd/variants/libkaiju_gpu_cur.so timeout 600 python tests/tools/prof_run.py /tmp/kjw mem 1 3 4000000 > $O/mem_cur.txt 2>&1
echo "== narrow mem cur"; grep -E "search|checksum|ops per" $O/mem_cur.txt | tail -3
