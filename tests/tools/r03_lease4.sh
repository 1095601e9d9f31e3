#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_l4; mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log
python tests/tools/prof_prepare.py /tmp/kjw 680001 4000000 > $O/prepare.log 2>&1
for v in base new; do
  lib=kaiju_amd/variants/libkaiju_gpu_$v.so; [ $v = new ] && lib=kaiju_amd/libkaiju_gpu.so
  for fw in 0 31; do
    if [ $fw = 0 ]; then KAIJU_GPU_LIB=$PWD/$lib python tests/tools/prof_run.py /tmp/kjw mem 1 3 4000000 > $O/mem_${v}_narrow.txt 2>&1
    else KAIJU_GPU_FORCE_WIDE=$fw KAIJU_GPU_LIB=$PWD/$lib python tests/tools/prof_run.py /tmp/kjw mem 1 3 4000000 > $O/mem_${v}_fw$fw.txt 2>&1; fi
    echo "== mem $v forcewide=$fw"; grep -E "search|checksum" $O/mem_${v}_*$( [ $fw = 0 ] && echo narrow || echo fw$fw ).txt | tail -2
  done
done
