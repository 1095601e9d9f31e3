#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_l14; mkdir -p $O
export TMPDIR=/tmp
python tests/tools/prof_prepare.py /tmp/kjw 680001 4000000 > $O/prepare.log 2>&1
run() { name=$1; lib=$2; mode=$3; shift 3
  env "$@" KAIJU_GPU_LIB=$PWD/$lib python tests/tools/prof_run.py /tmp/kjw $mode 1 3 4000000 > $O/$name.txt 2>&1
  echo "== $name"; grep -E "search|checksum" $O/$name.txt | tail -2; }
run mem_new kaiju_amd/libkaiju_gpu.so mem X=1
run mem_bgate1 kaiju_amd/variants/libkaiju_gpu_bgate1.so mem X=1
run mem_bgate3 kaiju_amd/variants/libkaiju_gpu_bgate3.so mem X=1
