#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_l8; mkdir -p $O
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -5 $O/gpu_tests.log
python tests/tools/prof_prepare.py /tmp/kjw 680001 4000000 > $O/prepare.log 2>&1
run() { name=$1; lib=$2; mode=$3; shift 3
  env "$@" KAIJU_GPU_LIB=$PWD/$lib python tests/tools/prof_run.py /tmp/kjw $mode 1 3 4000000 > $O/$name.txt 2>&1
  echo "== $name"; grep -E "search|checksum" $O/$name.txt | tail -2; }
run greedy_new kaiju_amd/libkaiju_gpu.so greedy X=1
run mem_new kaiju_amd/libkaiju_gpu.so mem X=1
