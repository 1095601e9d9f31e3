#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_l12; mkdir -p $O
export TMPDIR=/tmp
python tests/tools/prof_prepare.py /tmp/kjw 680001 4000000 > $O/prepare.log 2>&1
run() { name=$1; lib=$2; mode=$3; shift 3
  env "$@" KAIJU_GPU_LIB=$PWD/$lib python tests/tools/prof_run.py /tmp/kjw $mode 1 3 4000000 > $O/$name.txt 2>&1
  echo "== $name"; grep -E "search|checksum" $O/$name.txt | tail -2; }
run greedy_pass4 kaiju_amd/libkaiju_gpu.so greedy X=1
run greedy_pass8 kaiju_amd/variants/libkaiju_gpu_ext8.so greedy X=1
( timeout 900 python -m pytest tests/test_gpu_cli.py -m gpu -x -q ) > $O/gpu_cli_tests.log 2>&1; echo "cli tests rc=$?"; tail -2 $O/gpu_cli_tests.log
