#!/bin/bash
# A/B of the current library against the round-2 lanes (kaiju_amd/variants/libkaiju_gpu_base.so) on the prepared workload
#   r03_ab.sh <outdir> [gpu tests: 1|0] [variants...]
cd $GRAFT_REPO_ROOT
O=$1; mkdir -p $O; shift
T=${1:-0}; shift
export TMPDIR=/tmp
if [ "$T" = 1 ]; then ( timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log; fi
[ -f /tmp/kjw/reads.npy ] || python tests/tools/prof_prepare.py /tmp/kjw 680001 4000000 > $O/prepare.log 2>&1
for mode in mem greedy; do
  for v in base new "$@"; do
    lib=kaiju_amd/variants/libkaiju_gpu_$v.so; [ $v = new ] && lib=kaiju_amd/libkaiju_gpu.so
    [ -f $lib ] || continue
    KAIJU_GPU_LIB=$PWD/$lib python tests/tools/prof_run.py /tmp/kjw $mode 1 3 4000000 > $O/${mode}_$v.txt 2>&1
    echo "== $mode $v"; grep -E "search|checksum" $O/${mode}_$v.txt | tail -2
  done
done
