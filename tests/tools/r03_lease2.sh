#!/bin/bash
# lease 2: GPU tests of the k-mer-line build, A/B against the round-2 lanes (variants/base), translation experiments
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_l2; mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?" 
python tests/tools/prof_prepare.py /tmp/kjw 680001 4000000 > $O/prepare.log 2>&1
for mode in mem greedy; do
  for v in base new; do
    lib=kaiju_amd/variants/libkaiju_gpu_base.so; [ $v = new ] && lib=kaiju_amd/libkaiju_gpu.so
    KAIJU_GPU_LIB=$PWD/$lib KAIJU_GPU_LOAD_TIMES=1 python tests/tools/prof_run.py /tmp/kjw $mode 1 3 4000000 > $O/${mode}_$v.txt 2>&1
    echo "== $mode $v"; grep -E "search|checksum" $O/${mode}_$v.txt | tail -2
  done
done
for m in 1 2 3 41 42 4; do timeout 300 tests/tools/randreach 64 256 $m 8 0 1 8 40 >> $O/randreach2.txt 2>&1; done
cat $O/randreach2.txt
