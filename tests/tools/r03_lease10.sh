#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_l10; mkdir -p $O
export TMPDIR=/tmp
python tests/tools/prof_prepare.py /tmp/kjw 680001 4000000 > $O/prepare.log 2>&1
SWEEP="3:32,3:0,1:32,1:0,7:32,3:16,1:16,0:0" python tests/tools/greedy_sweep.py /tmp/kjw 4000000 > $O/gate_sweep.txt 2>&1; cat $O/gate_sweep.txt
KAIJU_GPU_LIB=$PWD/kaiju_amd/variants/libkaiju_gpu_prof.so python tests/tools/prof_run.py /tmp/kjw greedy 1 2 4000000 > $O/greedy_prof.txt 2>&1
grep "kj prof" $O/greedy_prof.txt | tail -27
