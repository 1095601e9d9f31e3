#!/bin/bash
# the >= 2^32-row index: build (7-8 min), parity of the new wide path, MEM / pairs rates of the round-2 library and of this one
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_wide; mkdir -p $O
export TMPDIR=/tmp
W=/tmp/kjwide
( time python tests/tools/wide_index.py prepare $W ) > $O/prepare.log 2>&1; tail -4 $O/prepare.log
KAIJU_GPU_LOAD_TIMES=1 timeout 1200 python tests/tools/wide_index.py parity $W > $O/parity.log 2>&1; echo "parity rc=$?"; grep -E "parity|HBM" $O/parity.log
WIDE_LEGS=mem,paired WIDE_NO_REF=1 KAIJU_GPU_LIB=$PWD/kaiju_amd/variants/libkaiju_gpu_base.so timeout 900 python tests/tools/wide_index.py bench $W $O/bench_base.json > $O/bench_base.log 2>&1; grep "\[wide\]" $O/bench_base.log
WIDE_LEGS=mem,paired timeout 900 python tests/tools/wide_index.py bench $W $O/bench_new.json > $O/bench_new.log 2>&1; grep "\[wide\]" $O/bench_new.log
