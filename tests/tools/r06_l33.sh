#!/bin/bash
# lease 33: fuzz runs on the final library - narrow, forced wide with the row -> taxon table, forced wide without it
O=$GRAFT_REPO_ROOT/gpurun_out/r06_l33; mkdir -p $O
( time timeout 900 python tests/tools/fuzz_gpu.py 60 101 ) > $O/fuzz_gpu_narrow.log 2>&1; echo "[l33] fuzz narrow rc=$?"; tail -2 $O/fuzz_gpu_narrow.log
( time KAIJU_GPU_FORCE_WIDE=16 timeout 900 python tests/tools/fuzz_gpu.py 40 102 ) > $O/fuzz_gpu_wide.log 2>&1; echo "[l33] fuzz forced wide rc=$?"; tail -2 $O/fuzz_gpu_wide.log
( time KAIJU_GPU_FORCE_WIDE=18 KAIJU_GPU_ROW_TAX=0 timeout 900 python tests/tools/fuzz_gpu.py 20 103 ) > $O/fuzz_gpu_wide_walks.log 2>&1; echo "[l33] fuzz forced wide, no table rc=$?"; tail -2 $O/fuzz_gpu_wide_walks.log
( time KAIJU_GPU_FORCE_WIDE=17 KAIJU_GPU_TV_SHIFT=-1 timeout 900 python tests/tools/fuzz_gpu.py 20 104 ) > $O/fuzz_gpu_wide_notext.log 2>&1; echo "[l33] fuzz forced wide, table only rc=$?"; tail -2 $O/fuzz_gpu_wide_notext.log
