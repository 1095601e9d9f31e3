#!/bin/bash
# the GPU suite's parity tests with every index forced into the wide layout (64-bit positions, team locate, wide lanes)
O=${1:-gpurun_out/fw}; mkdir -p $O
( time KAIJU_GPU_FORCE_WIDE=20 timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_zz_protein_kaijux_mem.py tests/test_gpu_zzz_wide.py -m gpu -q -x ) > $O/forced_wide_tests.log 2>&1; echo "forced wide rc=$?"; tail -3 $O/forced_wide_tests.log
( time KAIJU_GPU_NO_TEXT=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x ) > $O/no_text_tests.log 2>&1; echo "no text (team locate on a narrow index) rc=$?"; tail -3 $O/no_text_tests.log
