#!/bin/bash
# lease 18: the device side of the skip rules beyond the suite - the randomised hunt through the HIP path (narrow, then every
# index forced into the wide layout) and the GPU suite with KAIJU_GPU_FORCE_WIDE=20 (wide lanes and their probes on every index)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_l18; mkdir -p $O
export TMPDIR=/tmp
( time timeout 100 python tests/tools/fuzz_gpu.py 60 9801 ) > $O/fuzz_gpu_narrow.log 2>&1; tail -n 2 $O/fuzz_gpu_narrow.log | head -1
( time KAIJU_GPU_FORCE_WIDE=16 timeout 100 python tests/tools/fuzz_gpu.py 60 9802 ) > $O/fuzz_gpu_forced_wide.log 2>&1; tail -n 5 $O/fuzz_gpu_forced_wide.log | head -2
( time KAIJU_GPU_FORCE_WIDE=20 timeout 170 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_zzz_wide.py ) > $O/gpu_tests_forced_wide.log 2>&1; tail -n 6 $O/gpu_tests_forced_wide.log
