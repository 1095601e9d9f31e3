#!/bin/bash
# round 5, closing lease: the GPU suite as the driver runs it, the same with every index forced wide / without text arrays /
# with every .fmi streamed and packed on the device, the randomised hunt on the device (narrow and forced wide)
O=$1
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/gpu_tests.log 2>&1; echo "[final] suite rc=$?"; tail -3 $O/gpu_tests.log
bash tests/tools/forced_wide_suite.sh $O
( time KAIJU_GPU_FMI_STREAM=1 timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_zz_protein_kaijux_mem.py tests/test_gpu_cli.py -m gpu -q -x ) > $O/fmi_stream_tests.log 2>&1; echo "[final] every .fmi streamed rc=$?"; tail -3 $O/fmi_stream_tests.log
( time timeout 600 python tests/tools/fuzz_gpu.py 40 71 ) > $O/fuzz_gpu_narrow.log 2>&1; echo "[final] fuzz narrow rc=$?"; tail -2 $O/fuzz_gpu_narrow.log
( time KAIJU_GPU_FORCE_WIDE=16 timeout 600 python tests/tools/fuzz_gpu.py 40 72 ) > $O/fuzz_gpu_wide.log 2>&1; echo "[final] fuzz forced wide rc=$?"; tail -2 $O/fuzz_gpu_wide.log
