#!/bin/bash
# round 6, closing lease: the GPU suite as the driver runs it, forced wide / no text arrays / every .fmi streamed, the randomised
# hunt on the device (narrow and forced wide), -v against the reference binary with its cost
O=$1
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/gpu_tests.log 2>&1; echo "[final] suite rc=$?"; tail -3 $O/gpu_tests.log
bash tests/tools/forced_wide_suite.sh $O
( time KAIJU_GPU_FMI_STREAM=1 timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_zz_protein_kaijux_mem.py tests/test_gpu_cli.py -m gpu -q -x ) > $O/fmi_stream_tests.log 2>&1; echo "[final] every .fmi streamed rc=$?"; tail -3 $O/fmi_stream_tests.log
( time timeout 900 python tests/tools/fuzz_gpu.py 60 81 ) > $O/fuzz_gpu_narrow.log 2>&1; echo "[final] fuzz narrow rc=$?"; tail -2 $O/fuzz_gpu_narrow.log
( time KAIJU_GPU_FORCE_WIDE=16 timeout 900 python tests/tools/fuzz_gpu.py 40 82 ) > $O/fuzz_gpu_wide.log 2>&1; echo "[final] fuzz forced wide rc=$?"; tail -2 $O/fuzz_gpu_wide.log
python tests/tools/prof_prepare.py /tmp/kjw 680001 2000000 > /dev/null 2>&1
( timeout 900 python tests/tools/cli_verbose_check.py /tmp/kjw 2000000 ) > $O/verbose_check.txt 2>&1; echo "[final] verbose rc=$?"; cat $O/verbose_check.txt | cut -c1-300
