#!/bin/bash
# lease 38: kaiju -v in MEM mode from the second-generation lanes (k_mem_vb / k_mem_wide2_vb + k_mem_verbose): the tests that read
# columns 6 / 7 (library, command line programs, the reference-side shim; narrow, forced wide, without text arrays), then the
# cost of -v on 2 M reads against the first-generation lanes and the reference binary
O=$GRAFT_REPO_ROOT/gpurun_out/r06_l38; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py tests/test_gpu_shim.py tests/test_gpu_zz_protein_kaijux_mem.py -m gpu -q -x -k "verbose or cli or shim or kaijux or kaijup or protein" ) > $O/verbose_tests.log 2>&1; echo "[l38] verbose tests rc=$?"; tail -3 $O/verbose_tests.log
( time KAIJU_GPU_FORCE_WIDE=20 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_zz_protein_kaijux_mem.py -m gpu -q -x -k "verbose or kaijux or kaijup or protein" ) > $O/verbose_tests_wide.log 2>&1; echo "[l38] forced wide rc=$?"; tail -3 $O/verbose_tests_wide.log
( time KAIJU_GPU_NO_TEXT=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "verbose" ) > $O/verbose_tests_notext.log 2>&1; echo "[l38] no text rc=$?"; tail -3 $O/verbose_tests_notext.log
python tests/tools/prof_prepare.py /tmp/kjw 680001 2000000 > /dev/null 2>&1
( VB_MODES=mem timeout 900 python tests/tools/cli_verbose_check.py /tmp/kjw 2000000 ) > $O/verbose_check.txt 2>&1; echo "[l38] verbose check rc=$?"; cut -c1-400 $O/verbose_check.txt
