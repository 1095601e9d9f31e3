#!/bin/bash
# closing lease, second part: -v against the reference binary with its cost (both modes, both generations of lanes), forced-wide
# suite, the randomised hunt on the device
O=$GRAFT_REPO_ROOT/gpurun_out/r06_close6; mkdir -p $O
python tests/tools/prof_prepare.py /tmp/kjw 680001 2000000 > /dev/null 2>&1
( timeout 900 python tests/tools/cli_verbose_check.py /tmp/kjw 2000000 ) > $O/verbose_check.txt 2>&1; echo "[close6] verbose rc=$?"; grep "^-a\|OTHER\|gpu:\|ref:" $O/verbose_check.txt | cut -c1-330
bash tests/tools/forced_wide_suite.sh $O 2>&1 | grep -v "^$" | tail -n 8
( time timeout 600 python tests/tools/fuzz_gpu.py 25 91 ) > $O/fuzz_gpu_narrow.log 2>&1; echo "[close6] fuzz narrow rc=$?"; tail -n 2 $O/fuzz_gpu_narrow.log | head -n 1
( time KAIJU_GPU_FORCE_WIDE=16 timeout 600 python tests/tools/fuzz_gpu.py 15 92 ) > $O/fuzz_gpu_wide.log 2>&1; echo "[close6] fuzz forced wide rc=$?"; grep "ok\|MISMATCH" $O/fuzz_gpu_wide.log | tail -n 1
