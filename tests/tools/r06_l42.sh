#!/bin/bash
# lease 42: kaiju -v in Greedy mode from the second-generation lane (k_greedy2_vb / k_greedy2_wide_vb + k_mem_verbose<.., false>):
# the tests that read columns 6 / 7, then the cost of -v on 2 M reads in both modes against the first-generation lanes and the reference
O=$GRAFT_REPO_ROOT/gpurun_out/r06_l42; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py tests/test_gpu_shim.py tests/test_gpu_zz_protein_kaijux_mem.py -m gpu -q -x -k "verbose or cli or shim or kaijux or kaijup or protein" ) > $O/verbose_tests.log 2>&1; echo "[l42] verbose tests rc=$?"; tail -n 3 $O/verbose_tests.log
( time KAIJU_GPU_FORCE_WIDE=20 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_zz_protein_kaijux_mem.py -m gpu -q -x -k "verbose or kaijux or kaijup or protein" ) > $O/verbose_tests_wide.log 2>&1; echo "[l42] forced wide rc=$?"; tail -n 3 $O/verbose_tests_wide.log
python tests/tools/prof_prepare.py /tmp/kjw 680001 2000000 > /dev/null 2>&1
( timeout 900 python tests/tools/cli_verbose_check.py /tmp/kjw 2000000 ) > $O/verbose_check.txt 2>&1; echo "[l42] verbose check rc=$?"; grep "^-a\|OTHER\|gpu:\|ref:" $O/verbose_check.txt | cut -c1-400
W=/tmp/kjw
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- $GRAFT_REPO_ROOT/kaiju_amd/bin/kaiju -t $W/nodes.dmp -f $W/db.fmi -i $W/v_2000000.fq -o $W/vg.tsv -a greedy -v > /dev/null 2>&1 )
cp $O/stats/s_kernel_stats.csv $O/kernel_stats_cli_v_greedy.csv 2>/dev/null; rm -rf $O/stats; head -n 8 $O/kernel_stats_cli_v_greedy.csv | cut -c1-160
