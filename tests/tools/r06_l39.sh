#!/bin/bash
# lease 39: column 7 packed on the device (k_vb_pack, kaiju_gpu_classify_batch_verbose_packed in the command line programs): the
# tests that read columns 6 / 7 again, the cost of -v on 2 M reads, and the kernels of a -v run
O=$GRAFT_REPO_ROOT/gpurun_out/r06_l39; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py tests/test_gpu_shim.py tests/test_gpu_zz_protein_kaijux_mem.py -m gpu -q -x -k "verbose or cli or shim or kaijux or kaijup or protein" ) > $O/verbose_tests.log 2>&1; echo "[l39] verbose tests rc=$?"; tail -n 3 $O/verbose_tests.log
( time KAIJU_GPU_FORCE_WIDE=20 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_zz_protein_kaijux_mem.py -m gpu -q -x -k "verbose or kaijux or kaijup or protein" ) > $O/verbose_tests_wide.log 2>&1; echo "[l39] forced wide rc=$?"; tail -n 3 $O/verbose_tests_wide.log
python tests/tools/prof_prepare.py /tmp/kjw 680001 2000000 > /dev/null 2>&1
( timeout 900 python tests/tools/cli_verbose_check.py /tmp/kjw 2000000 ) > $O/verbose_check.txt 2>&1; echo "[l39] verbose check rc=$?"; grep "^-a\|OTHER\|gpu:\|ref:" $O/verbose_check.txt | cut -c1-400
W=/tmp/kjw
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- $GRAFT_REPO_ROOT/kaiju_amd/bin/kaiju -t $W/nodes.dmp -f $W/db.fmi -i $W/v_2000000.fq -o $W/vg.tsv -a mem -v > /dev/null 2>&1 )
cp $O/stats/s_kernel_stats.csv $O/kernel_stats_cli_v_mem.csv 2>/dev/null; rm -rf $O/stats; head -n 14 $O/kernel_stats_cli_v_mem.csv | cut -c1-160
( cd /tmp && KAIJU_GPU_WALL=1 $GRAFT_REPO_ROOT/kaiju_amd/bin/kaiju -t $W/nodes.dmp -f $W/db.fmi -i $W/v_2000000.fq -o $W/vg.tsv -a mem -v 2>&1 | tail -n 25 ) > $O/cli_v_wall.txt 2>&1
