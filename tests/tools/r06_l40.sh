#!/bin/bash
# lease 40: the kernels of a -a greedy -v run of the command line program (2 M reads): where its 1.7 s over the plain run go
O=$GRAFT_REPO_ROOT/gpurun_out/r06_close5; mkdir -p $O
W=/tmp/kjw
python tests/tools/prof_prepare.py $W 680001 2000000 > /dev/null 2>&1
( VB_MODES=mem timeout 600 python tests/tools/cli_verbose_check.py $W 2000000 ) > $O/verbose_check.txt 2>&1; grep "^-a\|OTHER" $O/verbose_check.txt | cut -c1-330
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- $GRAFT_REPO_ROOT/kaiju_amd/bin/kaiju -t $W/nodes.dmp -f $W/db.fmi -i $W/v_2000000.fq -o $W/vg.tsv -a greedy -v > /dev/null 2>&1 )
cp $O/stats/s_kernel_stats.csv $O/kernel_stats_cli_v_greedy.csv 2>/dev/null; rm -rf $O/stats; head -n 8 $O/kernel_stats_cli_v_greedy.csv | cut -c1-200
( cd /tmp && KAIJU_GPU_WALL=1 timeout 120 $GRAFT_REPO_ROOT/kaiju_amd/bin/kaiju -t $W/nodes.dmp -f $W/db.fmi -i $W/v_2000000.fq -o $W/vg.tsv -a greedy -v ) > $O/cli_v_greedy_wall.txt 2>&1
