#!/bin/bash
# lease 20: the drop-in command line end to end with the round's kernels (FASTQ in the page cache -> output file, index load included)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_l20; mkdir -p $O
export TMPDIR=/tmp
[ -f /tmp/kjw/reads.npy ] || python tests/tools/prof_prepare.py /tmp/kjw 680001 4000000 > $O/prepare.log 2>&1
timeout 120 python tests/tools/cli_time.py /tmp/kjw 4000000 mem > $O/cli_time_mem.txt 2>&1; cat $O/cli_time_mem.txt
timeout 120 python tests/tools/cli_time.py /tmp/kjw 4000000 greedy > $O/cli_time_greedy.txt 2>&1; cat $O/cli_time_greedy.txt
KAIJU_GPU_LOAD_TIMES=1 timeout 60 kaiju_amd/bin/kaiju -t /tmp/kjw/nodes.dmp -f /tmp/kjw/db.fmi -i /tmp/kjw/reads_4000000.fq -o /tmp/kjw/o.tsv -a mem 2> $O/load_times.txt; tail -n 20 $O/load_times.txt
