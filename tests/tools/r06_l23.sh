#!/bin/bash
# lease 23: the hostile (family-structured) database at 2^32 rows and more, sorted for real - the wide lanes on an index
# that is neither i.i.d. nor replicated; the reference binary on 200 000 of its reads in both modes.
O=$GRAFT_REPO_ROOT/gpurun_out/r06_l23; mkdir -p $O
W=/dev/shm/kaiju_hardwide; mkdir -p $W
NSEQ=${HARD_NSEQ:-15900001}
ARGS="--work $W --reads 500000 --steps 3 --warmup 1 --legs hard --hard-nseq $NSEQ --hard-reads 2000000 --leg-steps 3 --cpu-sample 100000 --cpu-sample-legs 200000"
( while sleep 5; do echo "$(date +%s) $(cat /sys/fs/cgroup/memory.current 2>/dev/null)" >> $O/memory.txt; done ) &
WD=$!
KAIJU_GPU_LOAD_TIMES=1 timeout 2100 python bench.py $ARGS > $O/bench_hardwide.json 2> $O/bench_hardwide.err
echo "[l23] bench rc=$?"; grep -v "^\[kaiju_gpu pack\]" $O/bench_hardwide.err | tail -30
ls -la $W > $O/files.txt
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $GRAFT_REPO_ROOT/bench.py $ARGS --no-cpu-baseline --leg-steps 2 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err )
cp $O/stats/s_kernel_stats.csv $O/kernel_stats_hardwide.csv 2>/dev/null; rm -rf $O/stats; head -12 $O/kernel_stats_hardwide.csv
kill $WD 2>/dev/null
sort -k2 -n $O/memory.txt | tail -1 > $O/memory_peak.txt; rm -f $O/memory.txt
rm -rf $W
