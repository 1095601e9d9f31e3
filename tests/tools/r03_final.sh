#!/bin/bash
# the round's closing measurements: GPU suite, PMC passes (-> profiles/traffic.json, so that the bench line that follows carries the
# traffic of the kernels it times), bench line, rocprofv3 kernel stats of the headline and the Greedy leg
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_final; mkdir -p $O
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -m gpu -q ) > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -4 $O/gpu_tests.log
bash tests/tools/pmc_bench.sh $O/pmc > $O/pmc.log 2>&1; tail -5 $O/pmc.log
python tests/tools/pmc_bench_collect.py $O/pmc profiles/traffic.json profiles/r03_pmc > $O/pmc_collect.log 2>&1; cp profiles/traffic.json $O/traffic.json; tail -3 $O/pmc_collect.log
timeout 1200 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; tail -5 $O/bench_n1.err
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -o s -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --legs "" --steps 5 > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/bench_under_rocprof.err )
cp $O/stats/s_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null; rm -rf $O/stats
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_g -o s -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --mode greedy --legs "" --steps 3 > $GRAFT_REPO_ROOT/$O/bench_greedy_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/bench_greedy_under_rocprof.err )
cp $O/stats_g/s_kernel_stats.csv $O/kernel_stats_greedy.csv 2>/dev/null; rm -rf $O/stats_g
find $O/pmc -name "*.csv" | xargs ls -la | head; du -sh $O
