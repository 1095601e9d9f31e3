#!/bin/bash
# lease 32: PMC traffic of k_mem at the launch sizes of the default line (5 M reads, 2.5 M pairs per launch)
O=$GRAFT_REPO_ROOT/gpurun_out/r06_l32; mkdir -p $O
PMC_PASSES="1 2" PMC_BENCH_ARGS=" " bash tests/tools/pmc_bench.sh $O/pmc > $O/pmc.log 2>&1; tail -3 $O/pmc.log
PMC_MEM_LAUNCH=5000000 PMC_PAIR_LAUNCH=2500000 PMC_MERGE=profiles/traffic.json python tests/tools/pmc_bench_collect.py $O/pmc $O/traffic.json profiles/r06_pmc_halves > $O/pmc_collect.log 2>&1; cat $O/pmc_collect.log
du -sh $O/pmc
