#!/bin/bash
# round 6, lease 14: Greedy with one-row variant chains that cannot matter not queued (-DKJ_CHAIN_PRUNE): A/B on the 4 M-read
# profiling workload (records must agree), section profile, and the bench's Greedy legs with their parity against the reference
O=$1
V=kaiju_amd/variants
PROF_RUN_COUNTS=1 VARIANTS="${AB_VARIANTS:-cur prune cur prune}" bash tests/tools/mem_variants.sh run $O/ab greedy 4000000
grep "ops per read" $O/ab/cur.txt $O/ab/prune.txt | cut -c1-700
VARIANTS="pruneprof" bash tests/tools/mem_variants.sh run $O/prof greedy 2000000
grep "kj prof" $O/prof/pruneprof.txt | head -28
KAIJU_GPU_LIB=$PWD/$V/libkaiju_gpu_${BENCH_VARIANT:-prune}.so timeout 1200 python bench.py --mode greedy --legs hard --steps 3 --warmup 1 --leg-steps 2 --cpu-sample 400000 --cpu-sample-legs 200000 > $O/bench_prune.json 2> $O/bench_prune.err
echo "bench rc=$?"; grep "leg \|mismatch\|parity" $O/bench_prune.err | cut -c1-300 | head; tail -c 1500 $O/bench_prune.json
