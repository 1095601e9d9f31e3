#!/bin/bash
# .gz input through the command line with the several-thread inflate (csrc/host/pargz.h): 48 M reads (eight gzip -6 members of
# 6 M reads = 1.9 GB of text each: `gzip` itself is one thread, a single member of that size would take the lease's time), threads 1 (zlib's gzread) / 8 / 16 / 32 / 64; parse only and MEM / Greedy
O=${1:-gpurun_out/cli_gz}; mkdir -p $O
R=$(cd "$(dirname "$0")/../.." && pwd)
W=/dev/shm/kjcli; mkdir -p $W
CLI=$R/kaiju_amd/bin/kaiju
python - <<PY
import sys, time, numpy as np
sys.path.insert(0, "$R")
import bench
from kaiju_amd import synth, mkfmi
lines, leaves = synth.make_taxonomy(); synth.write_nodes_dmp("$W/nodes.dmp", lines)
db = synth.make_db(nseq=680001, seed=12345, leaves=leaves)
synth.write_fasta(db, "$W/db.faa"); mkfmi.build_fmi("$W/db.faa", "$W/db.fmi", threads=0, exponent=3)
with open("$W/reads12.fq", "wb") as f:
    for k in range(3):
        f.write(bench.fastq_bytes(synth.make_reads(db, 4_000_000, seed=777 + k), first=4_000_000 * k))
PY
t0=$(date +%s)
# one member per file half: gzip is single-threaded, so two halves side by side, each ONE ordinary gzip member
split -n l/2 -d $W/reads12.fq $W/half_
for p in $W/half_0*; do gzip -6 -c $p > $p.gz & done; wait
cat $W/half_00.gz $W/half_01.gz $W/half_00.gz $W/half_01.gz $W/half_00.gz $W/half_01.gz $W/half_00.gz $W/half_01.gz > $W/reads48.fq.gz; rm -f $W/half_0*
echo "48 M reads gzip -6: $(ls -la $W/reads48.fq.gz | awk '{print $5}') bytes in $(( $(date +%s) - t0 )) s" | tee $O/gz.txt
mkdir -p /tmp/kjgz; g++ -O2 -std=c++17 -o /tmp/kjgz/pargz_test $R/tests/tools/pargz_test.cpp -lz -lpthread
for t in 16 32 64; do echo "inflate only, $t threads: $(PARGZ_NO_OUTPUT=1 /tmp/kjgz/pargz_test $W/reads48.fq.gz $t 2>&1 | tr '\n' ' ')" | tee -a $O/gz.txt; done
( time gzip -dc $W/reads48.fq.gz > /dev/null ) 2>&1 | grep real | sed 's/^/gzip -dc: /' | tee -a $O/gz.txt
run() { local tag=$1 mode=$2; shift 2; local t0=$(date +%s.%N)
  env "$@" KAIJU_GPU_STAGE_TIMES=1 $CLI -t $W/nodes.dmp -f $W/db.fmi -i $W/reads48.fq.gz -o $W/out_$tag.tsv -a $mode 2> $O/err_$tag.txt; local rc=$?; local t1=$(date +%s.%N)
  echo "== $tag rc=$rc: $(python3 -c "w=$t1-$t0; print(round(w, 2), 's wall ->', round(48 / w, 2), 'M reads/s end to end')")" | tee -a $O/gz.txt; }
run warm mem KAIJU_GPU_GZ_THREADS=32 > /dev/null
for t in 32; do run parse_only_t$t mem KAIJU_GPU_PARSE_ONLY=1 KAIJU_GPU_GZ_THREADS=$t; done
run mem_t1 mem KAIJU_GPU_GZ_THREADS=1
for t in 8 16 32 64; do run mem_t$t mem KAIJU_GPU_GZ_THREADS=$t; done
run mem_t32_piece8 mem KAIJU_GPU_GZ_THREADS=32 KAIJU_GPU_GZ_PIECE=8388608
run mem_t32_piece2 mem KAIJU_GPU_GZ_THREADS=32 KAIJU_GPU_GZ_PIECE=2097152
run mem_default mem A=1
grep '^\[gz' $O/err_mem_default.txt | tee -a $O/gz.txt
grep 'CPU time per stage' $O/err_mem_default.txt | tee -a $O/gz.txt
run greedy_default greedy A=1
cmp $W/out_mem_t1.tsv $W/out_mem_t32.tsv && cmp $W/out_mem_t1.tsv $W/out_mem_default.tsv && echo "outputs identical (zlib / 32 threads / default)" | tee -a $O/gz.txt
wc -l $W/out_mem_t1.tsv | tee -a $O/gz.txt
rm -rf $W
