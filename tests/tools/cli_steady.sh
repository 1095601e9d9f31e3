#!/bin/bash
# Steady-state rate of the drop-in command line (round 6): 408 M reads (a 24 M-read FASTQ file 17 times in a row, 126 GB in
# /dev/shm = the size class of BASELINE configs[4]'s input), plain and gzip (48 M reads: one inflate thread), MEM and Greedy,
# with the wall-clock marks and per-stage CPU times of KAIJU_GPU_STAGE_TIMES.   usage (lease.sh): sh:tests/tools/cli_steady.sh
O=${1:-gpurun_out/cli_steady}; mkdir -p $O
R=$(cd "$(dirname "$0")/../.." && pwd)
W=/dev/shm/kjcli; mkdir -p $W
CLI=$R/kaiju_amd/bin/kaiju
python - <<PY
import sys, time, zlib, numpy as np
from concurrent.futures import ProcessPoolExecutor
sys.path.insert(0, "$R")
import bench
from kaiju_amd import synth, mkfmi
lines, leaves = synth.make_taxonomy(); synth.write_nodes_dmp("$W/nodes.dmp", lines)
db = synth.make_db(nseq=680001, seed=12345, leaves=leaves)
synth.write_fasta(db, "$W/db.faa"); mkfmi.build_fmi("$W/db.faa", "$W/db.fmi", threads=0, exponent=3)
t = time.time()
def piece(k):
    raw = bench.fastq_bytes(synth.make_reads(db, 4_000_000, seed=777 + k), first=4_000_000 * k)
    return raw, zlib.compress(raw, 1) if False else None
with open("$W/reads24.fq", "wb") as f:
    for k in range(6):
        f.write(bench.fastq_bytes(synth.make_reads(db, 4_000_000, seed=777 + k), first=4_000_000 * k))
print("24 M reads written", round(time.time() - t, 1), "s", flush=True)
PY
t0=$(date +%s)
for k in $(seq 17); do cat $W/reads24.fq; done > $W/reads408.fq
echo "408 M reads: $(ls -la $W/reads408.fq | awk '{print $5}') bytes in $(( $(date +%s) - t0 )) s"
# gzip: 8 pieces compressed side by side (concatenated members are one gzip file), 24 M reads; twice in a row = 48 M
t0=$(date +%s)
split -n l/8 -d $W/reads24.fq $W/part_
for p in $W/part_0*; do gzip -1 -c $p > $p.gz & done; wait
cat $W/part_0*.gz $W/part_0*.gz > $W/reads48.fq.gz; rm -f $W/part_0*
echo "48 M reads gzip: $(ls -la $W/reads48.fq.gz | awk '{print $5}') bytes in $(( $(date +%s) - t0 )) s"
ls -la $W > $O/files.txt; free -g >> $O/files.txt
run() { local tag=$1 inp=$2 nM=$3 mode=$4; shift 4; local t0=$(date +%s.%N)
  env "$@" KAIJU_GPU_STAGE_TIMES=1 KAIJU_GPU_LOAD_TIMES=1 $CLI -t $W/nodes.dmp -f $W/db.fmi -i $inp -o /dev/null -a $mode 2> $O/err_$tag.txt; local rc=$?; local t1=$(date +%s.%N)
  echo "== $tag rc=$rc: $(python3 -c "w=$t1-$t0; print(round(w, 2), 's wall ->', round($nM / w, 1), 'M reads/s end to end;', round($nM / max(w - 1.0, 1e-9), 1), 'M reads/s after the first second')")" | tee -a $O/steady.txt
  grep -v "gpu call\|kaiju_gpu pack" $O/err_$tag.txt | tail -16 | tee -a $O/steady.txt; }
run warm $W/reads24.fq 24 mem A=1 > /dev/null
run mem_24M $W/reads24.fq 24 mem A=1
run mem_408M $W/reads408.fq 408 mem A=1
run greedy_408M $W/reads408.fq 408 greedy A=1
run mem_gz_48M $W/reads48.fq.gz 48 mem A=1
run greedy_gz_48M $W/reads48.fq.gz 48 greedy A=1
# a second GPU-free look at the host side alone: parse only (no index, no GPU)
run parse_only_408M $W/reads408.fq 408 mem KAIJU_GPU_PARSE_ONLY=1
run parse_only_gz_48M $W/reads48.fq.gz 48 mem KAIJU_GPU_PARSE_ONLY=1
# output written (not /dev/null): the writer's share
t0=$(date +%s.%N); KAIJU_GPU_STAGE_TIMES=1 $CLI -t $W/nodes.dmp -f $W/db.fmi -i $W/reads408.fq -o $W/out.tsv -a mem 2> $O/err_mem_408M_file.txt; t1=$(date +%s.%N)
echo "== mem_408M to a file in /dev/shm: $(python3 -c "w=$t1-$t0; print(round(w,2),'s ->', round(408/w,1), 'M reads/s')")  $(wc -l < $W/out.tsv) lines, $(ls -la $W/out.tsv | awk '{print $5}') bytes" | tee -a $O/steady.txt
rm -rf $W
