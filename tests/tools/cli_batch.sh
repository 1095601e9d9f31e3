#!/bin/bash
# The command line's batch size (KAIJU_GPU_BATCH; default: a sixteenth of the sample, 250 000 .. 1 M reads): 96 M reads from
# /dev/shm, Greedy and MEM.  The search kernels are persistent lanes (131 072 of them): a batch of 1 M reads is 7.6 reads a lane
# and its tail - the longest read of the batch - is not hidden.     usage (lease.sh): sh:tests/tools/cli_batch.sh
O=${1:-$GRAFT_REPO_ROOT/gpurun_out/cli_batch}; mkdir -p $O
R=$(cd "$(dirname "$0")/../.." && pwd)
W=/dev/shm/kjcli; mkdir -p $W
CLI=$R/kaiju_amd/bin/kaiju
python - <<PY
import sys, time
sys.path.insert(0, "$R")
import bench
from kaiju_amd import synth, mkfmi
lines, leaves = synth.make_taxonomy(); synth.write_nodes_dmp("$W/nodes.dmp", lines)
db = synth.make_db(nseq=680001, seed=12345, leaves=leaves)
synth.write_fasta(db, "$W/db.faa"); mkfmi.build_fmi("$W/db.faa", "$W/db.fmi", threads=0, exponent=3)
with open("$W/reads24.fq", "wb") as f:
    for k in range(6):
        f.write(bench.fastq_bytes(synth.make_reads(db, 4_000_000, seed=777 + k), first=4_000_000 * k))
PY
for k in 1 2 3 4; do cat $W/reads24.fq; done > $W/reads96.fq
run() { local tag=$1 mode=$2; shift 2; local t0=$(date +%s.%N)
  env "$@" KAIJU_GPU_STAGE_TIMES=1 $CLI -t $W/nodes.dmp -f $W/db.fmi -i $W/reads96.fq -o $W/out_$tag.tsv -a $mode 2> $O/err_$tag.txt; local rc=$?; local t1=$(date +%s.%N)
  echo "== $tag rc=$rc: $(python3 -c "w=$t1-$t0; print(round(w, 2), 's wall ->', round(96 / w, 1), 'M reads/s end to end;', round(96 / max(w - 0.6, 1e-9), 1), 'M reads/s without 0.6 s of start-up')") $(md5sum < $W/out_$tag.tsv | cut -c1-12)" | tee -a $O/batch.txt; }
run warm mem A=1 > /dev/null
for b in default 1000000 2000000 4000000 8000000; do
  if [ $b = default ]; then run greedy_$b greedy A=1; run mem_$b mem A=1
  else run greedy_$b greedy KAIJU_GPU_BATCH=$b; run mem_$b mem KAIJU_GPU_BATCH=$b; fi
done
grep -h "stage\|\[gz\|cpu" $O/err_greedy_default.txt $O/err_greedy_4000000.txt | cut -c1-300 | head -20 >> $O/batch.txt
rm -rf $W
