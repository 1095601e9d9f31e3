#!/bin/bash
# End-to-end time of the drop-in command line on 24 M reads from the page cache, with the wall-clock marks and the per-stage CPU
# times (KAIJU_GPU_STAGE_TIMES): where the time between "start" and "exit" goes.   usage (lease.sh): sh:tests/tools/cli_bench.sh
O=${1:-gpurun_out/cli}; mkdir -p $O
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
t = time.time()
with open("$W/reads24.fq", "wb") as f:
    for k in range(6):
        f.write(bench.fastq_bytes(synth.make_reads(db, 4_000_000, seed=777 + k), first=4_000_000 * k))
print("24 M reads written", time.time() - t, flush=True)
PY
ls -la $W > $O/files.txt
run() { local tag=$1; shift; local t0=$(date +%s.%N); env "$@" KAIJU_GPU_STAGE_TIMES=1 KAIJU_GPU_LOAD_TIMES=1 $CLI -t $W/nodes.dmp -f $W/db.fmi -i $W/reads24.fq -o $W/out_$tag.tsv -a mem 2> $O/err_$tag.txt; local rc=$?; local t1=$(date +%s.%N)
  echo "== $tag rc=$rc: $(python3 -c "print(round($t1 - $t0, 3), 's ->', round(24 / ($t1 - $t0), 1), 'M reads/s end to end')")"; grep -v "gpu call\|kaiju_gpu pack" $O/err_$tag.txt | tail -14; }
run warm A=1 > /dev/null
run default A=1
run default2 A=1
run writeimage KAIJU_GPU_WRITE_IMAGE=1 > /dev/null
run image A=1
run image2 A=1
run greedy_image A=1 2>/dev/null
cmp $W/out_default.tsv $W/out_image.tsv && echo "outputs identical (default / image)"
wc -l $W/out_default.tsv
rm -rf $W
