#!/bin/bash
# where the end-to-end time of the command line goes: cli_walls.sh <workdir of prof_prepare.py> <fastq> [brief]
W=$1; FQ=$2
R=$(cd "$(dirname "$0")/../.." && pwd)
CLI=$R/kaiju_amd/bin/kaiju
echo "transparent huge pages: $(cat /sys/kernel/mm/transparent_hugepage/enabled 2>/dev/null)"
run() { local tag=$1; local fq=$2; shift; shift; echo "== $tag"; local t0=$(date +%s.%N); env "$@" KAIJU_GPU_STAGE_TIMES=1 KAIJU_GPU_LOAD_TIMES=1 $CLI -t $W/nodes.dmp -f $W/db.fmi -i $fq -o $W/out_$tag.tsv -a mem 2> $W/err_$tag.txt; local t1=$(date +%s.%N); grep -v "gpu call\|kaiju_gpu pack" $W/err_$tag.txt; grep "gpu call" $W/err_$tag.txt | head -4; grep "gpu call" $W/err_$tag.txt | tail -1; echo "$tag: $(python3 -c "print(round($t1 - $t0, 3))") s"; }
rm -f $W/db.fmi.kjimg
run warm $FQ A=1
run default $FQ A=1
run default2 $FQ A=1
run no_hugepages $FQ KAIJU_GPU_NO_HUGEPAGES=1
run batches40 $FQ KAIJU_GPU_MAX_BATCHES=40
run batches6 $FQ KAIJU_GPU_MAX_BATCHES=6
run writeimage $FQ KAIJU_GPU_WRITE_IMAGE=1
run image $FQ A=1
cmp $W/out_default.tsv $W/out_image.tsv && echo "outputs identical"
cat $FQ $FQ $FQ > $W/reads_x3.fq
rm -f $W/db.fmi.kjimg
run x3_warm $W/reads_x3.fq A=1
run x3_default $W/reads_x3.fq A=1
run x3_batch500k $W/reads_x3.fq KAIJU_GPU_BATCH=500000
