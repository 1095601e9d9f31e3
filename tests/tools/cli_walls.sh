#!/bin/bash
# where the end-to-end time of the command line goes: cli_walls.sh <workdir of prof_prepare.py> <fastq>
W=$1; FQ=$2
R=$(cd "$(dirname "$0")/../.." && pwd)
CLI=$R/kaiju_amd/bin/kaiju
run() { local tag=$1; shift; echo "== $tag"; local t0=$(date +%s.%N); env "$@" KAIJU_GPU_STAGE_TIMES=1 KAIJU_GPU_LOAD_TIMES=1 $CLI -t $W/nodes.dmp -f $W/db.fmi -i $FQ -o $W/out_$tag.tsv -a mem 2> $W/err_$tag.txt; local t1=$(date +%s.%N); grep -v "gpu call" $W/err_$tag.txt; grep "gpu call" $W/err_$tag.txt | head -6; grep "gpu call" $W/err_$tag.txt | tail -2; echo "$tag: $(python3 -c "print(round($t1 - $t0, 3))") s"; }
rm -f $W/db.fmi.kjimg
run warm A=1
run default A=1
run clean_exit KAIJU_GPU_CLEAN_EXIT=1
run writeimage KAIJU_GPU_WRITE_IMAGE=1
run image A=1
run image_big_batches KAIJU_GPU_BATCH=2000000
cmp $W/out_default.tsv $W/out_image.tsv && echo "outputs identical"
