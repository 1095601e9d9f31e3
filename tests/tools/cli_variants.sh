#!/bin/bash
# end-to-end wall time of the drop-in command line in a few configurations: cli_variants.sh <workdir of prof_prepare.py> <fastq>
W=$1; FQ=$2
R=$(cd "$(dirname "$0")/../.." && pwd)
CLI=$R/kaiju_amd/bin/kaiju
run() { local tag=$1; shift; local t0=$(date +%s.%N); env "$@" KAIJU_GPU_STAGE_TIMES=1 $CLI -t $W/nodes.dmp -f $W/db.fmi -i $FQ -o $W/out_$tag.tsv -a mem 2> $W/err_$tag.txt; local t1=$(date +%s.%N); echo "$tag: $(python3 -c "print(round($t1 - $t0, 3))") s"; grep -i "stage\|read \|parse\|gpu\|format\|write" $W/err_$tag.txt | head -8; }
rm -f $W/db.fmi.kjimg
run default A=1
run default_again A=1
run noprescan KAIJU_GPU_PRESCAN=0
run writeimage KAIJU_GPU_WRITE_IMAGE=1
run image A=1
run image_again A=1
cmp $W/out_default.tsv $W/out_image.tsv && echo "outputs identical"
