#!/bin/bash
# What the text-verification code costs and gains in the wide MEM lane, on the 190 M-row profiling index forced into the wide
# layout: the library of the last commit without it (variants/head) against the current one (variants/cur) without text arrays,
# with the text position of every row and of every second row; then .fmi against image load.   usage (lease.sh): sh:tests/tools/wide_text_small_ab.sh
O=${1:-gpurun_out/wide_text_small}; mkdir -p $O
R=$(cd "$(dirname "$0")/../.." && pwd); V=$R/kaiju_amd/variants
N=4000000
[ -f /tmp/kjw/reads.npy ] || python $R/tests/tools/prof_prepare.py /tmp/kjw 680001 $N > /dev/null 2>&1
export KAIJU_GPU_FORCE_WIDE=31 PROF_RUN_COUNTS=1
run() { tag=$1; shift; env "$@" python $R/tests/tools/prof_run.py /tmp/kjw mem 1 3 $N > $O/$tag.txt 2>&1; echo "== $tag"; grep -E "search|checksum|ops per read" $O/$tag.txt | tail -3 | cut -c1-400; }
run head_notext KAIJU_GPU_LIB=$V/libkaiju_gpu_head.so KAIJU_GPU_NO_TEXT=1
run cur_notext KAIJU_GPU_LIB=$V/libkaiju_gpu_cur.so KAIJU_GPU_NO_TEXT=1
run cur_tv0 KAIJU_GPU_LIB=$V/libkaiju_gpu_cur.so KAIJU_GPU_TV_SHIFT=0
run cur_tv1 KAIJU_GPU_LIB=$V/libkaiju_gpu_cur.so KAIJU_GPU_TV_SHIFT=1
KAIJU_GPU_LIB=$V/libkaiju_gpu_head.so python -c "
import sys; sys.path.insert(0, '$R')
from kaiju_amd import api
api.write_index_image('/tmp/kjw/db.fmi', '/tmp/kjw/db.img')" > $O/write_image.txt 2>&1
run head_notext_image KAIJU_GPU_LIB=$V/libkaiju_gpu_head.so KAIJU_GPU_NO_TEXT=1 PROF_RUN_INDEX=/tmp/kjw/db.img KAIJU_GPU_STREAM_PIECE_MB=64
run head_notext_again KAIJU_GPU_LIB=$V/libkaiju_gpu_head.so KAIJU_GPU_NO_TEXT=1
