#!/bin/bash
# The cheaper text comparison of the MEM lanes (K_TEXT: kTextCmp letters per round, one byte-align per dword) against the library of
# the commit before it (variants/head: 64 letters, a 64-bit mask and count-leading-zeros per 8 bytes), 32 / 48 / 64 letters per round
# (variants/t32, t48, t64), on the 190 M-row profiling index - as it is (narrow lane) and forced into the wide layout.
# usage (lease.sh): sh:tests/tools/text_cmp_ab.sh
O=${1:-gpurun_out/text_cmp}; mkdir -p $O
R=$(cd "$(dirname "$0")/../.." && pwd); V=$R/kaiju_amd/variants
N=4000000
[ -f /tmp/kjw/reads.npy ] || python $R/tests/tools/prof_prepare.py /tmp/kjw 680001 $N > /dev/null 2>&1
run() { tag=$1; shift; env "$@" python $R/tests/tools/prof_run.py /tmp/kjw mem 1 4 $N > $O/$tag.txt 2>&1; echo "== $tag $(grep -E 'search' $O/$tag.txt | awk '{printf "%s ", $10}') $(grep checksum $O/$tag.txt)"; }
for v in head t32 t48 t64 head; do run narrow_$v KAIJU_GPU_LIB=$V/libkaiju_gpu_$v.so; done
export KAIJU_GPU_FORCE_WIDE=31
run wide_head KAIJU_GPU_LIB=$V/libkaiju_gpu_head.so
for v in t32 t48 t64; do run wide_$v KAIJU_GPU_LIB=$V/libkaiju_gpu_$v.so KAIJU_GPU_TV_SHIFT=0; done
run wide_t48_tv1 KAIJU_GPU_LIB=$V/libkaiju_gpu_t48.so KAIJU_GPU_TV_SHIFT=1
