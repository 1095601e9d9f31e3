#!/bin/bash
# Stage 1 with the codes of a block packed first and one frame at a time (-DKJ_S1_PACKED: 150 VGPRs instead of 214, three
# wavefronts per SIMD by registers, 2.75 by LDS): variants/cur against variants/s1p on 2 M reads of the profiling workload,
# alternating (the "translate" figure of prof_run.py is the kernel; the checksums must agree).  usage (lease.sh): sh:tests/tools/s1_packed_ab.sh
O=${1:-gpurun_out/s1_packed}; mkdir -p $O
R=$(cd "$(dirname "$0")/../.." && pwd); V=$R/kaiju_amd/variants
N=2000000
[ -f /tmp/kjw/reads.npy ] || python $R/tests/tools/prof_prepare.py /tmp/kjw 680001 $N > /dev/null 2>&1
for v in cur s1p cur s1p; do
  KAIJU_GPU_LIB=$V/libkaiju_gpu_$v.so python $R/tests/tools/prof_run.py /tmp/kjw mem 1 4 $N >> $O/$v.txt 2>&1
  echo "== $v $(grep -E 'translate' $O/$v.txt | tail -4 | awk '{printf "%s ", $5}') $(grep checksum $O/$v.txt | tail -1)"
done
