#!/bin/bash
# round 6, lease 3: where the row-pool Greedy lane spends its time (section profile), and how it depends on the threads per block
O=$1
for t in 512 384 256; do
  echo "=== threads $t"
  KAIJU_GPU_G3_THREADS=$t VARIANTS="cur" bash tests/tools/mem_variants.sh run $O/t$t greedy 4000000
done
KAIJU_GPU_G3_THREADS=512 VARIANTS="prof" bash tests/tools/mem_variants.sh run $O/p512 greedy 4000000
KAIJU_GPU_G3_THREADS=256 VARIANTS="prof" bash tests/tools/mem_variants.sh run $O/p256 greedy 4000000
grep "kj prof" $O/p512/prof.txt | head -30
echo; grep "kj prof" $O/p256/prof.txt | head -30
