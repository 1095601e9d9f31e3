#!/bin/bash
# lease 29: what k_mem_post1 spends its 3.35 ms per 10 M reads on - builds without the trigger check / the locate + LCA / both
# (timing experiments: wrong results, no parity check), rocprofv3 kernel averages of the headline leg
O=$GRAFT_REPO_ROOT/gpurun_out/r06_l29; mkdir -p $O
V=$GRAFT_REPO_ROOT/kaiju_amd/variants
for v in cur postnotrig postnoloc postnone; do
  lib=; [ $v != cur ] && lib="KAIJU_GPU_LIB=$V/libkaiju_gpu_$v.so"
  ( cd /tmp && env $lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$v -o s -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --legs "" --steps 5 > $O/bench_$v.json 2> $O/bench_$v.err )
  cp $O/stats_$v/s_kernel_stats.csv $O/kernel_stats_$v.csv 2>/dev/null; rm -rf $O/stats_$v
  echo "== $v"; grep "k_mem_post\|k_seg\|k_mem_second\|k_mem(" $O/kernel_stats_$v.csv | cut -d, -f1-4 | sed 's/(.*)"/"/' | cut -c1-120
done
