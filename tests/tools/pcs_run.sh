#!/bin/bash
# PC sampling of the Greedy (or MEM) search kernel: pcs_run.sh <out dir under the repo> <mode> <reads>
R=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$R/$1; MODE=${2:-greedy}; N=${3:-2000000}
mkdir -p $OUT
python $R/tests/tools/prof_prepare.py /tmp/kjw 680001 $N > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
export KAIJU_GPU_LIB=$R/kaiju_amd/libkaiju_gpu_g.so
try() {
  local tag=$1; shift
  rm -rf /tmp/pcs_$tag
  timeout 170 rocprofv3 --pc-sampling-beta-enabled "$@" --kernel-trace --output-format csv -d /tmp/pcs_$tag -o p -- python $R/tests/tools/prof_run.py /tmp/kjw $MODE 1 2 > $OUT/$tag.log 2>&1
  echo "$tag rc=$?" >> $OUT/summary.txt
  find /tmp/pcs_$tag -name "*.csv" -exec ls -la {} \; >> $OUT/summary.txt
  local f=$(find /tmp/pcs_$tag -name "*pc_sampling*.csv" | head -1)
  if [ -n "$f" ] && [ $(wc -l < $f) -gt 100 ]; then python $R/tests/tools/pcs_aggregate.py $f $OUT/$tag >> $OUT/summary.txt 2>&1; find /tmp/pcs_$tag -name "*kernel_trace.csv" -exec cp {} $OUT/${tag}_kernel_trace.csv \; ; return 0; fi
  return 1
}
try host_trap --pc-sampling-unit time --pc-sampling-method host_trap --pc-sampling-interval 1000 || \
try stochastic --pc-sampling-unit cycles --pc-sampling-method stochastic --pc-sampling-interval 1048576
tail -5 $OUT/*.log
cat $OUT/summary.txt
