#!/bin/bash
# PMC passes over one prepared workload (each pass is its own run; --pmc only with --kernel-trace)
# usage: pmc_passes.sh <workdir> <mode> <seg> <outdir> [nreads]
W=$1; MODE=$2; SEG=$3; OUT=$4; N=${5:-2000000}
R=$(cd "$(dirname "$0")/../.." && pwd)
export TMPDIR=/tmp
cd /tmp
i=0
for ctrs in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
  "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" \
  "FETCH_SIZE TCC_READ_sum" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TA_BUSY_avr" \
  "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  timeout 180 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/pass$i -o p -- python $R/tests/tools/prof_run.py $W $MODE $SEG 1 $N > $OUT/pass$i.log 2>&1
  echo "pass $i rc=$? : $ctrs"
done
