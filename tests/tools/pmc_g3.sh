#!/bin/bash
# PMC passes over the third-generation Greedy kernels (prof_run.py workload): pmc_g3.sh <workdir> <outdir> <nreads>
W=$1; OUT=$2; N=${3:-2000000}
R=$(cd "$(dirname "$0")/../.." && pwd)
export TMPDIR=/tmp
mkdir -p $OUT; OUT=$(cd $OUT && pwd)
cd /tmp
i=0
for ctrs in \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
  "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/pmc$i -o p -- python $R/tests/tools/prof_run.py $W greedy 1 1 $N > $OUT/pmc$i.log 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.OrderedDict()
for f in sorted(glob.glob(out + '/pmc*/**/p_counter_collection.csv', recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0]
        if not k.startswith('k_g3'): continue
        v = acc.setdefault((k, row['Counter_Name']), [0.0, 0.0, 0])
        v[0] += float(row['Counter_Value']); v[1] += (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e6; v[2] += 1
with open(out + '/pmc_summary.csv', 'w') as fp:
    fp.write('kernel,counter,value_sum_over_launches,duration_ms_sum,launches\n')
    for (k, c), v in acc.items(): fp.write('%s,%s,%.6g,%.4f,%d\n' % (k, c, v[0], v[1], v[2]))
print(open(out + '/pmc_summary.csv').read())
PY
