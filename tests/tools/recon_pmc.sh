#!/bin/bash
# Raw PMC passes (rocprofv3 --kernel-trace --pmc, one counter group per run) + a kernel-trace summary for one
# prepared workload, MEM and Greedy; everything is written below <outdir> as CSV (copied to profiles/ by hand).
# usage: [PASSES="1 2"] [SKIP_TRACE=1] recon_pmc.sh <workdir> <outdir> [nreads] [modes]
W=$1; OUT=$2; N=${3:-2000000}; MODES=${4:-"mem greedy"}
R=$(cd "$(dirname "$0")/../.." && pwd)
export TMPDIR=/tmp
mkdir -p $OUT
OUT=$(cd $OUT && pwd)
cd /tmp
[ -f $W/db.fmi ] || python $R/tests/tools/prof_prepare.py $W 680001 $N > $OUT/prepare.log 2>&1
for MODE in $MODES; do
  if [ -z "$SKIP_TRACE" ]; then
  python $R/tests/tools/prof_run.py $W $MODE 1 2 $N > $OUT/${MODE}_plain.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${MODE}_trace -o t -- python $R/tests/tools/prof_run.py $W $MODE 1 2 $N > $OUT/${MODE}_trace.log 2>&1
  fi
  i=0
  for ctrs in \
    "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
    "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
    "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_FLAT" \
    "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
    "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum TCC_READ_sum" \
    "FETCH_SIZE" \
    "WRITE_SIZE TCC_WRITE_sum" ; do
    i=$((i+1))
    if [ -n "$PASSES" ] && ! echo " $PASSES " | grep -q " $i "; then continue; fi
    timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/${MODE}_pmc$i -o p -- python $R/tests/tools/prof_run.py $W $MODE 1 1 $N > $OUT/${MODE}_pmc$i.log 2>&1
    echo "$MODE pass $i rc=$? : $ctrs"
  done
done
# one table: kernel, counter, value, duration
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
rows = []
for f in sorted(glob.glob(out + '/*_pmc*/**/p_counter_collection.csv', recursive=True)):
    tag = f[len(out) + 1:].split('/')[0]
    acc = collections.OrderedDict()
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0]
        if not k.startswith('k_'):
            continue
        key = (tag, k, row['Counter_Name'])
        v = acc.setdefault(key, [0.0, 0.0, 0])
        v[0] += float(row['Counter_Value']); v[1] += (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e6; v[2] += 1
    for (tag, k, c), v in acc.items():
        rows.append((tag, k, c, v[0], v[1], v[2]))
with open(out + '/pmc_summary.csv', 'w') as fp:
    fp.write('pass,kernel,counter,value_sum_over_launches,duration_ms_sum,launches\n')
    for r in rows:
        fp.write('%s,%s,%s,%.6g,%.4f,%d\n' % r)
print(open(out + '/pmc_summary.csv').read())
PY
