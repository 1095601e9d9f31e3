#!/bin/bash
# one PMC pass: pmc_one.sh <workdir> <mode> <seg> <outdir> <nreads> <counters...>
W=$1; MODE=$2; SEG=$3; OUT=$4; N=$5; shift 5
R=$(cd "$(dirname "$0")/../.." && pwd)
export TMPDIR=/tmp
cd /tmp
timeout 180 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT -o p -- python $R/tests/tools/prof_run.py $W $MODE $SEG 1 $N > $OUT.log 2>&1
python3 - "$OUT" <<'PY'
import csv,glob,sys,collections
res=collections.OrderedDict()
for f in glob.glob(sys.argv[1]+'/p_counter_collection.csv'):
    for row in csv.DictReader(open(f)):
        k=row['Kernel_Name'].split('(')[0]
        if not k.startswith('k_'): continue
        res[(k,row['Counter_Name'])]=(float(row['Counter_Value']), (int(row['End_Timestamp'])-int(row['Start_Timestamp']))/1e6)
for (k,c),v in res.items(): print(f"{k:14s} {c:34s} {v[0]:.4g}  dur {v[1]:.2f} ms")
PY
