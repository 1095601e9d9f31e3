#!/bin/bash
# round 6, lease 5: (a) what the per-chunk gather costs on ONE GPU (world-1 RCCL gather behind the search kernels),
# (b) PMC traffic of the legs hard / wide / long / protein
O=$1
B="python bench.py --legs , --steps 8 --warmup 2 --no-cpu-baseline --no-ref-ops"
run() { tag=$1; shift; echo "== $tag: $*"; env "$@" > $O/gather_$tag.json 2> $O/gather_$tag.err; echo "rc=$?"; python - <<P
import json
try:
    d=json.loads(open("$O/gather_$tag.json").read().strip().splitlines()[-1]); print("$tag", round(d["value"]/1e6,1), "M reads/s", round(d["ms_per_step"],3), "ms/step", d["config"].get("gather"), d["config"].get("gather_by"))
except Exception as e: print("$tag failed", e)
P
}
run none           A=1 timeout 600 $B
run lib_1ctx       KAIJU_DIST_FORCE_INIT=1 timeout 600 $B --gather lib
run torch_1ctx     KAIJU_DIST_FORCE_INIT=1 timeout 600 $B --gather torch
run none_2ctx      A=1 timeout 600 $B --contexts 2 --chunk 2500000
run lib_2ctx       KAIJU_DIST_FORCE_INIT=1 timeout 600 $B --gather lib --contexts 2 --chunk 2500000
run torch_2ctx     KAIJU_DIST_FORCE_INIT=1 timeout 600 $B --gather torch --contexts 2 --chunk 2500000
run none_b         A=1 timeout 600 $B
# the timeline: where RCCL's kernel runs relative to the next chunk's search kernel
( cd /tmp && export TMPDIR=/tmp && KAIJU_DIST_FORCE_INIT=1 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --legs , --steps 2 --warmup 1 --no-cpu-baseline --no-ref-ops --gather lib --contexts 2 --chunk 2500000 > $GRAFT_REPO_ROOT/$O/trace.json 2> $GRAFT_REPO_ROOT/$O/trace.err )
python - <<P
import csv, glob
rows=[]
for f in glob.glob("$O/trace/**/t_kernel_trace.csv", recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40], r.get("Stream_Id","?"), r.get("Queue_Id","?")) for r in csv.DictReader(open(f))]
for f in glob.glob("$O/trace/**/t_memory_copy_trace.csv", recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "memcpy " + r.get("Direction","?") , "-", "-") for r in csv.DictReader(open(f))]
rows.sort()
if rows:
    t0 = rows[0][0]
    # the last 120 events of the run = the timed steps
    with open("$O/timeline_tail.txt", "w") as f:
        for s,e,k,st,q in rows[-160:]:
            f.write(f"{(s-t0)/1e6:12.3f} {(e-s)/1e6:9.3f} ms  q{q} s{st}  {k}\n")
    print(open("$O/timeline_tail.txt").read()[-3000:])
P
rm -rf $O/trace
bash tests/tools/pmc_legs.sh $O/pmc_legs hard wide long protein > $O/pmc_legs.log 2>&1; tail -8 $O/pmc_legs.log
cp profiles/traffic.json $O/traffic.json
python tests/tools/pmc_legs_collect.py $O/pmc_legs $O/traffic.json profiles/r06_pmc_legs
find $O/pmc_legs -name "*.csv" -size +20M -delete
