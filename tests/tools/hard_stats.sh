#!/bin/bash
# rocprofv3 kernel summary of the hostile legs (bench.py --legs hard): which kernels the post-search bucket consists of there
O=${1:-gpurun_out/hard_stats}; mkdir -p $O
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/st -o s -- python $GRAFT_REPO_ROOT/bench.py --legs hard --steps 1 --warmup 0 --leg-steps 4 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench.json 2> $GRAFT_REPO_ROOT/$O/bench.err
cd $GRAFT_REPO_ROOT; cp $O/st/s_kernel_stats.csv $O/kernel_stats_hard.csv 2>/dev/null; cp $O/st/s_kernel_trace.csv $O/kernel_trace_hard.csv 2>/dev/null; rm -rf $O/st
python - <<PY
import csv, collections
# per kernel: the launches that belong to the hard legs = those after the hard index was built (the last k_suffix_walk)
rows = list(csv.DictReader(open("$O/kernel_trace_hard.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
walks = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_suffix_walk")]
start = walks[-1] if walks else 0
acc = collections.defaultdict(lambda: [0, 0.0])
greedy_seen = False
for r in rows[start:]:
    k = r["Kernel_Name"].split("(")[0]
    if k.startswith("k_greedy2"): greedy_seen = True
    key = ("greedy:" if greedy_seen else "mem:") + k
    acc[key][0] += 1; acc[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for k, (n, ms) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:24]:
    print("%-40s %3d launches %9.3f ms total %8.3f ms each" % (k, n, ms, ms / n))
PY
