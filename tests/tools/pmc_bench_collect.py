"""profiles/traffic.json from the raw PMC passes of tests/tools/pmc_bench.sh:
   pmc_bench_collect.py <dir with pass*/p_counter_collection.csv> <out traffic.json> [raw dir as named in the repo]
Per kernel and launch size: read requests x 128 B (a miss fetches a whole line: profiles/r01_randbench_calibration.txt;
32-B requests x 32 B) + write requests x 64 B (64-B ones; the others 32 B), averaged over the launches of the timed
size.  The counting instantiations (k_*_count) and the second search of the lazy SEG flow (few reads) are listed apart."""
import collections
import csv
import glob
import json
import os
import sys

src, out = sys.argv[1], sys.argv[2]
rawname = sys.argv[3] if len(sys.argv) > 3 else src
acc = collections.defaultdict(lambda: collections.defaultdict(list))      # kernel -> counter -> [(value, dur_ms)]
for f in sorted(glob.glob(src + "/pass*/**/p_counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("k_")]
    # the legs run one after the other (headline, greedy, paired): a k_mem launch after the first Greedy search belongs to the paired leg
    g0 = min([int(r["Start_Timestamp"]) for r in rows if r["Kernel_Name"].startswith("k_greedy2(")] or [1 << 62])
    for row in rows:
        k = row["Kernel_Name"].split("(")[0]
        acc[k][row["Counter_Name"]].append((float(row["Counter_Value"]), (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6,
                                            int(row["Start_Timestamp"]) > g0))
# bench.py --legs greedy,paired --steps 1 --warmup 0: per leg the timed steps (one launch of 10 M reads / of 5 M pairs each),
# the exclusive pass (the same launch again) and the counting pass
legs = {"mem": ("k_mem", False), "greedy": ("k_greedy2", False), "paired": ("k_mem", True)}
meas = []
def big(launches):
    """launches of the full chunk size: the long ones (the lazy SEG flow searches a few per cent of the reads again)"""
    if not launches:
        return []
    mx = max(d for _, d, _ in launches)
    return [x for x in launches if x[1] > 0.5 * mx]
for name, (kern, paired) in legs.items():
    c = acc.get(kern)
    if not c:
        continue
    def sel(counter):
        xs = big(c.get(counter, []))
        if kern == "k_mem":
            xs = [x for x in c.get(counter, []) if x[2] == paired]
            xs = big(xs)
        return xs
    rd, rd32 = sel("TCC_EA0_RDREQ_sum"), sel("TCC_EA0_RDREQ_32B_sum")
    wr, wr64 = sel("TCC_EA0_WRREQ_sum"), sel("TCC_EA0_WRREQ_64B_sum")
    if not rd or not wr:
        continue
    mean = lambda xs: sum(v for v, _, _ in xs) / len(xs)
    r, r32, w, w64 = mean(rd), mean(rd32) if rd32 else 0.0, mean(wr), mean(wr64) if wr64 else 0.0
    rec = {"mode": "greedy" if name == "greedy" else "mem", "paired": paired, "seg": 1, "nseq": 680001, "reads_per_launch": int(os.environ.get("PMC_PAIR_LAUNCH", 5000000)) if paired else (10000000 if name == "greedy" else int(os.environ.get("PMC_MEM_LAUNCH", 10000000))),
           "kernel": kern, "launches_averaged": len(rd),
           "hbm_bytes_per_launch": (r - r32) * 128.0 + r32 * 32.0 + w64 * 64.0 + (w - w64) * 32.0,
           "counters_per_launch": {"TCC_EA0_RDREQ_sum": r, "TCC_EA0_RDREQ_32B_sum": r32, "TCC_EA0_WRREQ_sum": w, "TCC_EA0_WRREQ_64B_sum": w64,
                                   "TCC_HIT_sum": mean(sel("TCC_HIT_sum")) if sel("TCC_HIT_sum") else None,
                                   "TCC_MISS_sum": mean(sel("TCC_MISS_sum")) if sel("TCC_MISS_sum") else None,
                                   "FETCH_SIZE": mean(sel("FETCH_SIZE")) if sel("FETCH_SIZE") else None,
                                   "WRITE_SIZE": mean(sel("WRITE_SIZE")) if sel("WRITE_SIZE") else None},
           "kernel_ms_under_pmc": sum(d for _, d, _ in rd) / len(rd),
           "raw": rawname + "/pass*/p_counter_collection.csv",
           "method": "rocprofv3 --kernel-trace --pmc, one counter group per run of `bench.py --contexts 1 --steps 1 --warmup 0 "
                     "--no-cpu-baseline --legs greedy,paired` (tests/tools/pmc_bench.sh); read requests x 128 B (32-B ones x 32 B), "
                     "write requests x 64 B (64-B ones) / x 32 B; mean over the launches of full size"}
    meas.append(rec)
if os.environ.get("PMC_MERGE"):
    # keep what is there; a new record replaces the one of the same leg / mode / pairing / launch size
    key = lambda m: (m.get("leg"), m.get("mode"), bool(m.get("paired")), m["reads_per_launch"])
    old = json.load(open(os.environ["PMC_MERGE"]))["measurements"]
    new = {key(m) for m in meas}
    meas = [m for m in old if key(m) not in new] + meas
json.dump({"measurements": meas}, open(out, "w"), indent=1)
for m in meas:
    print(m["kernel"], m.get("leg") or ("paired" if m.get("paired") else m.get("mode")), "%.1f GB per launch" % (m["hbm_bytes_per_launch"] / 1e9), "in %.2f ms" % m["kernel_ms_under_pmc"],
          "=> %.2f TB/s" % (m["hbm_bytes_per_launch"] / m["kernel_ms_under_pmc"] / 1e9))
# the other kernels: totals per launch, for DESIGN.md
summary = {}
for k, c in acc.items():
    summary[k] = {cn: {"mean_per_launch": sum(v for v, _, _ in xs) / len(xs), "launches": len(xs), "mean_ms": sum(d for _, d, _ in xs) / len(xs)} for cn, xs in c.items()}
json.dump(summary, open(out.replace(".json", "_all_kernels.json"), "w"), indent=1)
