"""Adds the legs' records to profiles/traffic.json from the raw PMC passes of tests/tools/pmc_legs.sh:
   pmc_legs_collect.py <dir with <leg>/pass*/**/p_counter_collection.csv> <traffic.json to extend> [raw dir as named in the repo]
Same arithmetic as pmc_bench_collect.py: read requests x 128 B (32-B ones x 32 B) + write requests x 64 B (64-B ones) / x 32 B,
mean over the launches of the leg's size (the longest launches of the kernel in that run; the 100 k-read headline in front of
the leg falls out by duration)."""
import collections
import csv
import glob
import json
import os
import sys

src, out = sys.argv[1], sys.argv[2]
rawname = sys.argv[3] if len(sys.argv) > 3 else src
KERNELS = {"hard": {"hard": "k_mem", "hard_greedy": "k_greedy2"}, "wide": {"wide": "k_mem_wide2", "wide_greedy": "k_greedy2_wide"},
           "long": {"long": "k_mem"}, "protein": {"protein": "k_mem"}}
doc = json.load(open(out)) if os.path.exists(out) else {"measurements": []}
doc["measurements"] = [m for m in doc["measurements"] if not m.get("leg") or m["leg"] not in sum((list(v) for v in KERNELS.values()), [])]
for group, legs in KERNELS.items():
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(f"{src}/{group}/pass*/**/p_counter_collection.csv", recursive=True)):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            acc[k][row["Counter_Name"]].append((float(row["Counter_Value"]), (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6,
                                                int(row["Start_Timestamp"])))
    try:
        line = json.loads(open(f"{src}/{group}/pass1.json").read().strip().splitlines()[-1])
        detail = json.load(open(line["detail"])) if os.path.exists(line.get("detail", "")) else {}
    except Exception:  # noqa: BLE001
        detail = {}
    for leg, kern in legs.items():
        c = acc.get(kern)
        if not c:
            continue
        def big(xs):
            mx = max(x[1] for x in xs)
            xs = sorted([x for x in xs if x[1] > 0.3 * mx], key=lambda x: x[2])
            # long / protein run k_mem like the headline in front of them (--reads = the leg's size): the leg's launches are the
            # later half of the full-size ones (headline: 1 timed + 2 exclusive passes, then the leg: the same)
            return xs[len(xs) // 2:] if leg in ("hard", "long", "protein") and len(xs) >= 2 else xs
        def mean(name):
            xs = c.get(name)
            if not xs:
                return 0.0
            xs = big(xs)
            return sum(x[0] for x in xs) / len(xs)
        r, r32, w, w64 = mean("TCC_EA0_RDREQ_sum"), mean("TCC_EA0_RDREQ_32B_sum"), mean("TCC_EA0_WRREQ_sum"), mean("TCC_EA0_WRREQ_64B_sum")
        if r == 0.0 or w == 0.0:
            continue
        xs = big(c["TCC_EA0_RDREQ_sum"])
        units = None
        for key in (leg,):
            lr = detail.get(key) or {}
            units = (lr.get("roofline") or {}).get("units_per_launch") or lr.get("units_per_gpu_per_step")
        rec = {"leg": leg, "kernel": kern, "reads_per_launch": int(units) if units else 2000000, "launches_averaged": len(xs),
               "hbm_bytes_per_launch": (r - r32) * 128.0 + r32 * 32.0 + w64 * 64.0 + (w - w64) * 32.0,
               "counters_per_launch": {"TCC_EA0_RDREQ_sum": r, "TCC_EA0_RDREQ_32B_sum": r32, "TCC_EA0_WRREQ_sum": w, "TCC_EA0_WRREQ_64B_sum": w64,
                                       "TCC_HIT_sum": mean("TCC_HIT_sum") or None, "TCC_MISS_sum": mean("TCC_MISS_sum") or None},
               "kernel_ms_under_pmc": sum(x[1] for x in xs) / len(xs),
               "raw": f"{rawname}/{group}/pass*/p_counter_collection.csv",
               "method": "rocprofv3 --kernel-trace --pmc, one counter group per run of `bench.py --reads <leg size> --contexts 1 --steps 1 --warmup 0 "
                         f"--leg-steps 1 --no-cpu-baseline --legs {group}` (tests/tools/pmc_legs.sh); read requests x 128 B (32-B ones x 32 B), "
                         "write requests x 64 B (64-B ones) / x 32 B; mean over the launches of the leg's size"}
        doc["measurements"].append(rec)
        print(leg, kern, "%.2f GB per launch" % (rec["hbm_bytes_per_launch"] / 1e9), "in %.2f ms" % rec["kernel_ms_under_pmc"], "x", rec["launches_averaged"])
json.dump(doc, open(out, "w"), indent=1)
