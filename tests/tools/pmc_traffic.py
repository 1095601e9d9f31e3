"""HBM traffic of the search kernel from two rocprofv3 PMC passes (run on the GPU box):
   pmc_traffic.py <workdir from prof_prepare.py> <mode> <seg> <reads per launch> <nseq> <out.json>
Appends {"mode", "seg", "nseq", "reads_per_launch", "hbm_bytes_per_launch", ...} to the JSON file."""
import csv
import json
import os
import subprocess
import sys

W, mode, seg, n, nseq, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
kern = "k_mem" if mode == "mem" else "k_greedy2"
vals = {}
for tag, counters in (("rd", ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum"]), ("wr", ["TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"])):
    d = f"/tmp/pmc_traffic_{tag}"
    subprocess.run(["rm", "-rf", d])
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(["timeout", "240", "rocprofv3", "--kernel-trace", "--pmc"] + counters +
                   ["--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
                    os.path.join(ROOT, "tests", "tools", "prof_run.py"), W, mode, str(seg), "1", str(n)],
                   cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for row in csv.DictReader(open(os.path.join(d, "p_counter_collection.csv"))):
        if row["Kernel_Name"].split("(")[0] == kern:
            vals[row["Counter_Name"]] = float(row["Counter_Value"])
            vals["dur_ms_" + tag] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
rd, rd32 = vals.get("TCC_EA0_RDREQ_sum", 0.0), vals.get("TCC_EA0_RDREQ_32B_sum", 0.0)
wr = vals.get("TCC_EA0_WRREQ_sum", 0.0)
rec = {"mode": mode, "seg": seg, "nseq": nseq, "reads_per_launch": n, "kernel": kern,
       "hbm_bytes_per_launch": (rd - rd32) * 128.0 + rd32 * 32.0 + wr * 64.0,
       "counters": vals,
       "method": "rocprofv3 --pmc, separate passes; read requests x 128 B (a miss fetches a whole line, "
                 "profiles/r01_randbench_calibration.txt), write requests x 64 B"}
data = {"measurements": []}
if os.path.exists(out):
    data = json.load(open(out))
data["measurements"] = [m for m in data["measurements"]
                        if not (m["mode"] == mode and m["seg"] == seg and m["nseq"] == nseq and m["reads_per_launch"] == n)]
data["measurements"].append(rec)
json.dump(data, open(out, "w"), indent=1)
print(json.dumps(rec))
