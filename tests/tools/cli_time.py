"""End-to-end wall time of the drop-in CLI (and of the reference) on a prepared workload (GPU box):
   cli_time.py <workdir from prof_prepare.py> <nreads> [mode]"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
W, n = sys.argv[1], int(sys.argv[2])
mode = sys.argv[3] if len(sys.argv) > 3 else "mem"
reads = np.load(f"{W}/reads.npy")[:n]
n, L = reads.shape
fq = f"{W}/reads_{n}.fq"
if not os.path.exists(fq):
    t = time.time()
    names = np.char.add("@r", np.arange(n).astype(str)).astype("S")
    with open(fq, "wb") as f:
        qual = b"I" * L
        step = 200000
        for lo in range(0, n, step):
            hi = min(n, lo + step)
            parts = []
            for i in range(lo, hi):
                parts.append(names[i] + b"\n" + reads[i].tobytes() + b"\n+\n" + qual + b"\n")
            f.write(b"".join(parts))
    print(f"wrote {fq} ({os.path.getsize(fq)/1e6:.0f} MB) in {time.time()-t:.1f}s", flush=True)
cli = os.path.join(ROOT, "kaiju_amd", "bin", "kaiju")
for rep in range(2):
    t = time.time()
    subprocess.run([cli, "-t", f"{W}/nodes.dmp", "-f", f"{W}/db.fmi", "-i", fq, "-o", f"{W}/out_gpu.tsv", "-a", mode], check=True)
    dt = time.time() - t
    print(f"kaiju (GPU) -a {mode}: {dt:.2f} s wall incl. index load -> {n/dt:,.0f} reads/s end to end", flush=True)
ref = os.path.join(ROOT, "oracle", "_ref", "kaiju")
if os.path.exists(ref) and len(sys.argv) > 4:
    t = time.time()
    subprocess.run([ref, "-z", str(os.cpu_count()), "-t", f"{W}/nodes.dmp", "-f", f"{W}/db.fmi", "-i", fq, "-o", f"{W}/out_ref.tsv", "-a", mode], check=True)
    dt = time.time() - t
    print(f"reference kaiju -z {os.cpu_count()}: {dt:.2f} s -> {n/dt:,.0f} reads/s", flush=True)
    a = sorted(open(f"{W}/out_gpu.tsv").read().split("\n")); b = sorted(open(f"{W}/out_ref.tsv").read().split("\n"))
    print("outputs identical (sorted):", a == b)
