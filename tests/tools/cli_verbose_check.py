"""-v output of the drop-in CLI vs the reference binary on a prepared workload (GPU box):
   cli_verbose_check.py <workdir from prof_prepare.py> <nreads>"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
W, n = sys.argv[1], int(sys.argv[2])
reads = np.load(f"{W}/reads.npy")[:n]
n, L = reads.shape
fq = f"{W}/v_{n}.fq"
with open(fq, "wb") as f:
    f.write(b"".join(b"@r%d\n" % i + reads[i].tobytes() + b"\n+\n" + b"I" * L + b"\n" for i in range(n)))
cli = os.path.join(ROOT, "kaiju_amd", "bin", "kaiju")
ref = os.path.join(ROOT, "oracle", "_ref", "kaiju")
for mode in os.environ.get("VB_MODES", "mem,greedy").split(","):     # (VB_MODES=mem: one mode)
    for seg in ([], ["-X"]):
        t = time.time()
        subprocess.run([cli, "-t", f"{W}/nodes.dmp", "-f", f"{W}/db.fmi", "-i", fq, "-o", f"{W}/pg.tsv", "-a", mode] + seg, check=True)
        tp = time.time() - t                                   # (the same run without -v: what the verbose columns cost)
        t = time.time()
        subprocess.run([cli, "-t", f"{W}/nodes.dmp", "-f", f"{W}/db.fmi", "-i", fq, "-o", f"{W}/vg.tsv", "-a", mode, "-v"] + seg, check=True)
        tg = time.time() - t
        t1 = None
        if True:                                               # (the first-generation lanes, which wrote the columns until round 6)
            t = time.time()
            subprocess.run([cli, "-t", f"{W}/nodes.dmp", "-f", f"{W}/db.fmi", "-i", fq, "-o", f"{W}/vg1.tsv", "-a", mode, "-v"] + seg, check=True,
                           env=dict(os.environ, KAIJU_GPU_VERBOSE_LANE="v1"))
            t1 = time.time() - t
            if open(f"{W}/vg1.tsv").read() != open(f"{W}/vg.tsv").read():
                print("  KAIJU_GPU_VERBOSE_LANE=v1 writes OTHER lines than the default", flush=True)
        t = time.time()
        subprocess.run([ref, "-z", str(os.cpu_count()), "-t", f"{W}/nodes.dmp", "-f", f"{W}/db.fmi", "-i", fq, "-o", f"{W}/vr.tsv",
                        "-a", mode, "-v"] + seg, check=True)
        tr = time.time() - t
        a = sorted(open(f"{W}/vg.tsv").read().split("\n"))
        b = sorted(open(f"{W}/vr.tsv").read().split("\n"))
        bad = [(x, y) for x, y in zip(a, b) if x != y]
        print(f"-a {mode} {' '.join(seg)} -v: {n} reads, GPU {tg:.2f}s" + (f" (first-generation lanes {t1:.2f}s)" if t1 else "") + f" (without -v {tp:.2f}s; all include ~0.5 s of index load), reference {tr:.1f}s, "
              f"differing lines (all seven columns): {len(bad) + abs(len(a) - len(b))}", flush=True)
        for x, y in bad[:3]:
            print("  gpu:", x[:300]); print("  ref:", y[:300])
