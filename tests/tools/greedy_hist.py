"""Workload statistics of the Greedy search (host emulation built with -DKJ_HIST): sizes of the match
lists, the queue and the best-SI lists.  usage: greedy_hist.py <workdir from prof_prepare.py> [nreads]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import util  # noqa: E402

_x = tuple(os.environ.get("HIST_DEFINES", "").split())
emu = util.Emu("/tmp/libkaiju_kernel_emu_hist" + "".join("_" + d for d in _x) + ".so", defines=("KJ_HIST",) + _x)
emu.lib.emu_hist.restype = C.POINTER(C.c_ulonglong * (16 * 64))
W = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
seg = int(sys.argv[3]) if len(sys.argv) > 3 else 1
reads = np.load(f"{W}/reads.npy")[:n]
L = reads.shape[1]
off = np.zeros(2 * n + 1, dtype=np.uint64)
off[0::2] = np.arange(n + 1, dtype=np.uint64) * L
off[1::2] = np.arange(1, n + 1, dtype=np.uint64) * L
h = emu.load(f"{W}/db.fmi")
out, nretry = emu.classify(h, util.gp("greedy", seg=seg), reads.reshape(-1), off, caps=(16, 4096, 1024))
hist = np.array(emu.lib.emu_hist().contents, dtype=np.uint64).reshape(16, 64)
names = ["matches per searched original", "matches per searched variant", "queue length at push",
         "push needs a shift (1) / appends (0)", "nbest after eval", "pool items per read", "nbest at finish", "v2: live queue entries at push"]
for i, nm in enumerate(names):
    row = hist[i]
    tot = row.sum()
    nz = np.nonzero(row)[0]
    print(f"{nm}: total {tot}")
    print("   " + " ".join(f"{k}:{row[k] / max(tot, 1):.4f}" for k in nz[:64]))
kinds = "STEP KMER PROBE LF1 LF2 SA VMULTI META FRAG FILL POPITEM MLOAD WAIT IDLE EXIT".split()
print("v2 iterations per read by kind:", {k: round(float(hist[6][i]) / n, 1) for i, k in enumerate(kinds)}, "total", round(float(hist[6][:15].sum()) / n, 1))
# how much of the slow part runs on ONE database row: rows of the interval at a multi-letter step, of the
# variants popped, and the UpdateSI steps of the searches by kind
per = lambda row, k: [round(float(x) / n, 2) for x in hist[row][:k]]
print("multi-letter steps per read on 1 / 2-4 / more rows:", per(8, 3), " by substitutions so far:", per(9, 6))
print("variants popped per read on 1 / 2-4 / more rows:", per(10, 3), " by substitutions:", per(11, 6))
print("UpdateSI steps per read: originals / variants on one row / variants on more:", per(12, 3))
print("retries", nretry, "classified", int((out['n_ids'] > 0).sum()))
