"""Search time of the Greedy lane under different settings of the heavy-iteration gate (one index load, one read set):
   greedy_sweep.py <dir of prof_prepare.py> [n reads]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kaiju_amd import api, synth  # noqa: E402

W = sys.argv[1]
reads = np.load(f"{W}/reads.npy")
if len(sys.argv) > 2:
    reads = reads[: int(sys.argv[2])]
seqs, off = synth.pack_reads(reads)
index = api.Index(f"{W}/db.fmi")
ref = None
configs = [(g, w) for g in (3, 1, 7, 0) for w in (0, 8, 16, 24, 32, 48)]
if os.environ.get("SWEEP"):                       # e.g. SWEEP=3:0,3:32,0:0
    configs = [tuple(int(v) for v in c.split(":")) for c in os.environ["SWEEP"].split(",")]
for g, w in configs:
    os.environ["KAIJU_GPU_GREEDY_GATE"] = str(g)
    os.environ["KAIJU_GPU_GREEDY_WAITERS"] = str(w)
    clf = api.Classifier(index, api.default_params("greedy", seg=1))
    best = 1e9
    for _ in range(2):
        hits = clf.classify(seqs, off)
        best = min(best, clf.stats().ms_search)
    key = (hits["n_ids"].astype(np.int64).sum(), hits["best"].astype(np.int64).sum(), int(hits["taxid"].astype(np.uint64).sum() & 0xffffffffffff))
    if ref is None:
        ref = key
    print(f"gate {g} waiters {w:2d}: search {best:8.2f} ms  -> {len(reads)/best*1e3/1e6:6.2f} M reads/s  same results: {key == ref}", flush=True)
    del clf
