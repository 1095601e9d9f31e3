"""kaiju -v in Greedy mode (first-generation lanes): what the main pass and the retry pass cost for given scratch sizes
   g1_probe.py <workdir from prof_prepare.py> <nreads>      (KAIJU_GPU_G1_POOL / _G1_MATCH / KAIJU_GPU_RETRY_BLOCKS in the environment)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kaiju_amd import api  # noqa: E402

W, n = sys.argv[1], int(sys.argv[2])
reads = np.load(f"{W}/reads.npy")[:n]
n, L = reads.shape
idx = api.Index(f"{W}/db.fmi")
clf = api.Classifier(idx, api.default_params("greedy", seg=1))
seqs = np.ascontiguousarray(reads).reshape(-1)
off = np.zeros(2 * n + 1, dtype=np.uint64)
off[1::2] = np.arange(1, n + 1, dtype=np.uint64) * L
off[2::2] = off[1::2]
t = time.time()
clf.classify_verbose_raw(seqs, off)
el0 = time.time() - t
t = time.time()
clf.classify_verbose_raw(seqs, off)
el = time.time() - t
print(f"first call {el0*1e3:.0f} ms (buffers of the context allocated)", flush=True)
st = clf.stats()
print({k: os.environ.get(k) for k in ("KAIJU_GPU_G1_POOL", "KAIJU_GPU_G1_MATCH", "KAIJU_GPU_RETRY_BLOCKS")}, f"{n} reads: call {el*1e3:.0f} ms, search {st.ms_search:.1f} ms, "
      f"retry pass {st.ms_retry:.1f} ms, reads in the retry pass {st.n_overflow_retries}", flush=True)
