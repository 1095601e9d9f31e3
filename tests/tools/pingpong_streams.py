"""Two classifier contexts ping-ponging chunks on two HIP streams (GPU box): does stage 1 / SEG of one chunk
overlap with the search of the other?   overlap_test.py <workdir> <nreads> <chunk> [mode]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kaiju_amd import api  # noqa: E402

W, n, chunk = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
mode = sys.argv[4] if len(sys.argv) > 4 else "mem"
dev = torch.device("cuda:0")
reads = np.load(f"{W}/reads.npy")[:n]
n, L = reads.shape
index = api.Index(f"{W}/db.fmi")
HIT = api.HIT_DTYPE.itemsize
d_seqs = torch.from_numpy(reads.reshape(-1)).to(dev)
d_out = torch.zeros(n * HIT, dtype=torch.uint8, device=dev)
bounds = [(lo, min(n, lo + chunk)) for lo in range(0, n, chunk)]
d_offs = []
for lo, hi in bounds:
    m = hi - lo
    o = np.empty(2 * m + 1, dtype=np.int64)
    o[0::2] = np.arange(m + 1, dtype=np.int64) * L
    o[1::2] = o[2::2]
    d_offs.append(torch.from_numpy(o).to(dev))
for nctx in (1, 2, 3):
    clfs = [api.Classifier(index, api.default_params(mode, seg=1)) for _ in range(nctx)]
    for c in clfs:
        c.set_max_read_length(L)
    streams = [torch.cuda.Stream(dev) for _ in range(nctx)]
    def step():
        for k, ((lo, hi), d_off) in enumerate(zip(bounds, d_offs)):
            c, s = clfs[k % nctx], streams[k % nctx]
            c.classify_device(d_seqs.data_ptr() + lo * L, (hi - lo) * L, d_off.data_ptr(), hi - lo,
                              d_out.data_ptr() + lo * HIT, paired=False, stream=s.cuda_stream)
        for s in streams:
            s.synchronize()
    step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 3
    print(f"{nctx} context(s), chunk {chunk}: {dt*1e3:.1f} ms per {n} reads -> {n/dt/1e6:.1f} M reads/s", flush=True)
    del clfs
