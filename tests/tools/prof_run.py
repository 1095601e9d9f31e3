"""Classify a prepared workload a few times (run this one under rocprofv3).
usage: prof_run.py <dir> <mode> <seg> <reps> [chunk]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kaiju_amd import api, synth  # noqa: E402

W, mode, seg, reps = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
reads = np.load(f"{W}/reads.npy")
if len(sys.argv) > 5:
    reads = reads[: int(sys.argv[5])]
seqs, off = synth.pack_reads(reads)
index = api.Index(os.environ.get("PROF_RUN_INDEX", f"{W}/db.fmi"))       # (PROF_RUN_INDEX: e.g. an image of the same index)
clf = api.Classifier(index, api.default_params(mode, seg=seg))
for _ in range(reps):
    hits = clf.classify(seqs, off)
    st = clf.stats()
    print(f"{mode} seg={seg} n={len(reads)}: translate {st.ms_translate:.2f} seg {st.ms_seg:.2f} [{st.n_seg_fragments}] "
          f"search {st.ms_search:.2f} retry {st.ms_retry:.2f} [{st.n_overflow_retries}] total {st.ms_total:.2f} ms "
          f"-> {len(reads)/st.ms_total*1e3:,.0f} reads/s", flush=True)
if os.environ.get("PROF_RUN_COUNTS"):
    # one more pass with the counting instantiation of the search kernel: memory steps per read
    clf.count_ops(True)
    clf.classify(seqs, off)
    oc = clf.op_counts()
    clf.count_ops(False)
    print("ops per read", {k: round(v / len(reads), 2) for k, v in oc.items() if v})
print("hit fraction", float((hits['n_ids'] > 0).mean()))
# (for A/B runs of kernel variants: the same reads must give the same records)
print("checksum", int(hits['n_ids'].astype(np.int64).sum()), int(hits['best'].astype(np.int64).sum()),
      int(hits['taxid'].astype(np.uint64).sum() & np.uint64(0xffffffffffff)))
