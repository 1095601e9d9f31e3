"""Prepare the hostile workload of bench.py's `hard` legs in a directory (index of synth.make_db_hard + reads with Ns), in the
layout tests/tools/prof_run.py reads: db.fmi, reads.npy, nodes.dmp."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kaiju_amd import mkfmi, synth  # noqa: E402

W = sys.argv[1]
nseq = int(sys.argv[2]) if len(sys.argv) > 2 else 200001
nreads = int(sys.argv[3]) if len(sys.argv) > 3 else 2000000
os.makedirs(W, exist_ok=True)
lines, leaves = synth.make_taxonomy()
synth.write_nodes_dmp(f"{W}/nodes.dmp", lines)
db = synth.make_db_hard(nseq=nseq, seed=4321, leaves=leaves)
synth.write_fasta(db, f"{W}/db.faa")
mkfmi.build_fmi(f"{W}/db.faa", f"{W}/db.fmi", threads=0, exponent=3)
reads = synth.sprinkle_n(synth.make_reads(db, nreads, seed=779))
np.save(f"{W}/reads.npy", reads)
print("prepared", W, db.nseq, db.total_aa, reads.shape)
