"""Prepare a profiling workload in a directory: viruses-like index + packed reads (no profiler)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kaiju_amd import mkfmi, synth  # noqa: E402

W = sys.argv[1]
nseq = int(sys.argv[2]) if len(sys.argv) > 2 else 680001
nreads = int(sys.argv[3]) if len(sys.argv) > 3 else 2000000
os.makedirs(W, exist_ok=True)
lines, leaves = synth.make_taxonomy()
synth.write_nodes_dmp(f"{W}/nodes.dmp", lines)
hard = os.environ.get("PREP_HARD") == "1"      # bench.py's hostile database: protein families, low-complexity inserts; reads with Ns
db = synth.make_db_hard(nseq=nseq, seed=4321, leaves=leaves) if hard else synth.make_db(nseq=nseq, seed=12345, leaves=leaves)
synth.write_fasta(db, f"{W}/db.faa")
mkfmi.build_fmi(f"{W}/db.faa", f"{W}/db.fmi", threads=0, exponent=3)
reads = synth.sprinkle_n(synth.make_reads(db, nreads, seed=779)) if hard else synth.make_reads(db, nreads, seed=777)
np.save(f"{W}/reads.npy", reads)
print("prepared", W, db.nseq, db.total_aa, reads.shape)
