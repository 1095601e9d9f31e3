"""TEST TOOL (host only): the wide lanes of kj_core.h (host emulation, tests/emu) against the oracle on an index of MORE THAN
2^33 ROWS - beyond what the GPU suite's 2^32-row index exercises (sample numbers above 2^30, rows above 33 bits, several count
bases) - before GPU minutes are spent on a refseq-class index.  A small database replicated by kaiju_build_fmi_replicated
(tests/test_mkfmi_pin.py pins that builder); needs ~55 GB of host memory and 17 GB under the work directory.

   python tests/tools/big_rows_emu.py [workdir] [rows_log2 = 33.02] [reads = 1500]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from kaiju_amd import mkfmi, synth  # noqa: E402
import pyoracle as po  # noqa: E402
import util  # noqa: E402


def main():
    W = sys.argv[1] if len(sys.argv) > 1 else "/tmp/kaiju_big_rows"
    rows = 2.0 ** float(sys.argv[2]) if len(sys.argv) > 2 else 2.0 ** 33.02
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
    os.makedirs(W, exist_ok=True)
    lines, leaves = synth.make_taxonomy(6, 5, 5)
    db = synth.make_db(nseq=3701, seed=4242, leaves=leaves, max_len=900)
    copies = int(rows / (db.total_aa + db.nseq)) + 1
    faa, fmi, nodes = f"{W}/base.faa", f"{W}/big.fmi", f"{W}/nodes.dmp"
    synth.write_fasta(db, faa)
    synth.write_nodes_dmp(nodes, lines)
    t0 = time.time()
    if not os.path.exists(fmi):
        mkfmi.build_fmi_replicated(faa, fmi + ".tmp", copies, threads=0, exponent=3, copy_taxids=np.asarray(leaves, dtype=np.uint64))
        os.replace(fmi + ".tmp", fmi)
    print(f"[big rows] {db.total_aa} aa x {copies} copies: .fmi {os.path.getsize(fmi)/1e9:.2f} GB ({time.time()-t0:.0f}s)", flush=True)
    os.environ["KAIJU_EMU_DROP_FILE"] = "1"
    os.environ["KAIJU_EMU_NO_TEXT"] = "1"
    E = util.Emu()
    t0 = time.time()
    h = E.load(fmi)
    print(f"[big rows] emulation index packed ({time.time()-t0:.0f}s)", flush=True)
    O = po.Oracle()
    t0 = time.time()
    oix, otax = O.load_fmi(fmi), O.load_nodes(nodes)
    bwtlen = O.lib.ko_bwtlen(oix)
    print(f"[big rows] oracle loaded: bwtlen {bwtlen} = 2^{np.log2(bwtlen):.3f} ({time.time()-t0:.0f}s)", flush=True)
    assert bwtlen > 2 ** 33
    bad_total = 0
    for mode, paired, k in (("mem", False, n), ("greedy", False, n // 2), ("mem", True, n // 2), ("greedy", True, n // 4)):
        if paired:
            m1, m2 = synth.make_pairs(db, k, seed=778)
            seqs, off = synth.pack_reads(m1, m2)
        else:
            seqs, off = synth.pack_reads(synth.make_reads(db, k, seed=777))
        t0 = time.time()
        hits, nretry = E.classify(h, util.gp(mode), seqs, off, paired=paired)
        te = time.time() - t0
        t0 = time.time()
        oh = O.classify(oix, otax, O.params(mode, seg=1, use_evalue=0), seqs, off, paired=paired)
        bad = [i for i in range(len(oh)) if not util.same_hit(oh[i], hits[i])]
        bad_total += len(bad)
        print(f"[big rows] {mode}{' pairs' if paired else ''}: {len(oh)} checked, {len(bad)} mismatches {bad[:5]}, with hit "
              f"{int((hits['n_ids'] > 0).sum())}, retries {nretry} (emulation {te:.0f}s, oracle {time.time()-t0:.0f}s)", flush=True)
    print("[big rows] OK" if bad_total == 0 else "[big rows] MISMATCHES", flush=True)
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())
