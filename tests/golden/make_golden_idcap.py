"""Golden vectors for kaiju -v when ids_from_SI's limit (more than max_match_ids = 20 distinct taxon ids,
ConsumerThread.cpp:805-807) ends the traversal before the last fragment: the reference has pushed the peptide of EVERY fragment
that holds a longest match by then (:580-590), and later matches add neither ids nor accessions.  A family of 30 identical
proteins under 30 taxa + unrelated proteins; reads (single and paired) whose fragments hold equally long matches in the family
and in an unrelated protein, in both orders.  Generated with the UNMODIFIED reference (oracle/_ref, `make -C oracle ref`):
    python tests/golden/make_golden_idcap.py
Outputs (small, committed) under tests/golden/idcap/: db.faa, db.fmi, reads.fq, pairs_{1,2}.fq, ref_<mode>_<seg>[_pe].tsv;
nodes.dmp is the golden set's (../nodes.dmp)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "idcap")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from kaiju_amd import synth  # noqa: E402
import pyoracle as po  # noqa: E402

AA = "ACDEFGHIKLMNPQRSTVWY"
CODON = {"A": "GCT", "C": "TGT", "D": "GAT", "E": "GAA", "F": "TTT", "G": "GGT", "H": "CAT", "I": "ATT", "K": "AAA", "L": "CTT",
         "M": "ATG", "N": "AAT", "P": "CCT", "Q": "CAA", "R": "CGT", "S": "TCT", "T": "ACT", "V": "GTT", "W": "TGG", "Y": "TAT"}


def back(pep):
    return "".join(CODON[c] for c in pep)


def main():
    assert po.have_ref(), "build oracle/_ref first: make -C oracle ref"
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(606)
    _, leaves = synth.make_taxonomy(4, 3, 4)                     # (the golden set's tree: ../nodes.dmp)
    leaves = [int(x) for x in leaves]

    def prot(n):
        return "".join(AA[i] for i in rng.integers(0, 20, n))
    fam, fam2, solo, solo2 = prot(120), prot(90), prot(120), prot(100)
    with open(f"{OUT}/db.faa", "w") as f:
        for k in range(30):                                       # 30 copies under 30 taxa: one match, 30 rows, 30 ids
            f.write(f">FAM{k:03d}.1_{leaves[k % len(leaves)]}\n{fam}\n")
        for k in range(25):                                       # a second family (25 taxa)
            f.write(f">FAN{k:03d}.1_{leaves[(k + 7) % len(leaves)]}\n{fam2}\n")
        f.write(f">SOLO1.1_{leaves[40 % len(leaves)]}\n{solo}\n>SOLO2.1_{leaves[41 % len(leaves)]}\n{solo2}\n")
        for k in range(40):
            f.write(f">RND{k:03d}.1_{leaves[(3 * k) % len(leaves)]}\n{prot(int(rng.integers(60, 200)))}\n")
    po.ref_build_index(f"{OUT}/db.faa", f"{OUT}/db", threads=2, exponent=3)
    stop = "TAA"
    reads = []
    for a, b in ((fam, solo), (solo, fam), (fam, fam2), (fam2, fam), (solo, solo2), (fam2, solo2)):
        for la in (20, 25):
            # two fragments of la residues each in frame 0, a stop codon between them: equally long longest matches
            reads.append((back(a[10:10 + la]) + stop + back(b[30:30 + la])).encode())
            # three fragments: the capped family in the middle
            reads.append((back(b[5:5 + la]) + stop + back(a[40:40 + la]) + stop + back(solo2[50:50 + la])).encode())
    reads.append(back(fam[0:45]).encode())                        # one fragment, one match of 30 rows
    reads.append(back(solo[0:45]).encode())
    with open(f"{OUT}/reads.fq", "wb") as f:
        for i, r in enumerate(reads):
            f.write(b"@c%d\n" % i + r + b"\n+\n" + b"I" * len(r) + b"\n")
    p1, p2 = [], []
    for a, b in ((fam, solo), (solo, fam), (fam, fam2), (fam2, solo2)):
        p1.append(back(a[20:60]).encode()); p2.append(back(b[20:60]).encode())
    for nm, rr in (("pairs_1.fq", p1), ("pairs_2.fq", p2)):
        with open(f"{OUT}/{nm}", "wb") as f:
            for i, r in enumerate(rr):
                f.write(b"@d%d\n" % i + r + b"\n+\n" + b"I" * len(r) + b"\n")
    nodes = os.path.join(HERE, "nodes.dmp")
    for mode in ("mem", "greedy"):
        for seg in (1, 0):
            po.ref_kaiju(nodes, f"{OUT}/db.fmi", f"{OUT}/reads.fq", f"{OUT}/ref_{mode}_{seg}.tsv", mode=mode, seg=bool(seg))
            po.ref_kaiju(nodes, f"{OUT}/db.fmi", f"{OUT}/pairs_1.fq", f"{OUT}/ref_{mode}_{seg}_pe.tsv", mode=mode, seg=bool(seg),
                         reads2=f"{OUT}/pairs_2.fq")
    n_c = sum(1 for line in open(f"{OUT}/ref_mem_0.tsv") if line.startswith("C"))
    print("written", OUT, len(reads), "reads,", n_c, "classified in MEM mode")


if __name__ == "__main__":
    main()
