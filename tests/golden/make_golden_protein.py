"""Protein-input golden lines (reference binaries oracle/_ref/kaiju -p and oracle/_ref/kaijup, built by
`make -C oracle ref`) on the committed golden index:   python tests/golden/make_golden_protein.py
  prot.fa                   protein reads: stretches of the golden proteins with substitutions, letters that are no
                            amino acid (B J O U X Z) and '*' (removed by strip(), util.cpp:25-32), lower case,
                            low-complexity inserts, lengths around the -m gate, multi-line FASTA records, names with
                            blanks (kaiju cuts them, kaijup keeps them)
  refp_<mode>_<seg>.tsv     `kaiju -p -v -z 1`
  refpx_<mode>[_v].tsv      `kaijup -z 1 [-v]`
  regions_prot.fa, regions_nuc.fa   reads whose single fragment holds more low-complexity regions than a record of the SEG
                            pass (15): the exact pass of the kernels (tests/util.py: many_region_reads)
  refr_prot_<mode>.tsv, refr_nuc_<mode>.tsv   `kaiju [-p] -v -z 1` on them"""
import os
import subprocess

import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref")
AA = "ARNDCQEGHILKMFPSTWYV"


def proteins():
    out, cur = [], []
    with open(f"{HERE}/db.faa") as f:
        for line in f:
            if line.startswith(">"):
                if cur:
                    out.append("".join(cur))
                cur = []
            else:
                cur.append(line.strip())
    out.append("".join(cur))
    return out


def make_reads(n=400, seed=99):
    rng = np.random.default_rng(seed)
    prots = proteins()
    reads = ["", "ACDEFGHIKL", "ACDEFGHIKLM", "ACDEFGHIKLMN", "WWWWWW", "WWWWWWX" * 3, "acdefghiklmnpqrstvwy" * 3,
             "K" * 80, "QN" * 40, "ACDEFGHIKLMNPQRSTVWY*ACDEFGHIKLMNPQRSTVWY", "XXXXXXXXXXXXXXXX", "MKT" * 30 + "AAAAAAAAAAAAAAAA" + prots[3][:60]]
    while len(reads) < n:
        L = int(rng.choice([8, 11, 12, 25, 50, 100, 200, 400, 900]))
        if rng.random() < 0.15:
            s = "".join(rng.choice(list(AA), L))
        else:
            parts = []
            while sum(len(x) for x in parts) < L:
                p = prots[int(rng.integers(0, len(prots)))]
                a = int(rng.integers(0, max(1, len(p) - 12)))
                parts.append(p[a: a + int(rng.integers(8, 250))])
                r = rng.random()
                if r < 0.25:
                    parts.append("BJOUXZ"[int(rng.integers(0, 6))] * int(rng.integers(1, 3)))
                elif r < 0.3:
                    parts.append("*")
                elif r < 0.38:
                    parts.append(AA[int(rng.integers(0, 20))] * int(rng.integers(10, 30)))
            s = list("".join(parts)[:L])
            for _ in range(int(rng.integers(0, 6))):
                s[int(rng.integers(0, len(s)))] = (AA + "XB")[int(rng.integers(0, 22))]
            s = "".join(s)
        if rng.random() < 0.1:
            s = s.lower()
        reads.append(s)
    return reads


def main():
    reads = make_reads()
    with open(f"{HERE}/prot.fa", "w") as f:
        for i, s in enumerate(reads):
            name = f"p{i}" + (" some description/1" if i % 7 == 3 else "")
            f.write(f">{name}\n")
            if i % 5 == 2 and len(s) > 70:                     # multi-line record
                for k in range(0, len(s), 60):
                    f.write(s[k: k + 60] + "\n")
            else:
                f.write(s + "\n")
    for mode in ("mem", "greedy"):
        for seg in (1, 0):
            out = f"{HERE}/refp_{mode}_{seg}.tsv"
            cmd = [f"{REF}/kaiju", "-p", "-t", f"{HERE}/nodes.dmp", "-f", f"{HERE}/db.fmi", "-i", f"{HERE}/prot.fa", "-a", mode,
                   "-z", "1", "-v", "-o", out] + ([] if seg else ["-X"])
            subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
            print(out, sum(1 for _ in open(out)))
        for v in (False, True):
            out = f"{HERE}/refpx_{mode}{'_v' if v else ''}.tsv"
            cmd = [f"{REF}/kaijup", "-f", f"{HERE}/db.fmi", "-i", f"{HERE}/prot.fa", "-a", mode, "-z", "1", "-o", out] + (["-v"] if v else [])
            subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
            print(out, sum(1 for _ in open(out)))


def regions():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import util
    prot, nuc = util.many_region_reads()
    for name, reads in (("prot", prot), ("nuc", nuc)):
        with open(f"{HERE}/regions_{name}.fa", "wb") as f:
            for i, s in enumerate(reads):
                f.write(b">g%d\n%s\n" % (i, s))
        for mode in ("mem", "greedy"):
            out = f"{HERE}/refr_{name}_{mode}.tsv"
            cmd = [f"{REF}/kaiju", "-t", f"{HERE}/nodes.dmp", "-f", f"{HERE}/db.fmi", "-i", f"{HERE}/regions_{name}.fa", "-a", mode,
                   "-z", "1", "-v", "-o", out] + (["-p"] if name == "prot" else [])
            subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
            print(out, sum(1 for _ in open(out)))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "regions":
        regions()
        sys.exit(0)
    main()
    regions()
