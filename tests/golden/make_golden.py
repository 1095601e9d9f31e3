"""Generate the committed golden fixtures with the UNMODIFIED reference (oracle/_ref).

Run in the build container (needs /root/reference compiled into oracle/_ref by
`make -C oracle ref`):   python tests/golden/make_golden.py

Outputs (all small, committed):
  db.faa, nodes.dmp          synthetic protein database + taxonomy (kaiju_amd.synth, fixed seeds)
  db.fmi                     index built by the reference's kaiju-mkbwt -e 3 + kaiju-mkfmi
  reads.fq, pairs_{1,2}.fq   synthetic reads incl. hand-made edge cases
  ref_<mode>_<seg>[_pe].tsv  `kaiju -v -z 1` output of the reference binary
  kat_fm.npz                 FMindex / FMindexCurrent / get_suffix known answers (libkaijuref.so)
  kat_seg.json               SeqBufferSeg known answers (libkaijuref.so)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from kaiju_amd import synth  # noqa: E402
import pyoracle as po  # noqa: E402


def edge_reads(rng):
    """hand-made corner cases appended to the synthetic reads"""
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = []
    out.append(b"")                                   # empty read
    out.append(b"ACGTACGTAC")                         # far too short
    out.append(acgt[rng.integers(0, 4, 32)].tobytes())   # one below 3*m
    out.append(acgt[rng.integers(0, 4, 33)].tobytes())   # exactly 3*m
    out.append(acgt[rng.integers(0, 4, 34)].tobytes())
    out.append(b"A" * 150)                            # homopolymer: poly-K / poly-F, SEG food
    out.append(b"AAG" * 50)
    out.append(b"GCA" * 20 + b"GAA" * 30)
    out.append(acgt[rng.integers(0, 4, 150)].tobytes().lower())   # lower case
    r = acgt[rng.integers(0, 4, 150)].copy(); r[::17] = ord("N"); out.append(r.tobytes())
    r = acgt[rng.integers(0, 4, 150)].copy(); r[r == ord("T")] = ord("U"); out.append(r.tobytes())
    out.append(acgt[rng.integers(0, 4, 301)].tobytes())  # longer reads
    out.append(acgt[rng.integers(0, 4, 1000)].tobytes())
    return out


def write_fastq_list(reads, path, prefix="r"):
    with open(path, "wb") as f:
        for i, r in enumerate(reads):
            f.write(b"@" + f"{prefix}{i}".encode() + b"\n" + r + b"\n+\n" + b"I" * len(r) + b"\n")


def main():
    assert po.have_ref(), "build oracle/_ref first: make -C oracle ref"
    rng = np.random.default_rng(2024)
    lines, leaves = synth.make_taxonomy(4, 3, 4)
    synth.write_nodes_dmp(f"{HERE}/nodes.dmp", lines)
    db = synth.make_db(nseq=301, seed=4242, leaves=leaves, max_len=600)
    synth.write_fasta(db, f"{HERE}/db.faa")
    po.ref_build_index(f"{HERE}/db.faa", f"{HERE}/db", threads=2, exponent=3)
    # reads: synthetic + long-window reads made from the DB + edge cases
    reads = [r.tobytes() for r in synth.make_reads(db, 600, seed=99)]
    reads += [r.tobytes() for r in synth.make_reads(db, 60, seed=98, read_len=100)]
    reads += [r.tobytes() for r in synth.make_reads(db, 40, seed=97, read_len=250)]
    reads += edge_reads(rng)
    write_fastq_list(reads, f"{HERE}/reads.fq")
    m1, m2 = synth.make_pairs(db, 300, seed=55)
    p1 = [r.tobytes() for r in m1] + [b"ACGT" * 5, reads[3]]
    p2 = [r.tobytes() for r in m2] + [reads[5], b"ACG"]
    write_fastq_list(p1, f"{HERE}/pairs_1.fq", "p")
    write_fastq_list(p2, f"{HERE}/pairs_2.fq", "p")
    for mode in ("mem", "greedy"):
        for seg in (1, 0):
            po.ref_kaiju(f"{HERE}/nodes.dmp", f"{HERE}/db.fmi", f"{HERE}/reads.fq",
                         f"{HERE}/ref_{mode}_{seg}.tsv", mode=mode, seg=bool(seg))
            po.ref_kaiju(f"{HERE}/nodes.dmp", f"{HERE}/db.fmi", f"{HERE}/pairs_1.fq",
                         f"{HERE}/ref_{mode}_{seg}_pe.tsv", mode=mode, seg=bool(seg), reads2=f"{HERE}/pairs_2.fq")
    # extra parameter sets
    po.ref_kaiju(f"{HERE}/nodes.dmp", f"{HERE}/db.fmi", f"{HERE}/reads.fq", f"{HERE}/ref_greedy_e5_s50.tsv",
                 mode="greedy", extra=["-e", "5", "-s", "50", "-E", "10"])
    po.ref_kaiju(f"{HERE}/nodes.dmp", f"{HERE}/db.fmi", f"{HERE}/reads.fq", f"{HERE}/ref_greedy_e0.tsv",
                 mode="greedy", extra=["-e", "0"])
    po.ref_kaiju(f"{HERE}/nodes.dmp", f"{HERE}/db.fmi", f"{HERE}/reads.fq", f"{HERE}/ref_mem_m15.tsv",
                 mode="mem", extra=["-m", "15"])
    # function-level known answers
    R = po.RefLib()
    bw = R.read_indexes(f"{HERE}/db.fmi")
    bl = bw.len
    ks = np.unique(np.concatenate([rng.integers(0, bl + 1, 3000), np.arange(0, 300), np.arange(bl - 300, bl + 1)]))
    fm = np.zeros((len(ks), 21), dtype=np.int64)
    for a, k in enumerate(ks):
        for c in range(21):
            fm[a, c] = R.fmindex(bw, c, int(k))
    kk = ks[(ks < bl) & (ks >= bw.nseq)]      # get_suffix is only defined for rows of real suffixes
    cur = np.zeros((len(kk), 2), dtype=np.int64)
    suf = np.zeros((len(kk), 2), dtype=np.int64)
    for a, k in enumerate(kk):
        v, c = R.fmindex_current(bw, int(k))
        cur[a] = (v, c)
        suf[a] = R.get_suffix(bw, int(k))
    np.savez_compressed(f"{HERE}/kat_fm.npz", ks=ks, fm=fm, kk=kk, cur=cur, suf=suf)
    AA = synth.AA
    seg = []
    for it in range(1500):
        L = int(rng.integers(5, 130))
        mode = it % 4
        s = rng.choice(20, L)
        if mode == 1:
            s = rng.choice(rng.choice(20, 3), L)
        elif mode == 2:
            a = int(rng.integers(0, L)); b = int(rng.integers(a, L)); s[a:b] = rng.choice(rng.choice(20, 2), b - a)
        elif mode == 3:
            for _ in range(2):
                a = int(rng.integers(0, L)); b = min(L, a + int(rng.integers(5, 30)))
                s[a:b] = rng.choice(rng.choice(20, int(rng.integers(1, 4))), b - a)
        aa = "".join(AA[i] for i in s)
        seg.append([aa, [list(x) for x in R.seg(aa.encode())]])
    long_lc = "".join(AA[i] for i in rng.choice(rng.choice(20, 4), 400))
    seg.append([long_lc, [list(x) for x in R.seg(long_lc.encode())]])
    with open(f"{HERE}/kat_seg.json", "w") as f:
        json.dump(seg, f)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
