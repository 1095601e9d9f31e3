"""Deterministic synthetic workloads for the Kaiju classification path.

No network is available, so the NCBI-derived indexes of BASELINE.json cannot be
fetched; these generators produce a *viruses-like* protein database, a matching
``nodes.dmp`` and 150-bp reads with the statistics described in SURVEY.md §8(d):

* database: ``nseq`` proteins, lengths ``clip(Gamma(2, 140), 30, 3000)``, residues
  i.i.d. from UniProt-like background frequencies, 35 % of the sequences are
  mutated copies (1 % / 5 % / 15 % substitutions) of earlier ones, headers
  ``>WPnnnnnnnnn.1_<taxid>`` (>= 12 bytes, see SURVEY.md §7 on the mkbwt buffer bug);
* taxonomy: root 1 -> families -> genera -> species (leaf taxa);
* reads: 70 % back-translated from a random 51-aa window of a random protein
  (uniform synonymous codons, 0-2 nt frame offset, {0,0,1,3,8} random nt
  substitutions, 50 % reverse-complemented), 30 % uniform random ACGT.

Everything is numpy-vectorised so that 10 M reads are generated in tens of seconds.
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np

AA = "ACDEFGHIKLMNPQRSTVWY"  # index alphabet order (kaiju-makedb: -a ACDEFGHIKLMNPQRSTVWY)
# approximate UniProt background frequencies, same order as AA
_BG = np.array([8.25, 1.37, 5.45, 6.75, 3.86, 7.07, 2.27, 5.96, 5.84, 9.66,
                2.42, 4.06, 4.70, 3.93, 5.53, 6.56, 5.34, 6.87, 1.08, 2.92])
_BG = _BG / _BG.sum()

# standard genetic code -> synonymous codon lists per amino acid
_CODON_AA = "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSS*CWCLFLF"  # ACGT order
_NT = "ACGT"


def _codon_tables():
    syn = {a: [] for a in AA}
    for i, aa in enumerate(_CODON_AA):
        if aa != "*":
            syn[aa].append(i)
    nsyn = np.array([len(syn[a]) for a in AA], dtype=np.int64)
    tab = np.zeros((20, 6), dtype=np.int64)
    for k, a in enumerate(AA):
        for j, c in enumerate(syn[a]):
            tab[k, j] = c
    return nsyn, tab


_NSYN, _CODTAB = _codon_tables()


@dataclass
class SynthDB:
    """A synthetic protein database held as code arrays (0..19 = AA order)."""
    codes: np.ndarray      # uint8, concatenated residues
    offsets: np.ndarray    # int64[nseq+1]
    taxids: np.ndarray     # int64[nseq]
    names: list            # header strings without '>'

    @property
    def nseq(self):
        return len(self.taxids)

    @property
    def total_aa(self):
        return int(self.offsets[-1])


def make_taxonomy(n_families=50, genera_per_family=10, species_per_genus=10):
    """Return (lines for nodes.dmp, array of leaf taxon ids)."""
    lines = ["1\t|\t1\t|\tno rank\t|"]
    leaves = []
    fid = 10
    gid = 1000
    sid = 100000
    for _ in range(n_families):
        lines.append(f"{fid}\t|\t1\t|\tfamily\t|")
        for _ in range(genera_per_family):
            lines.append(f"{gid}\t|\t{fid}\t|\tgenus\t|")
            for _ in range(species_per_genus):
                lines.append(f"{sid}\t|\t{gid}\t|\tspecies\t|")
                leaves.append(sid)
                sid += 1
            gid += 1
        fid += 1
    return lines, np.array(leaves, dtype=np.int64)


def write_nodes_dmp(path, lines):
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")


def make_db(nseq=680001, seed=12345, leaves=None, min_len=30, max_len=3000,
            gamma_shape=2.0, gamma_scale=140.0, frac_copies=0.35):
    """Generate the protein database.  Avoids the two reference index bugs
    (SURVEY.md §7): nseq % 8 != 0 and bwtlen % 65536 < 65408, by trimming the
    last sequence if necessary."""
    rng = np.random.default_rng(seed)
    if leaves is None:
        _, leaves = make_taxonomy()
    lens = np.clip(rng.gamma(gamma_shape, gamma_scale, size=nseq), min_len, max_len).astype(np.int64)
    is_copy = rng.random(nseq) < frac_copies
    is_copy[: max(1, nseq // 100)] = False
    src = np.zeros(nseq, dtype=np.int64)
    idx = np.nonzero(is_copy)[0]
    src[idx] = (rng.random(len(idx)) * idx).astype(np.int64)  # an earlier sequence
    # copies of copies are allowed: resolve lengths in order
    for i in idx:
        lens[i] = lens[src[i]]
    # steer clear of the reference's rank bug for bwtlen % 65536 >= 65408 (or == 0)
    bwtlen = int(lens.sum()) + nseq
    while bwtlen % 65536 >= 65408 or bwtlen % 65536 == 0:
        j = np.nonzero(~is_copy)[0][-1]
        lens[j] += 1
        bwtlen += 1
    offsets = np.zeros(nseq + 1, dtype=np.int64)
    np.cumsum(lens, out=offsets[1:])
    codes = rng.choice(20, size=int(offsets[-1]), p=_BG).astype(np.uint8)
    rates = rng.choice(np.array([0.01, 0.05, 0.15]), size=nseq)
    for i in idx:                      # sequential: copies may chain
        s = src[i]
        seg = codes[offsets[s]:offsets[s + 1]].copy()
        mut = rng.random(len(seg)) < rates[i]
        seg[mut] = rng.integers(0, 20, size=int(mut.sum()), dtype=np.uint8)
        codes[offsets[i]:offsets[i + 1]] = seg
    taxids = leaves[rng.integers(0, len(leaves), size=nseq)]
    # mutated copies mostly stay close in the taxonomy: same genus block 70 % of the time
    near = is_copy & (rng.random(nseq) < 0.7)
    for i in np.nonzero(near)[0]:
        base = taxids[src[i]]
        taxids[i] = base - (base % 10) + rng.integers(0, 10)
    names = [f"WP{n:09d}.1_{t}" for n, t in enumerate(taxids)]
    return SynthDB(codes=codes, offsets=offsets, taxids=taxids, names=names)


def make_db_large(nseq, seed=12345, leaves=None, min_len=30, max_len=3000, gamma_shape=2.0, gamma_scale=140.0,
                  frac_copies=0.35, chunk=1 << 22):
    """The database of make_db() for sizes where its per-sequence Python loops take too long (refseq-class indexes,
    >= 2^32 rows): same length / residue / copy-rate statistics, fully vectorised.  Copies are made of ORIGINAL
    sequences only (no chains), the source lies anywhere before the copy; taxa as in make_db().  Not the same
    sequences as make_db() for the same seed."""
    rng = np.random.default_rng(seed)
    if leaves is None:
        _, leaves = make_taxonomy()
    lens = np.clip(rng.gamma(gamma_shape, gamma_scale, size=nseq), min_len, max_len).astype(np.int64)
    is_copy = rng.random(nseq) < frac_copies
    is_copy[: max(1, nseq // 100)] = False
    orig_idx = np.nonzero(~is_copy)[0]
    copy_idx = np.nonzero(is_copy)[0]
    # source = a random original in front of the copy
    n_before = np.searchsorted(orig_idx, copy_idx)                     # originals with a smaller index
    src = orig_idx[(rng.random(len(copy_idx)) * n_before).astype(np.int64)]
    lens[copy_idx] = lens[src]
    bwtlen = int(lens.sum()) + nseq
    while bwtlen % 65536 >= 65408 or bwtlen % 65536 == 0:              # the reference's rank bug, see make_db()
        lens[orig_idx[-1]] += 1
        bwtlen += 1
    offsets = np.zeros(nseq + 1, dtype=np.int64)
    np.cumsum(lens, out=offsets[1:])
    total = int(offsets[-1])
    codes = np.empty(total, dtype=np.uint8)
    cum = np.cumsum(_BG)
    cum[-1] = 1.0
    step = 1 << 27
    for lo in range(0, total, step):
        hi = min(total, lo + step)
        codes[lo:hi] = np.searchsorted(cum, rng.random(hi - lo, dtype=np.float32), side="right").astype(np.uint8)
    np.minimum(codes, 19, out=codes)
    rates = rng.choice(np.array([0.01, 0.05, 0.15]), size=len(copy_idx))
    for lo in range(0, len(copy_idx), chunk):                          # copies, a few million sequences at a time
        ci, sr, rt = copy_idx[lo:lo + chunk], src[lo:lo + chunk], rates[lo:lo + chunk]
        ln = lens[ci]
        tot = int(ln.sum())
        within = np.arange(tot, dtype=np.int64) - np.repeat(np.cumsum(ln) - ln, ln)
        seg = codes[np.repeat(offsets[sr], ln) + within]
        mut = rng.random(tot, dtype=np.float32) < np.repeat(rt, ln).astype(np.float32)
        seg[mut] = rng.integers(0, 20, size=int(mut.sum()), dtype=np.uint8)
        codes[np.repeat(offsets[ci], ln) + within] = seg
    taxids = leaves[rng.integers(0, len(leaves), size=nseq)]
    near = np.zeros(nseq, dtype=bool)
    near[copy_idx] = rng.random(len(copy_idx)) < 0.7
    nc = near[copy_idx]
    base = taxids[src[nc]]
    taxids[copy_idx[nc]] = base - (base % 10) + rng.integers(0, 10, size=int(nc.sum()))
    names = _LazyNames(taxids)
    return SynthDB(codes=codes, offsets=offsets, taxids=taxids, names=names)


def make_db_hard(nseq=200001, seed=4321, leaves=None, fam_lo=50, fam_hi=500, frac_lowcomplexity=0.05,
                 min_len=30, max_len=1500, gamma_shape=2.0, gamma_scale=140.0):
    """A database that is NOT i.i.d. (bench.py's `hard` leg): protein FAMILIES of fam_lo .. fam_hi near-identical members (0.5 - 3 %
    substitutions against the family's founder, so that exact matches have intervals of dozens to hundreds of rows and the
    members' taxa differ), and low-complexity inserts (runs of one to three letters, 10 - 40 residues) in frac_lowcomplexity of
    the sequences - what the skip rules, the queue / match-list capacities and the SEG region bound of the kernels were NOT
    tuned on.  Same return type as make_db()."""
    rng = np.random.default_rng(seed)
    if leaves is None:
        _, leaves = make_taxonomy()
    sizes = []
    while sum(sizes) < nseq:
        sizes.append(int(rng.integers(fam_lo, fam_hi + 1)))
    sizes[-1] -= sum(sizes) - nseq
    if sizes[-1] <= 0:
        sizes.pop()
        sizes[-1] += nseq - sum(sizes)
    seqs, taxids = [], []
    for fam, size in enumerate(sizes):
        L = int(np.clip(rng.gamma(gamma_shape, gamma_scale), min_len, max_len))
        founder = rng.choice(20, size=L, p=_BG).astype(np.uint8)
        members = np.tile(founder, (size, 1))
        rates = rng.choice(np.array([0.005, 0.01, 0.03]), size=size)
        mut = rng.random((size, L)) < rates[:, None]
        mut[0] = False
        members[mut] = rng.integers(0, 20, size=int(mut.sum()), dtype=np.uint8)
        genus = leaves[rng.integers(0, len(leaves))]
        tx = genus - (genus % 10) + rng.integers(0, 10, size=size)          # a family mostly stays in one genus ...
        far = rng.random(size) < 0.1
        tx[far] = leaves[rng.integers(0, len(leaves), size=int(far.sum()))]  # ... with a tenth of its members elsewhere
        lc = rng.random(size) < frac_lowcomplexity
        for m in range(size):
            row = members[m]
            if lc[m]:
                k = int(rng.integers(1, 4))
                unit = rng.integers(0, 20, size=k, dtype=np.uint8)
                ins = np.tile(unit, int(rng.integers(10, 41)) // k + 1)[: int(rng.integers(10, 41))]
                at = int(rng.integers(0, L + 1))
                row = np.concatenate([row[:at], ins, row[at:]])
            seqs.append(row)
        taxids.append(tx)
    lens = np.array([len(x) for x in seqs], dtype=np.int64)
    # steer clear of the reference's two latent index bugs as make_db() does (nseq % 8 != 0 is the caller's choice of nseq)
    bwtlen = int(lens.sum()) + nseq
    while bwtlen % 65536 >= 65408 or bwtlen % 65536 == 0:
        seqs[-1] = np.concatenate([seqs[-1], rng.integers(0, 20, size=1, dtype=np.uint8)])
        lens[-1] += 1
        bwtlen += 1
    offsets = np.zeros(nseq + 1, dtype=np.int64)
    np.cumsum(lens, out=offsets[1:])
    taxids = np.concatenate(taxids).astype(np.int64)
    names = _LazyNames(taxids) if nseq > 1_000_000 else [f"WP{n:09d}.1_{t}" for n, t in enumerate(taxids)]
    return SynthDB(codes=np.concatenate(seqs), offsets=offsets, taxids=taxids, names=names)


def sprinkle_n(reads: np.ndarray, seed=99, frac_reads=0.05, max_n=4):
    """reads with ambiguous bases: frac_reads of the rows get 1 .. max_n letters replaced by 'N' (in place; returns reads)"""
    rng = np.random.default_rng(seed)
    n, L = reads.shape
    sel = np.nonzero(rng.random(n) < frac_reads)[0]
    cnt = rng.integers(1, max_n + 1, size=len(sel))
    for k in range(max_n):
        rows = sel[cnt > k]
        reads[rows, rng.integers(0, L, size=len(rows))] = ord("N")
    return reads


class _LazyNames:
    """names[i] of a large database without materialising millions of Python strings"""

    def __init__(self, taxids):
        self.taxids = taxids

    def __getitem__(self, i):
        return f"WP{i:09d}.1_{self.taxids[i]}"

    def __len__(self):
        return len(self.taxids)


def write_fasta_large(db: SynthDB, path, chunk=1 << 20):
    """write_fasta() without a Python loop per sequence: headers `>WPnnnnnnnnn.1_tttttt` of fixed width (taxon ids of
    make_taxonomy() have six digits)"""
    aa = np.frombuffer(AA.encode(), dtype=np.uint8)
    H = 1 + 2 + 9 + 3 + 6 + 1                                          # ">WP" 9 digits ".1_" 6 digits "\n"
    with open(path, "wb") as f:
        for lo in range(0, db.nseq, chunk):
            hi = min(db.nseq, lo + chunk)
            m = hi - lo
            ln = np.diff(db.offsets[lo:hi + 1])
            assert int(db.taxids[lo:hi].max()) < 1000000 and int(db.taxids[lo:hi].min()) >= 100000 and hi <= 10 ** 9
            rec_len = ln + H + 1
            rec_off = np.cumsum(rec_len) - rec_len
            out = np.empty(int(rec_len.sum()), dtype=np.uint8)
            hdr = np.empty((m, H), dtype=np.uint8)
            hdr[:, 0:3] = np.frombuffer(b">WP", dtype=np.uint8)
            idx = np.arange(lo, hi, dtype=np.int64)
            for d in range(9):
                hdr[:, 3 + d] = 48 + (idx // 10 ** (8 - d)) % 10
            hdr[:, 12:15] = np.frombuffer(b".1_", dtype=np.uint8)
            t = db.taxids[lo:hi].astype(np.int64)
            for d in range(6):
                hdr[:, 15 + d] = 48 + (t // 10 ** (5 - d)) % 10
            hdr[:, 21] = 10
            out[(rec_off[:, None] + np.arange(H)[None, :]).reshape(-1)] = hdr.reshape(-1)
            tot = int(ln.sum())
            within = np.arange(tot, dtype=np.int64) - np.repeat(np.cumsum(ln) - ln, ln)
            out[np.repeat(rec_off + H, ln) + within] = aa[db.codes[db.offsets[lo]:db.offsets[hi]]]
            out[rec_off + H + ln] = 10
            f.write(out.tobytes())


def write_fasta(db: SynthDB, path, width=0):
    aa = np.frombuffer(AA.encode(), dtype=np.uint8)
    text = aa[db.codes]
    with open(path, "wb") as f:
        for i in range(db.nseq):
            f.write(b">" + db.names[i].encode() + b"\n")
            f.write(text[db.offsets[i]:db.offsets[i + 1]].tobytes())
            f.write(b"\n")


def make_reads(db: SynthDB, n, seed=777, read_len=150, frac_db=0.70, chunk=1 << 20):
    """Return (uint8 ASCII array [n, read_len]) of synthetic reads."""
    rng = np.random.default_rng(seed)
    naa = read_len // 3 + 1           # 51-aa window for 150 nt
    lens = np.diff(db.offsets)
    elig = np.nonzero(lens >= naa)[0]
    nt = np.frombuffer(_NT.encode(), dtype=np.uint8)
    comp = np.array([3, 2, 1, 0], dtype=np.uint8)
    out = np.empty((n, read_len), dtype=np.uint8)
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        seq = elig[rng.integers(0, len(elig), size=m)]
        start = db.offsets[seq] + (rng.random(m) * (lens[seq] - naa + 1)).astype(np.int64)
        win = db.codes[start[:, None] + np.arange(naa)[None, :]]            # [m, naa]
        pick = (rng.random((m, naa)) * _NSYN[win]).astype(np.int64)
        cod = _CODTAB[win, pick]                                            # codon index 0..63
        nts = np.stack([(cod >> 4) & 3, (cod >> 2) & 3, cod & 3], axis=2).reshape(m, naa * 3)
        off = rng.integers(0, 3, size=m)
        reads = nts[np.arange(m)[:, None], off[:, None] + np.arange(read_len)[None, :]].astype(np.uint8)
        nsub = rng.choice(np.array([0, 0, 1, 3, 8]), size=m)
        for k in range(8):
            sel = np.nonzero(nsub > k)[0]
            pos = rng.integers(0, read_len, size=len(sel))
            reads[sel, pos] = rng.integers(0, 4, size=len(sel), dtype=np.uint8)
        rc = rng.random(m) < 0.5
        reads[rc] = comp[reads[rc][:, ::-1]]
        rnd = rng.random(m) >= frac_db
        reads[rnd] = rng.integers(0, 4, size=(int(rnd.sum()), read_len), dtype=np.uint8)
        out[lo:lo + m] = nt[reads]
    return out


def make_protein_reads(db: SynthDB, n, seed=779, read_len=100, frac_db=0.70, chunk=1 << 20):
    """Protein reads (kaiju -p / kaijup workloads): uint8 ASCII array [n, read_len]; frac_db of them windows of database
    proteins with {0, 0, 1, 2, 5} substituted residues, the rest i.i.d. background residues."""
    rng = np.random.default_rng(seed)
    lens = np.diff(db.offsets)
    elig = np.nonzero(lens >= read_len)[0]
    letters = np.frombuffer(AA.encode(), dtype=np.uint8)
    out = np.empty((n, read_len), dtype=np.uint8)
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        seq = elig[rng.integers(0, len(elig), size=m)]
        start = db.offsets[seq] + (rng.random(m) * (lens[seq] - read_len + 1)).astype(np.int64)
        win = db.codes[start[:, None] + np.arange(read_len)[None, :]].copy()
        nsub = rng.choice(np.array([0, 0, 1, 2, 5]), size=m)
        for k in range(5):
            sel = np.nonzero(nsub > k)[0]
            win[sel, rng.integers(0, read_len, size=len(sel))] = rng.integers(0, 20, size=len(sel), dtype=np.uint8)
        rnd = rng.random(m) >= frac_db
        win[rnd] = rng.choice(20, size=(int(rnd.sum()), read_len), p=_BG).astype(np.uint8)
        out[lo:lo + m] = letters[win]
    return out


def make_pairs(db: SynthDB, n, seed=778, read_len=150, insert=300):
    """Paired-end reads: mate 1 forward from the insert start, mate 2 the reverse
    complement of the insert end.  Returns two uint8 arrays [n, read_len]."""
    rng = np.random.default_rng(seed)
    naa = insert // 3 + 1
    lens = np.diff(db.offsets)
    elig = np.nonzero(lens >= naa)[0]
    nt = np.frombuffer(_NT.encode(), dtype=np.uint8)
    comp = np.array([3, 2, 1, 0], dtype=np.uint8)
    seq = elig[rng.integers(0, len(elig), size=n)]
    start = db.offsets[seq] + (rng.random(n) * (lens[seq] - naa + 1)).astype(np.int64)
    win = db.codes[start[:, None] + np.arange(naa)[None, :]]
    pick = (rng.random((n, naa)) * _NSYN[win]).astype(np.int64)
    cod = _CODTAB[win, pick]
    nts = np.stack([(cod >> 4) & 3, (cod >> 2) & 3, cod & 3], axis=2).reshape(n, naa * 3).astype(np.uint8)
    off = rng.integers(0, 3, size=n)
    ins = nts[np.arange(n)[:, None], off[:, None] + np.arange(insert)[None, :]]
    rnd = rng.random(n) >= 0.7
    ins[rnd] = rng.integers(0, 4, size=(int(rnd.sum()), insert), dtype=np.uint8)
    m1 = ins[:, :read_len].copy()
    m2 = comp[ins[:, ::-1][:, :read_len]]
    for mate in (m1, m2):
        nsub = rng.choice(np.array([0, 0, 1, 3]), size=n)
        for k in range(3):
            sel = np.nonzero(nsub > k)[0]
            pos = rng.integers(0, read_len, size=len(sel))
            mate[sel, pos] = rng.integers(0, 4, size=len(sel), dtype=np.uint8)
    swap = rng.random(n) < 0.5
    a = np.where(swap[:, None], m2, m1)
    b = np.where(swap[:, None], m1, m2)
    return nt[a], nt[b]


def write_fastq(reads: np.ndarray, path, prefix="r", suffix=""):
    n, L = reads.shape
    qual = b"I" * L
    with open(path, "wb") as f:
        for i in range(n):
            f.write(b"@" + f"{prefix}{i}{suffix}".encode() + b"\n")
            f.write(reads[i].tobytes())
            f.write(b"\n+\n" + qual + b"\n")


def pack_reads(reads: np.ndarray, reads2: np.ndarray | None = None):
    """Batch layout of the C-ABI (include/kaiju_gpu.h): concatenated ASCII plus
    off[2n+1]; mate 2 is empty when unpaired."""
    n, L = reads.shape
    if reads2 is None:
        seqs = np.ascontiguousarray(reads).reshape(-1)
        off = np.empty(2 * n + 1, dtype=np.uint64)
        off[0::2] = np.arange(n + 1, dtype=np.uint64) * L
        off[1::2] = off[2::2]
        return seqs, off
    L2 = reads2.shape[1]
    seqs = np.concatenate([reads, reads2], axis=1).reshape(-1)
    off = np.empty(2 * n + 1, dtype=np.uint64)
    base = np.arange(n + 1, dtype=np.uint64) * (L + L2)
    off[0::2] = base
    off[1::2] = base[:-1] + L
    return np.ascontiguousarray(seqs), off


def default_workdir():
    d = os.environ.get("KAIJU_AMD_WORKDIR", "/tmp/kaiju_amd_work")
    os.makedirs(d, exist_ok=True)
    return d
