"""Shared helpers of the test-suite (test infrastructure)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU_SO = os.path.join(EMU_DIR, "libkaiju_kernel_emu.so")
CSRC = os.path.join(ROOT, "kaiju_amd", "csrc")

GPU_HIT = np.dtype([("best", "<u4"), ("n_ids", "<u4"), ("flags", "<u4"), ("reserved", "<u4"),
                    ("taxid", "<u8", (21,))])


class GP(C.Structure):      # kaiju_gpu_params
    _fields_ = [("mode", C.c_int32), ("min_fragment_length", C.c_uint32), ("mismatches", C.c_uint32),
                ("min_score", C.c_uint32), ("seed_length", C.c_uint32), ("seg", C.c_int32),
                ("use_evalue", C.c_int32), ("input_is_protein", C.c_int32), ("min_evalue", C.c_double),
                ("max_matches_SI", C.c_uint32), ("max_match_ids", C.c_uint32)]


def gp(mode, m=11, mismatches=3, min_score=65, seed_length=7, seg=1, use_evalue=None, min_evalue=0.01, protein=0):
    md = 0 if mode in ("mem", 0) else 1
    if use_evalue is None:
        use_evalue = md
    return GP(md, m, mismatches, min_score, seed_length, seg, use_evalue, protein, min_evalue, 20, 20)


def build_emu(so=None, defines=()):
    so = so or EMU_SO
    srcs = [os.path.join(EMU_DIR, "kernel_emu.cpp")] + [os.path.join(CSRC, f) for f in
                                                        ("host_index.cpp", "host_tables.cpp", "taxonomy.cpp")]
    deps = srcs + [os.path.join(CSRC, f) for f in ("kj_core.h", "kj_greedy3.h", "host_index.h", "host_tables.h", "fmi_stream.h")]
    if os.path.exists(so) and all(os.path.getmtime(d) <= os.path.getmtime(so) for d in deps):
        return
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-w", "-pthread"]
                   + ["-D" + d for d in defines] + ["-o", so] + srcs, check=True)


class Emu:
    def __init__(self, so=None, defines=()):
        so = so or EMU_SO
        build_emu(so, defines)
        E = self.lib = C.CDLL(so)
        E.emu_index_load.restype = C.c_void_p
        E.emu_index_load.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        E.emu_index_free.argtypes = [C.c_void_p]
        E.emu_index_warnings.restype = C.c_uint32
        E.emu_index_warnings.argtypes = [C.c_void_p]
        E.emu_rank.restype = C.c_uint64
        E.emu_rank.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64]
        E.emu_symbol.restype = C.c_uint32
        E.emu_symbol.argtypes = [C.c_void_p, C.c_uint64]
        E.emu_seg.restype = C.c_int
        E.emu_seg.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        E.emu_classify.restype = C.c_int
        E.emu_classify.argtypes = [C.c_void_p, C.POINTER(GP), C.c_void_p, C.c_void_p, C.c_uint32, C.c_int,
                                   C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32),
                                   C.c_char_p, C.c_uint64]

    def load(self, fmi):
        err = C.create_string_buffer(512)
        h = self.lib.emu_index_load(fmi.encode(), err, 512)
        if not h:
            raise RuntimeError(err.value.decode())
        return h

    def seg(self, h, aa: bytes):
        l = (C.c_int32 * 64)()
        r = (C.c_int32 * 64)()
        n = self.lib.emu_seg(h, aa, len(aa), l, r)
        assert n >= 0
        return [(l[i], r[i]) for i in range(n)]

    def classify(self, h, params, seqs, off, paired=False, caps=(16, 192, 64), want_frags=False, allow_capacity=False):
        """allow_capacity: return (None, 0) instead of failing when a capacity bound of the kernels was hit
        (kaiju_gpu_stats.error_flags on the device: a fragment with more than 15 SEG regions, ...)"""
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n = (len(off) - 1) // 2
        out = np.zeros(n, dtype=GPU_HIT)
        nretry = C.c_uint32(0)
        dump = C.create_string_buffer(max(1 << 20, 8 * len(seqs) + 64 * n)) if want_frags else None
        rc = self.lib.emu_classify(h, C.byref(params), seqs.ctypes.data, off.ctypes.data, n, 1 if paired else 0,
                                   out.ctypes.data, caps[0], caps[1], caps[2], C.byref(nretry), dump,
                                   len(dump) if dump else 0)
        if rc == -100 and allow_capacity:
            return None, 0
        assert rc == 0, rc
        if want_frags:
            return out, nretry.value, dump.value.decode()
        return out, nretry.value


def read_fastq(path):
    names, seqs = [], []
    with open(path, "rb") as f:
        while True:
            h = f.readline()
            if not h:
                break
            s = f.readline().rstrip(b"\n")
            f.readline()
            f.readline()
            names.append(h[1:].strip().decode())
            seqs.append(s)
    return names, seqs


def read_fasta(path, keep_names=False):
    """FASTA records as the reference drivers read them: multi-line records joined, sequences strip()'d of everything
    that is no letter (util.cpp:25-32), names cut at the first of " /\\t\\r" (kaiju.cpp:303-331) unless keep_names
    (kaijup.cpp:249-262)"""
    names, seqs = [], []
    with open(path, "rb") as f:
        for line in f:
            line = line.rstrip(b"\n")
            if line.startswith(b">"):
                nm = line[1:]
                if not keep_names:
                    for i, ch in enumerate(nm):
                        if ch in b" /\t\r":
                            nm = nm[:i]
                            break
                names.append(nm.decode())
                seqs.append(b"")
            elif names:
                seqs[-1] += bytes(c for c in line if chr(c).isalpha())
    return names, seqs


def pack(seqs1, seqs2=None):
    """list(s) of bytes -> (uint8 array, off[2n+1]) in the batch layout of include/kaiju_gpu.h"""
    n = len(seqs1)
    parts = []
    off = np.zeros(2 * n + 1, dtype=np.uint64)
    pos = 0
    for i in range(n):
        a = seqs1[i]
        b = seqs2[i] if seqs2 is not None else b""
        parts.append(a)
        parts.append(b)
        off[2 * i] = pos
        pos += len(a)
        off[2 * i + 1] = pos
        pos += len(b)
    off[2 * n] = pos
    buf = np.frombuffer(b"".join(parts), dtype=np.uint8).copy() if pos else np.zeros(0, dtype=np.uint8)
    return buf, off


def parse_tsv(path):
    """reference `kaiju -v` output -> dict name -> (C/U, taxon, best, sorted ids)"""
    res = {}
    with open(path) as f:
        for line in f:
            p = line.rstrip("\n").split("\t")
            if p[0] == "C":
                ids = tuple(int(x) for x in p[4].split(",") if x)
                res[p[1]] = ("C", int(p[2]), int(p[3]), ids)
            else:
                res[p[1]] = ("U", 0, None, ())
    return res


class Golden:
    def __init__(self):
        self.dir = GOLD
        self.fmi = os.path.join(GOLD, "db.fmi")
        self.nodes = os.path.join(GOLD, "nodes.dmp")
        self.names, self.reads = read_fastq(os.path.join(GOLD, "reads.fq"))
        self.pnames, self.p1 = read_fastq(os.path.join(GOLD, "pairs_1.fq"))
        _, self.p2 = read_fastq(os.path.join(GOLD, "pairs_2.fq"))
        self.seqs, self.off = pack(self.reads)
        self.pseqs, self.poff = pack(self.p1, self.p2)
        # protein reads (kaiju -p, kaijup): tests/golden/make_golden_protein.py
        self.prot_fa = os.path.join(GOLD, "prot.fa")
        self.prot_names, self.prot_reads = read_fasta(self.prot_fa)
        self.prot_fullnames, _ = read_fasta(self.prot_fa, keep_names=True)
        self.prot_seqs, self.prot_off = pack(self.prot_reads)

    def tsv(self, name):
        return parse_tsv(os.path.join(GOLD, name))

    def short(self, max_len=191):
        """(indices, seqs, off) of the golden single reads of at most max_len nucleotides: batches the fast stage 1 serves
        (kj_core.h: kS1MaxLen), lazy SEG and all - the whole set holds 301- and 1000-nt reads and takes the general path"""
        idx = [i for i, r in enumerate(self.reads) if len(r) <= max_len]
        seqs, off = pack([self.reads[i] for i in idx])
        return idx, seqs, off


def oracle_records(hits):
    """oracle hits -> list of (C/U, taxon, best, sorted ids) like parse_tsv values"""
    out = []
    for h in hits:
        if h["classified"]:
            out.append(("C", int(h["lca"]), int(h["best"]), tuple(sorted(int(x) for x in h["taxid"][:h["n_ids"]]))))
        else:
            out.append(("U", 0, None, ()))
    return out


def same_hit(o, g, mask=3):
    """oracle hit vs device hit: best, ids in traversal order, public flag bits"""
    return (int(o["best"]) == int(g["best"]) and int(o["n_ids"]) == int(g["n_ids"]) and
            list(o["taxid"][:o["n_ids"]]) == list(g["taxid"][:g["n_ids"]]) and
            (int(o["flags"]) & mask) == (int(g["flags"]) & mask))


BACK = {"A": "GCT", "R": "CGT", "N": "AAT", "D": "GAT", "C": "TGT", "Q": "CAA", "E": "GAA", "G": "GGT", "H": "CAT", "I": "ATT",
        "L": "CTT", "K": "AAA", "M": "ATG", "F": "TTT", "P": "CCT", "S": "TCT", "T": "ACT", "W": "TGG", "Y": "TAT", "V": "GTT"}


def repetitive_db(faa, nseq=201, seed=8):
    """a database in which the order of a fragment's matches decides which sequences get through the 21-id cap: motif A
    sits in the first half of the sequences, motif B in the second half (flanks random, never W).  Returns reads whose
    fragments hold both motifs separated by W: two matches of equal length with disjoint, large intervals."""
    rng = np.random.default_rng(seed)
    AA = "ARNDCQEGHILKMFPSTYV"                      # no W
    A = "".join(rng.choice(list(AA), 20))
    B = "".join(rng.choice(list(AA), 20))
    with open(faa, "w") as f:
        for i in range(nseq):
            pre = "".join(rng.choice(list(AA), int(rng.integers(5, 40))))
            suf = "".join(rng.choice(list(AA), int(rng.integers(5, 40))))
            f.write(f">S{i}_7\n{pre}{A if i < nseq // 2 else B}{suf}\n")
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    reads = []
    for k, pep in enumerate(("AWB", "BWA", "AWBWA", "BWAWB", "AWA", "BWB", "A", "B") * 6):
        s = "".join(BACK[c] for c in "W".join({"A": A, "B": B}[x] for x in pep.split("W")))
        s = "ACGT"[: k % 3] + s
        if k % 2:
            s = "".join(comp[c] for c in reversed(s))
        reads.append(s.encode())
    return reads


def long_reads(n=120, seed=5, lo=400, hi=3000):
    """long reads (stage 1 writes in place, windows refill, k-mer starts cross window borders): back-translated
    stretches of the golden proteins with substitutions, low-complexity inserts, N runs, plus random reads;
    odd lengths on purpose"""
    rng = np.random.default_rng(seed)
    prots = []
    with open(os.path.join(GOLD, "db.faa")) as f:
        cur = []
        for line in f:
            if line.startswith(">"):
                if cur:
                    prots.append("".join(cur))
                cur = []
            else:
                cur.append(line.strip())
        if cur:
            prots.append("".join(cur))
    codon = {"A": "GCT", "R": "CGT", "N": "AAT", "D": "GAT", "C": "TGT", "Q": "CAA", "E": "GAA", "G": "GGT", "H": "CAT",
             "I": "ATT", "L": "CTT", "K": "AAA", "M": "ATG", "F": "TTT", "P": "CCT", "S": "TCT", "T": "ACT", "W": "TGG",
             "Y": "TAT", "V": "GTT"}
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
    out = []
    for i in range(n):
        L = int(rng.integers(lo, hi))
        if i % 5 == 4:
            s = "".join(rng.choice(list("ACGT"), L))
        else:
            parts = []
            while sum(len(p) for p in parts) < L:
                p = prots[int(rng.integers(0, len(prots)))]
                a = int(rng.integers(0, max(1, len(p) - 30)))
                b = min(len(p), a + int(rng.integers(20, 400)))
                nt = "".join(codon[c] for c in p[a:b] if c in codon)
                if rng.random() < 0.3:
                    nt += "GCT" * int(rng.integers(5, 30))          # poly-A peptide: SEG material
                if rng.random() < 0.2:
                    nt += "N" * int(rng.integers(1, 5))
                parts.append(nt + "A" * int(rng.integers(0, 3)))    # frame shifts
            s = list("".join(parts)[:L])
            for _ in range(int(rng.integers(0, 6))):
                s[int(rng.integers(0, len(s)))] = "ACGT"[int(rng.integers(0, 4))]
            s = "".join(s)
            if rng.random() < 0.5:
                s = "".join(comp[c] for c in reversed(s))
        out.append(s.encode())
    return out


def many_region_reads(n=12, seed=6, islands=24):
    """proteins (and their back-translations) whose single fragment holds far more low-complexity regions than a record of
    the SEG pass (15): stretches of the golden proteins alternating with short low-complexity islands"""
    rng = np.random.default_rng(seed)
    prots = []
    with open(os.path.join(GOLD, "db.faa")) as f:
        for line in f:
            if not line.startswith(">"):
                prots.append(line.strip())
    AA = "ARNDCQEGHILKMFPSTWYV"
    out_p, out_n = [], []
    for _ in range(n):
        parts = []
        for _ in range(islands):
            p = prots[int(rng.integers(0, len(prots)))]
            a = int(rng.integers(0, max(1, len(p) - 40)))
            parts.append(p[a: a + int(rng.integers(25, 40))])
            x, y = AA[int(rng.integers(0, 20))], AA[int(rng.integers(0, 20))]
            parts.append("".join(rng.choice([x, x, x, y], int(rng.integers(12, 20)))))
        s = "".join(parts)
        out_p.append(s.encode())
        out_n.append("".join(BACK[c] for c in s).encode())
    return out_p, out_n
