"""ctypes bindings for the CPU oracle — TEST INFRASTRUCTURE ONLY.

``Oracle`` wraps oracle/libkaiju_oracle.so (our plain-C restatement of the
reference path).  ``RefLib`` wraps oracle/_ref/libkaijuref.so (the UNMODIFIED
reference FM-index + SEG objects) for function-level known-answer checks and
``ref_kaiju`` runs the unmodified reference binary oracle/_ref/kaiju.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; nothing in kaiju_amd/ does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libkaiju_oracle.so")
REF_DIR = os.path.join(HERE, "_ref")
REF_SO = os.path.join(REF_DIR, "libkaijuref.so")
REF_KAIJU = os.path.join(REF_DIR, "kaiju")
REF_KAIJUX = os.path.join(REF_DIR, "kaijux")
REF_MKBWT = os.path.join(REF_DIR, "kaiju-mkbwt")
REF_MKFMI = os.path.join(REF_DIR, "kaiju-mkfmi")

KO_MAX_IDS = 21


class KoParams(C.Structure):
    _fields_ = [("mode", C.c_int32), ("min_fragment_length", C.c_uint32),
                ("mismatches", C.c_uint32), ("min_score", C.c_uint32),
                ("seed_length", C.c_uint32), ("seg", C.c_int32),
                ("use_evalue", C.c_int32), ("min_evalue", C.c_double),
                ("max_matches_SI", C.c_uint32), ("max_match_ids", C.c_uint32),
                ("kaijux", C.c_int32), ("protein", C.c_int32)]


class KoHit(C.Structure):
    _fields_ = [("best", C.c_uint32), ("n_ids", C.c_uint32), ("flags", C.c_uint32),
                ("classified", C.c_uint32), ("lca", C.c_uint64),
                ("taxid", C.c_uint64 * KO_MAX_IDS)]


HIT_DTYPE = np.dtype([("best", "<u4"), ("n_ids", "<u4"), ("flags", "<u4"),
                      ("classified", "<u4"), ("lca", "<u8"), ("taxid", "<u8", (KO_MAX_IDS,))])


class KoCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("initial_si", "update_si", "fmindex", "fmindex_current", "get_suffix",
                 "sa_decode", "bwt_scanned", "fragments_searched", "seg_calls")]


def build_oracle():
    subprocess.run(["make", "-s", "-C", HERE, "oracle"], check=True)


def have_ref():
    return all(os.path.exists(p) for p in (REF_SO, REF_KAIJU, REF_MKBWT, REF_MKFMI))


class Oracle:
    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        L = self.lib = C.CDLL(ORACLE_SO)
        L.ko_load_fmi.restype = C.c_void_p
        L.ko_load_fmi.argtypes = [C.c_char_p]
        L.ko_free_index.argtypes = [C.c_void_p]
        L.ko_bwtlen.restype = C.c_int64
        L.ko_bwtlen.argtypes = [C.c_void_p]
        L.ko_nseq.restype = C.c_int32
        L.ko_nseq.argtypes = [C.c_void_p]
        L.ko_alen.restype = C.c_int32
        L.ko_alen.argtypes = [C.c_void_p]
        L.ko_seq_name.restype = C.c_char_p
        L.ko_seq_name.argtypes = [C.c_void_p, C.c_int32]
        L.ko_seq_taxid.restype = C.c_uint64
        L.ko_seq_taxid.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int)]
        L.ko_fmindex.restype = C.c_int64
        L.ko_fmindex.argtypes = [C.c_void_p, C.c_int, C.c_int64]
        L.ko_fmindex_current.restype = C.c_int64
        L.ko_fmindex_current.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_int)]
        L.ko_initial_si.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64)]
        L.ko_update_si.restype = C.c_int64
        L.ko_update_si.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.ko_get_suffix.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
        L.ko_fragments.restype = C.c_int
        L.ko_fragments.argtypes = [C.POINTER(KoParams), C.c_char_p, C.c_int, C.c_char_p, C.c_int,
                                   C.POINTER(C.c_uint32), C.c_int]
        L.ko_seg.restype = C.c_int
        L.ko_seg.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int]
        L.ko_load_nodes.restype = C.c_void_p
        L.ko_load_nodes.argtypes = [C.c_char_p]
        L.ko_free_taxonomy.argtypes = [C.c_void_p]
        L.ko_lca.restype = C.c_uint64
        L.ko_lca.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
        L.ko_classify_batch.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(KoParams), C.c_void_p,
                                        C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
        L.ko_default_params.argtypes = [C.POINTER(KoParams), C.c_int]
        L.ko_counters_get.argtypes = [C.POINTER(KoCounters)]

    def params(self, mode="mem", **kw):
        p = KoParams()
        self.lib.ko_default_params(C.byref(p), 0 if mode == "mem" else 1)
        for k, v in kw.items():
            setattr(p, k, v)
        return p

    def load_fmi(self, path):
        h = self.lib.ko_load_fmi(path.encode())
        if not h:
            raise IOError(f"oracle: cannot load {path}")
        return h

    def load_nodes(self, path):
        h = self.lib.ko_load_nodes(path.encode())
        if not h:
            raise IOError(f"oracle: cannot load {path}")
        return h

    def fragments(self, params, read: bytes, max_frags=256):
        buf = C.create_string_buffer(len(read) * 4 + 64)
        keys = (C.c_uint32 * max_frags)()
        n = self.lib.ko_fragments(C.byref(params), read, len(read), buf, len(buf), keys, max_frags)
        raw = buf.raw
        out, off = [], 0
        for i in range(n):
            e = raw.index(b"\0", off)
            out.append((keys[i], raw[off:e]))
            off = e + 1
        return out

    def seg(self, aa: bytes, max_regions=64):
        l = (C.c_int32 * max_regions)()
        r = (C.c_int32 * max_regions)()
        n = self.lib.ko_seg(aa, len(aa), l, r, max_regions)
        return [(l[i], r[i]) for i in range(min(n, max_regions))]

    def classify(self, ix, tax, params, seqs: np.ndarray, off: np.ndarray, paired=False):
        n = (len(off) - 1) // 2
        hits = np.zeros(n, dtype=HIT_DTYPE)
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        self.lib.ko_classify_batch(ix, tax, C.byref(params), seqs.ctypes.data, off.ctypes.data,
                                   n, 1 if paired else 0, hits.ctypes.data)
        return hits

    def counters(self, reset=False):
        c = KoCounters()
        self.lib.ko_counters_get(C.byref(c))
        if reset:
            self.lib.ko_counters_reset()
        return {n: getattr(c, n) for n, _ in KoCounters._fields_}


class RefLib:
    """Function-level access to the unmodified reference objects."""

    def __init__(self):
        self.lib = L = C.CDLL(REF_SO)
        self.libc = C.CDLL(None)
        self.libc.fopen.restype = C.c_void_p
        self.libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
        self.libc.fclose.argtypes = [C.c_void_p]
        L.readIndexes.restype = C.c_void_p
        L.readIndexes.argtypes = [C.c_void_p]
        L.FMindex.restype = C.c_long
        L.FMindex.argtypes = [C.c_void_p, C.c_ubyte, C.c_long]
        L.FMindexCurrent.restype = C.c_long
        L.FMindexCurrent.argtypes = [C.c_void_p, C.POINTER(C.c_ubyte), C.c_long]
        L.get_suffix.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.POINTER(C.c_int), C.POINTER(C.c_long)]
        L.SegParametersNewAa.restype = C.c_void_p
        L.SeqBufferSeg.restype = C.c_short
        L.SeqBufferSeg.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]
        L.BlastSeqLocFree.restype = C.c_void_p
        L.BlastSeqLocFree.argtypes = [C.c_void_p]
        self.segp = L.SegParametersNewAa()
        # SegParameters: {int window; double locut; double hicut; int period; int hilenmin;
        #                 Boolean overlaps; int maxtrim; int maxbogus;}  -> overlaps = TRUE (Config.cpp:26)
        class SegP(C.Structure):
            _fields_ = [("window", C.c_int), ("locut", C.c_double), ("hicut", C.c_double),
                        ("period", C.c_int), ("hilenmin", C.c_int), ("overlaps", C.c_ubyte),
                        ("maxtrim", C.c_int), ("maxbogus", C.c_int)]
        sp = C.cast(self.segp, C.POINTER(SegP)).contents
        assert sp.window == 12 and sp.maxtrim == 50 and sp.maxbogus == 2
        sp.overlaps = 1
        self.ncbistdaa = (C.c_ubyte * 128).in_dll(L, "AMINOACID_TO_NCBISTDAA")

    def read_indexes(self, path):
        fp = self.libc.fopen(path.encode(), b"r")
        assert fp
        b = self.lib.readIndexes(fp)
        self.libc.fclose(fp)

        # BWT { long len; int nseq; uchar* bwt; int alen; char* alphabet; FMI* f; suffixArray* s; }
        class BWT(C.Structure):
            _fields_ = [("len", C.c_long), ("nseq", C.c_int), ("bwt", C.c_void_p), ("alen", C.c_int),
                        ("alphabet", C.c_char_p), ("f", C.c_void_p), ("s", C.c_void_p)]
        bw = C.cast(b, C.POINTER(BWT)).contents
        return bw

    def fmindex(self, bw, c, k):
        return self.lib.FMindex(bw.f, c, k)

    def fmindex_current(self, bw, k):
        c = C.c_ubyte()
        v = self.lib.FMindexCurrent(bw.f, C.byref(c), k)
        return v, c.value

    def get_suffix(self, bw, i):
        iseq, pos = C.c_int(), C.c_long()
        self.lib.get_suffix(bw.f, bw.s, i, C.byref(iseq), C.byref(pos))
        return iseq.value, pos.value

    def seg(self, aa: bytes):
        conv = bytes(self.ncbistdaa[c] for c in aa)
        locs = C.c_void_p()
        self.lib.SeqBufferSeg(conv, len(conv), 0, self.segp, C.byref(locs))

        class SSR(C.Structure):
            _fields_ = [("left", C.c_int32), ("right", C.c_int32)]

        class Loc(C.Structure):
            pass
        Loc._fields_ = [("next", C.POINTER(Loc)), ("ssr", C.POINTER(SSR))]
        out = []
        p = C.cast(locs, C.POINTER(Loc))
        while p:
            out.append((p.contents.ssr.contents.left, p.contents.ssr.contents.right))
            p = p.contents.next
        if locs:
            self.lib.BlastSeqLocFree(locs)
        return out


def ref_build_index(faa_path, out_prefix, threads=8, exponent=3):
    """kaiju-mkbwt + kaiju-mkfmi with the parameters of util/kaiju-makedb:16,373-375."""
    subprocess.run([REF_MKBWT, "-n", str(threads), "-e", str(exponent), "-a", "ACDEFGHIKLMNPQRSTVWY",
                    "-o", out_prefix, faa_path], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.run([REF_MKFMI, out_prefix], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for ext in (".bwt", ".sa"):
        try:
            os.remove(out_prefix + ext)
        except OSError:
            pass
    return out_prefix + ".fmi"


def ref_kaiju(nodes, fmi, reads, out, mode="mem", reads2=None, seg=True, threads=1, extra=()):
    cmd = [REF_KAIJU, "-t", nodes, "-f", fmi, "-i", reads, "-o", out, "-z", str(threads), "-v"]
    cmd += ["-a", mode]
    if reads2:
        cmd += ["-j", reads2]
    if not seg:
        cmd += ["-X"]
    cmd += list(extra)
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return out


def ref_kaijux(fmi, reads, out, mode="greedy", reads2=None, seg=True, threads=1, extra=()):
    """the reference's kaijux (no taxonomy; lines name database sequences)"""
    cmd = [REF_KAIJUX, "-f", fmi, "-i", reads, "-o", out, "-z", str(threads), "-a", mode]
    if reads2:
        cmd += ["-j", reads2]
    if not seg:
        cmd += ["-X"]
    cmd += list(extra)
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return out


def parse_kaiju_tsv(path):
    """-> dict name -> (C/U, taxid, best or None, sorted id tuple)"""
    res = {}
    with open(path) as f:
        for line in f:
            p = line.rstrip("\n").split("\t")
            if p[0] == "C":
                ids = tuple(int(x) for x in p[4].split(",") if x) if len(p) > 4 else ()
                res[p[1]] = ("C", int(p[2]), int(p[3]) if len(p) > 3 else None, ids)
            else:
                res[p[1]] = ("U", 0, None, ())
    return res
