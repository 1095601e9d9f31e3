"""Python mirror of the host side of the seam, over the C-ABI (include/kaiju_gpu.h).

The names follow the reference: ``Config``-like parameters (Config.hpp:33-48), an index
loaded from a ``.fmi`` file (readFMI, util.cpp:265-276), a classifier that turns batches of
reads into per-read hit records (ConsumerThread::doWork, ConsumerThread.cpp:630-749) and the
taxonomy / LCA helpers (util.cpp:79-99, 194-263).  Everything numerical happens in
``libkaiju_gpu.so``; if that library or a HIP device is missing the calls raise — there is
no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build as _build

MAX_IDS = 21
MEM, GREEDY = 0, 1


class Params(C.Structure):
    _fields_ = [("mode", C.c_int32), ("min_fragment_length", C.c_uint32),
                ("mismatches", C.c_uint32), ("min_score", C.c_uint32),
                ("seed_length", C.c_uint32), ("seg", C.c_int32),
                ("use_evalue", C.c_int32), ("input_is_protein", C.c_int32), ("min_evalue", C.c_double),
                ("max_matches_SI", C.c_uint32), ("max_match_ids", C.c_uint32)]


class IndexInfo(C.Structure):
    _fields_ = [("bwtlen", C.c_int64), ("nseq", C.c_int32), ("alen", C.c_int32),
                ("chpt_exp", C.c_int32), ("db_length", C.c_double),
                ("device_bytes", C.c_uint64), ("warnings", C.c_uint32),
                ("alphabet", C.c_char * 64)]


class IndexFootprint(C.Structure):        # kaiju_gpu_index_footprint
    _fields_ = [(k, C.c_uint64) for k in ("rank_blocks", "count_bases", "sa_seq", "sa_taxid", "seq_tables", "kmer_table",
                                          "kmer_lines", "text", "sa_full", "other", "total")] + [("kmer_k", C.c_uint32), ("wide", C.c_uint32)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class Stats(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("n_seg_fragments", C.c_uint64), ("n_overflow_retries", C.c_uint64),
                ("error_flags", C.c_uint64),
                ("ms_translate", C.c_double), ("ms_seg", C.c_double), ("ms_search", C.c_double),
                ("ms_retry", C.c_double), ("ms_total", C.c_double)]


HIT_DTYPE = np.dtype([("best", "<u4"), ("n_ids", "<u4"), ("flags", "<u4"), ("reserved", "<u4"),
                      ("taxid", "<u8", (MAX_IDS,))])
RESULT_DTYPE = np.dtype([("taxon", "<u8"), ("best", "<u4"), ("classified", "u1"), ("pad", "u1", (3,))])
VERBOSE_DTYPE = np.dtype([("n_acc", "<u4"), ("text_len", "<u4"), ("truncated", "<u4"), ("acc_iseq", "<u4", (20,))])  # kaiju_gpu_verbose
COMPACT_DTYPE = np.dtype([("lca", "<u8"), ("best", "<u4"), ("info", "<u4")])     # kaiju_gpu_compact
assert HIT_DTYPE.itemsize == 184 and RESULT_DTYPE.itemsize == 16 and COMPACT_DTYPE.itemsize == 16


class KaijuGpuError(RuntimeError):
    pass


_lib = None


def lib():
    """Load (building if necessary) libkaiju_gpu.so.  Raises if it cannot be loaded."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("KAIJU_GPU_LIB") or _build.LIB      # (KAIJU_GPU_LIB: another build of the library, A/B measurements)
    if not os.path.exists(path):
        _build.build()
    L = C.CDLL(path)
    L.kaiju_gpu_strerror.restype = C.c_char_p
    L.kaiju_gpu_last_error.restype = C.c_char_p
    L.kaiju_gpu_index_load.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
    L.kaiju_gpu_index_get_info.argtypes = [C.c_void_p, C.POINTER(IndexInfo)]
    L.kaiju_gpu_index_free.argtypes = [C.c_void_p]
    if hasattr(L, "kaiju_gpu_index_get_footprint"):        # (an older build loaded through KAIJU_GPU_LIB for an A/B run has none)
        L.kaiju_gpu_index_get_footprint.argtypes = [C.c_void_p, C.POINTER(IndexFootprint)]
    L.kaiju_gpu_default_params.argtypes = [C.POINTER(Params), C.c_int]
    L.kaiju_gpu_create.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.POINTER(Params)]
    L.kaiju_gpu_destroy.argtypes = [C.c_void_p]
    L.kaiju_gpu_classify_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
    L.kaiju_gpu_classify_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32,
                                                  C.c_int, C.c_void_p, C.c_void_p]
    L.kaiju_gpu_set_max_read_length.argtypes = [C.c_void_p, C.c_uint32]
    L.kaiju_gpu_synchronize.argtypes = [C.c_void_p]
    L.kaiju_gpu_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
    L.kaiju_taxonomy_load.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
    L.kaiju_taxonomy_free.argtypes = [C.c_void_p]
    L.kaiju_taxonomy_lca.restype = C.c_uint64
    L.kaiju_taxonomy_lca.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    L.kaiju_finalize_hits.argtypes = [C.c_void_p, C.POINTER(Params), C.c_double, C.c_void_p, C.c_void_p,
                                      C.c_uint32, C.c_int, C.c_void_p]
    L.kaiju_gpu_taxonomy_upload.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    L.kaiju_gpu_taxonomy_free.argtypes = [C.c_void_p]
    L.kaiju_gpu_lca_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.kaiju_gpu_classify_batch_device_compact.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32,
                                                          C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.kaiju_gpu_classify_batch_verbose.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p,
                                                   C.c_void_p, C.c_void_p, C.c_uint32]
    L.kaiju_gpu_index_seq_name.restype = C.c_char_p
    L.kaiju_gpu_index_seq_name.argtypes = [C.c_void_p, C.c_uint32]
    L.kaiju_gpu_lca_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.kaiju_finalize_compact.argtypes = [C.POINTER(Params), C.c_double, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int,
                                         C.c_void_p]
    L.kaiju_gpu_classify_batch_compact.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int,
                                                   C.c_void_p]
    L.kaiju_gpu_set_count_ops.argtypes = [C.c_void_p, C.c_int]
    L.kaiju_gpu_get_op_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        L = lib()
        raise KaijuGpuError(f"{L.kaiju_gpu_strerror(rc).decode()} ({rc}): {L.kaiju_gpu_last_error().decode()}")


def default_params(mode="greedy", **kw) -> Params:
    p = Params()
    lib().kaiju_gpu_default_params(C.byref(p), GREEDY if mode in ("greedy", GREEDY) else MEM)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def device_count() -> int:
    return lib().kaiju_gpu_device_count()


IDS_TAXON, IDS_SEQUENCE = 0, 1     # kaiju_gpu_index_load_ex


class Index:
    """FM-index resident in HBM (readFMI + Config::init of the reference)."""

    def __init__(self, fmi_path: str, device: int = 0, id_mode: int = 0):
        """id_mode IDS_SEQUENCE: hits collect database sequence numbers instead of taxon ids (kaijux / kaijup)"""
        self._h = C.c_void_p()
        L = lib()
        L.kaiju_gpu_index_load_ex.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        _check(L.kaiju_gpu_index_load_ex(fmi_path.encode(), device, id_mode, C.byref(self._h)))
        self.info = IndexInfo()
        _check(lib().kaiju_gpu_index_get_info(self._h, C.byref(self.info)))
        self.footprint = IndexFootprint()
        if hasattr(lib(), "kaiju_gpu_index_get_footprint"):
            _check(lib().kaiju_gpu_index_get_footprint(self._h, C.byref(self.footprint)))
        self.device = device

    @property
    def db_length(self):
        return self.info.db_length

    DIGEST_NAMES = ("rank_blocks", "count_bases", "sa_seq", "sa_taxid", "term_rows", "seq_taxid", "seq_valid", "kmer_table",
                    "kmer_lines", "text", "sa_full", "row_seq", "kmer_k", "C")

    def digest(self) -> dict:
        """kaiju_gpu_index_digest: a digest of every array this index holds in HBM (two loads of one index compare equal)"""
        L = lib()
        L.kaiju_gpu_index_digest.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        out = np.zeros(len(self.DIGEST_NAMES), dtype=np.uint64)
        _check(L.kaiju_gpu_index_digest(self._h, out.ctypes.data, len(out)))
        return {k: int(v) for k, v in zip(self.DIGEST_NAMES, out)}

    def close(self):
        if self._h:
            lib().kaiju_gpu_index_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """The processes of a node, one per GPU (kaiju_gpu_comm_create): rank 0 leaves the communicator's id in `rendezvous_path`.
    gather_compact() = ONE RCCL gather of 16-byte records to `root` (kaiju_gpu_gather_compact), asynchronous on `stream`."""

    def __init__(self, rendezvous_path: str, rank: int, world: int, device: int = 0):
        L = lib()
        L.kaiju_gpu_comm_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.kaiju_gpu_comm_destroy.argtypes = [C.c_void_p]
        L.kaiju_gpu_gather_compact.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.c_void_p]
        L.kaiju_gpu_comm_last_error.restype = C.c_char_p
        self._h = C.c_void_p()
        rc = L.kaiju_gpu_comm_create(rendezvous_path.encode(), rank, world, device, C.byref(self._h))
        if rc != 0:
            raise KaijuGpuError(f"{L.kaiju_gpu_strerror(rc).decode()} ({rc}): {L.kaiju_gpu_comm_last_error().decode()}")
        self.rank, self.world = rank, world

    def gather_compact(self, d_send_ptr: int, n: int, d_recv_ptr: int, root: int = 0, stream: int = 0):
        L = lib()
        rc = L.kaiju_gpu_gather_compact(self._h, d_send_ptr, n, d_recv_ptr, root, stream)
        if rc != 0:
            raise KaijuGpuError(f"{L.kaiju_gpu_strerror(rc).decode()} ({rc}): {L.kaiju_gpu_comm_last_error().decode()}")

    def close(self):
        if self._h:
            lib().kaiju_gpu_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def write_index_image(fmi_path: str, image_path: str):
    """pack a .fmi once into the HBM layout and write it as a device image (loads with Index(image_path))"""
    L = lib()
    L.kaiju_gpu_index_write_image.argtypes = [C.c_char_p, C.c_char_p]
    _check(L.kaiju_gpu_index_write_image(fmi_path.encode(), image_path.encode()))


class Taxonomy:
    def __init__(self, nodes_dmp: str):
        self._h = C.c_void_p()
        _check(lib().kaiju_taxonomy_load(nodes_dmp.encode(), C.byref(self._h)))

    def lca(self, ids):
        a = np.ascontiguousarray(ids, dtype=np.uint64)
        return int(lib().kaiju_taxonomy_lca(self._h, a.ctypes.data, len(a)))

    def close(self):
        if self._h:
            lib().kaiju_taxonomy_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceTaxonomy:
    """nodes.dmp in HBM for the LCA kernel (kaiju_gpu_taxonomy_upload)."""

    def __init__(self, tax: Taxonomy, device: int = 0):
        self._h = C.c_void_p()
        _check(lib().kaiju_gpu_taxonomy_upload(tax._h, device, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().kaiju_gpu_taxonomy_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Classifier:
    """One classification stream on one GPU (the ConsumerThread of the reference)."""

    def __init__(self, index: Index, params: Params):
        self.index = index
        self.params = params
        self._h = C.c_void_p()
        _check(lib().kaiju_gpu_create(C.byref(self._h), index._h, C.byref(params)))

    def classify(self, seqs: np.ndarray, off: np.ndarray, paired=False) -> np.ndarray:
        """Host buffers in, hit records out (blocking)."""
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n = (len(off) - 1) // 2
        hits = np.zeros(n, dtype=HIT_DTYPE)
        _check(lib().kaiju_gpu_classify_batch(self._h, seqs.ctypes.data, off.ctypes.data, n,
                                              1 if paired else 0, hits.ctypes.data))
        return hits

    def set_max_read_length(self, n: int):
        """upper bound of the read lengths given to classify_device (sizes scratch, picks the staged kernel)"""
        _check(lib().kaiju_gpu_set_max_read_length(self._h, int(n)))

    def classify_device(self, d_seqs_ptr: int, seq_bytes: int, d_off_ptr: int, n: int, d_out_ptr: int,
                        paired=False, stream: int = 0):
        """Device-resident buffers (raw pointers, e.g. torch ``data_ptr()``); asynchronous on ``stream``."""
        _check(lib().kaiju_gpu_classify_batch_device(self._h, d_seqs_ptr, seq_bytes, d_off_ptr, n,
                                                     1 if paired else 0, d_out_ptr, stream))

    def stream_handle(self) -> int:
        """the context's own HIP stream (hipStream_t as an integer): wrap it with torch.cuda.ExternalStream to queue
        torch work (a collective, a copy) behind a batch that was launched with stream=0"""
        h = C.c_void_p()
        lib().kaiju_gpu_get_stream.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        _check(lib().kaiju_gpu_get_stream(self._h, C.byref(h)))
        return int(h.value or 0)

    def synchronize(self):
        _check(lib().kaiju_gpu_synchronize(self._h))

    def stats(self) -> Stats:
        s = Stats()
        _check(lib().kaiju_gpu_get_stats(self._h, C.byref(s)))
        return s

    def finalize(self, tax: Taxonomy, hits: np.ndarray, off: np.ndarray, paired=False) -> np.ndarray:
        """E-value gate + LCA + C/U decision on the host (ConsumerThread.cpp:500-513,538,625,724-739)."""
        hits = np.ascontiguousarray(hits)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n = len(hits)
        res = np.zeros(n, dtype=RESULT_DTYPE)
        _check(lib().kaiju_finalize_hits(tax._h, C.byref(self.params), self.index.db_length, hits.ctypes.data,
                                         off.ctypes.data, n, 1 if paired else 0, res.ctypes.data))
        return res

    def lca_device(self, dtax: "DeviceTaxonomy", d_hits_ptr: int, n: int, d_out_ptr: int, stream: int = 0):
        """hit records -> 16-byte compact records (LCA on the device); asynchronous on ``stream``."""
        _check(lib().kaiju_gpu_lca_batch_device(self._h, dtax._h, d_hits_ptr, n, d_out_ptr, stream))

    def classify_device_compact(self, dtax: "DeviceTaxonomy", d_seqs_ptr: int, seq_bytes: int, d_off_ptr: int, n: int,
                                d_hits_ptr: int, d_out_ptr: int, paired=False, stream: int = 0):
        """classify_device + lca_device in one call (kaiju_gpu_classify_batch_device_compact): the search's own post-search
        pass writes the 16-byte records where the configuration allows"""
        _check(lib().kaiju_gpu_classify_batch_device_compact(self._h, dtax._h, d_seqs_ptr, seq_bytes, d_off_ptr, n,
                                                             1 if paired else 0, d_hits_ptr, d_out_ptr, stream))

    def classify_verbose_raw(self, seqs: np.ndarray, off: np.ndarray, paired=False):
        """kaiju -v, the library call alone (kaiju_gpu_classify_batch_verbose): (hit records, kaiju_gpu_verbose records, the rows
        of column-7 text, their stride); the output arrays are kept between calls of the same size"""
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n = (len(off) - 1) // 2
        maxpair = int((off[2::2] - off[0:-1:2]).max()) if n else 0
        lib().kaiju_gpu_verbose_text_stride.restype = C.c_uint32
        lib().kaiju_gpu_verbose_text_stride.argtypes = [C.c_uint32, C.c_int]
        stride = int(lib().kaiju_gpu_verbose_text_stride(maxpair, int(self.params.input_is_protein)))
        kept = getattr(self, "_vb_out", None)
        if kept is None or kept[0] != (n, stride):
            kept = ((n, stride), np.zeros(n, dtype=HIT_DTYPE), np.zeros(n, dtype=VERBOSE_DTYPE), np.zeros(n * stride, dtype=np.uint8))
            self._vb_out = kept
        _, hits, v, text = kept
        _check(lib().kaiju_gpu_classify_batch_verbose(self._h, seqs.ctypes.data, off.ctypes.data, n, 1 if paired else 0,
                                                      hits.ctypes.data, v.ctypes.data, text.ctypes.data, stride))
        return hits, v, text, stride

    def classify_verbose_packed(self, seqs: np.ndarray, off: np.ndarray, paired=False):
        """kaiju -v with column 7 of the batch as ONE string (kaiju_gpu_classify_batch_verbose_packed, what the command line
        programs call): (hit records, kaiju_gpu_verbose records, per read the position of its text, a COPY of the string -
        the library's own is valid until the context's next verbose call)"""
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n = (len(off) - 1) // 2
        hits = np.zeros(n, dtype=HIT_DTYPE)
        v = np.zeros(n, dtype=VERBOSE_DTYPE)
        pos = np.zeros(n, dtype=np.uint64)
        text = C.c_void_p()
        nbytes = C.c_uint64()
        f = lib().kaiju_gpu_classify_batch_verbose_packed
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                      C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        f.restype = C.c_int
        _check(f(self._h, seqs.ctypes.data, off.ctypes.data, n, 1 if paired else 0, hits.ctypes.data, v.ctypes.data, pos.ctypes.data,
                 C.byref(text), C.byref(nbytes)))
        return hits, v, pos, (C.string_at(text.value, nbytes.value) if nbytes.value else b"")

    def verbose_columns(self, v, text, stride):
        """(per read the sorted accession list of column 6, the text of column 7) from what classify_verbose_raw returned"""
        n = len(v)
        accs, peps = [], []
        for r in range(n):
            names = set()
            for q in range(int(v[r]["n_acc"])):
                nm = lib().kaiju_gpu_index_seq_name(self.index._h, int(v[r]["acc_iseq"][q]))
                if nm and b"_" in nm:
                    names.add(nm[: nm.rindex(b"_")].decode())
            accs.append(sorted(names))
            peps.append(bytes(text[r * stride: r * stride + int(v[r]["text_len"])]).decode())
        return accs, peps

    def classify_verbose(self, seqs: np.ndarray, off: np.ndarray, paired=False):
        """kaiju -v: (hit records, per read the sorted accession list of column 6, the text of column 7)"""
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n = (len(off) - 1) // 2
        hits = np.zeros(n, dtype=HIT_DTYPE)
        v = np.zeros(n, dtype=VERBOSE_DTYPE)
        maxpair = int((off[2::2] - off[0:-1:2]).max()) if n else 0
        lib().kaiju_gpu_verbose_text_stride.restype = C.c_uint32
        lib().kaiju_gpu_verbose_text_stride.argtypes = [C.c_uint32, C.c_int]
        stride = int(lib().kaiju_gpu_verbose_text_stride(maxpair, int(self.params.input_is_protein)))
        text = np.zeros(n * stride, dtype=np.uint8)
        _check(lib().kaiju_gpu_classify_batch_verbose(self._h, seqs.ctypes.data, off.ctypes.data, n, 1 if paired else 0,
                                                      hits.ctypes.data, v.ctypes.data, text.ctypes.data, stride))
        accs, peps = [], []
        for r in range(n):
            names = set()
            for q in range(int(v[r]["n_acc"])):
                nm = lib().kaiju_gpu_index_seq_name(self.index._h, int(v[r]["acc_iseq"][q]))
                if nm and b"_" in nm:
                    names.add(nm[: nm.rindex(b"_")].decode())
            accs.append(sorted(names))
            peps.append(bytes(text[r * stride: r * stride + int(v[r]["text_len"])]).decode())
        return hits, accs, peps

    def classify_compact(self, dtax: "DeviceTaxonomy", seqs: np.ndarray, off: np.ndarray, paired=False, out=None) -> np.ndarray:
        """Host buffers in, 16-byte records (LCA on the device) out; blocking (kaiju_gpu_classify_batch_compact).
        Arrays are used as they are (no copies): pass page-locked memory for full PCIe rates."""
        n = (len(off) - 1) // 2
        if out is None:
            out = np.zeros(n, dtype=COMPACT_DTYPE)
        assert seqs.dtype == np.uint8 and off.dtype == np.uint64 and seqs.flags.c_contiguous and off.flags.c_contiguous
        _check(lib().kaiju_gpu_classify_batch_compact(self._h, dtax._h, seqs.ctypes.data, off.ctypes.data, n,
                                                      1 if paired else 0, out.ctypes.data))
        return out

    OP_COUNT_NAMES = ("kmer_lookups", "update_si", "update_si_lines", "lf_steps", "lf_lines", "sa_samples", "read_meta",
                      "frag_desc", "window_fills", "term_searches", "si_spills", "hits", "multi_letter_steps", "items_read",
                      "matches_read", "items_written", "matches_written", "wave_iterations", "lane_iterations", "record_bytes", "pruned_chains", "window_lines")

    def count_ops(self, on: bool):
        """accounting: the next batches run the counting instantiation of the search lane (never a timed launch)"""
        _check(lib().kaiju_gpu_set_count_ops(self._h, 1 if on else 0))

    def op_counts(self) -> dict:
        v = np.zeros(len(self.OP_COUNT_NAMES), dtype=np.uint64)
        _check(lib().kaiju_gpu_get_op_counts(self._h, v.ctypes.data, len(v)))
        return {k: int(x) for k, x in zip(self.OP_COUNT_NAMES, v)}

    def lca(self, dtax: "DeviceTaxonomy", hits: np.ndarray) -> np.ndarray:
        """host hit records -> compact records through the LCA kernel (blocking)"""
        hits = np.ascontiguousarray(hits)
        out = np.zeros(len(hits), dtype=COMPACT_DTYPE)
        _check(lib().kaiju_gpu_lca_batch(self._h, dtax._h, hits.ctypes.data, len(hits), out.ctypes.data))
        return out

    def finalize_compact(self, recs: np.ndarray, off: np.ndarray, paired=False) -> np.ndarray:
        """E-value gate + C/U decision for compact records (their LCA was computed on the device)."""
        recs = np.ascontiguousarray(recs)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n = len(recs)
        res = np.zeros(n, dtype=RESULT_DTYPE)
        _check(lib().kaiju_finalize_compact(C.byref(self.params), self.index.db_length, recs.ctypes.data,
                                            off.ctypes.data, n, 1 if paired else 0, res.ctypes.data))
        return res

    def close(self):
        if self._h:
            lib().kaiju_gpu_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
