"""Index construction: protein FASTA -> Kaiju ``.fmi`` (the off-line step of kaiju-makedb,
util/kaiju-makedb:373-375: ``kaiju-mkbwt -a ACDEFGHIKLMNPQRSTVWY -e 3`` + ``kaiju-mkfmi``).
The file is format-compatible with the reference, which can read it unchanged.

Test / benchmark infrastructure (csrc/mkfmi.h): a library of its own, kaiju_amd/libkaiju_mkfmi.so, not part of the product's
C-ABI - the GPU box has no network and no reference binaries, bench.py and the tests make their indexes with it."""
from __future__ import annotations

import ctypes as C

from . import api, build as _build

_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(_build.build_mkfmi())
    return _LIB


def build_fmi(faa_path: str, out_fmi_path: str, threads: int = 0, exponent: int = 3) -> str:
    L = _lib()
    L.kaiju_build_fmi.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int]
    L.kaiju_build_fmi_error.restype = C.c_char_p
    rc = L.kaiju_build_fmi(faa_path.encode(), out_fmi_path.encode(), threads, exponent)
    if rc != 0:
        raise api.KaijuGpuError(f"kaiju_build_fmi failed ({rc}): {L.kaiju_build_fmi_error().decode()}")
    return out_fmi_path


def build_fmi_replicated(faa_path: str, out_fmi_path: str, copies: int, threads: int = 0, exponent: int = 3, copy_taxids=None) -> str:
    """the .fmi of the database in which every sequence of the FASTA occurs `copies` times in a row (kaiju_build_fmi_replicated:
    no second sort; test / benchmark infrastructure for indexes of 2^32 rows and more)"""
    import numpy as np
    L = _lib()
    L.kaiju_build_fmi_replicated.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_uint64, C.c_void_p, C.c_uint32]
    L.kaiju_build_fmi_error.restype = C.c_char_p
    tx = np.ascontiguousarray(copy_taxids, dtype=np.uint64) if copy_taxids is not None and len(copy_taxids) else None
    rc = L.kaiju_build_fmi_replicated(faa_path.encode(), out_fmi_path.encode(), threads, exponent, int(copies),
                                      tx.ctypes.data if tx is not None else None, len(tx) if tx is not None else 0)
    if rc != 0:
        raise api.KaijuGpuError(f"kaiju_build_fmi_replicated failed ({rc}): {L.kaiju_build_fmi_error().decode()}")
    return out_fmi_path
