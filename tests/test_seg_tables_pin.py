"""The ln(n!) table of SEG (blast_seg.c:52-1308, 10 001 entries printed with six decimals) is DERIVED - from lgamma, rounded to
six decimals - in the product (kaiju_amd/csrc/host_tables.cpp: build_seg_tables) and in the oracle (oracle/kaiju_oracle.c:
init_lnfact) alike: a shared assumption.  Where /root/reference exists the printed table is parsed and compared entry by
entry, bit for bit, with both; everywhere the two derivations are compared with each other."""
import ctypes as C
import os
import re

import pytest

import pyoracle as po

BLAST_SEG = "/root/reference/src/include/ncbi-blast+/algo/blast/core/blast_seg.c"


def _tables(emu, golden):
    h = emu.load(golden.fmi)
    emu.lib.emu_lnfact.restype = C.c_double
    emu.lib.emu_lnfact.argtypes = [C.c_void_p, C.c_uint32]
    emu.lib.emu_lnfact_n.restype = C.c_uint32
    emu.lib.emu_lnfact_n.argtypes = [C.c_void_p]
    product = [emu.lib.emu_lnfact(h, n) for n in range(emu.lib.emu_lnfact_n(h))]
    O = po.Oracle()
    O.lib.ko_lnfact.restype = C.c_double
    O.lib.ko_lnfact.argtypes = [C.c_int]
    oracle = [O.lib.ko_lnfact(n) for n in range(O.lib.ko_lnfact_n())]
    return product, oracle


def test_product_and_oracle_derive_the_same_table(emu, golden):
    product, oracle = _tables(emu, golden)
    assert len(product) == len(oracle) == 10001
    assert product == oracle                      # doubles compared exactly
    assert product[0] == 0.0 and product[2] == 0.693147 and product[20] == 42.335616


def test_table_equals_the_one_printed_in_blast_seg_c(emu, golden):
    if not os.path.exists(BLAST_SEG):
        pytest.skip("the reference's sources are not on this machine")
    text = open(BLAST_SEG).read()
    body = text[text.index("double lnfact[]"):]
    body = body[body.index("{") + 1: body.index("};")]
    printed = [float(x) for x in re.findall(r"-?\d+\.\d+", body)]
    product, oracle = _tables(emu, golden)
    assert len(printed) == 10001
    assert printed == product and printed == oracle
