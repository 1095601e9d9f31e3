"""The oracle against the committed golden vectors produced by the unmodified reference
(tests/golden/make_golden.py): end-to-end TSVs of `kaiju -v -z 1` for every mode, plus
function-level known answers of FMindex / FMindexCurrent / get_suffix / SeqBufferSeg."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import util


@pytest.fixture(scope="module")
def ox(oracle, golden):
    return oracle.load_fmi(golden.fmi), oracle.load_nodes(golden.nodes)


CASES = [("mem", 1, {}), ("mem", 0, {}), ("greedy", 1, {}), ("greedy", 0, {})]


@pytest.mark.parametrize("mode,seg,kw", CASES)
def test_end_to_end_single(oracle, golden, ox, mode, seg, kw):
    ix, tax = ox
    ref = golden.tsv(f"ref_{mode}_{seg}.tsv")
    hits = oracle.classify(ix, tax, oracle.params(mode, seg=seg), golden.seqs, golden.off)
    got = util.oracle_records(hits)
    assert len(ref) == len(golden.names)
    bad = [(n, g, ref[n]) for n, g in zip(golden.names, got) if g != ref[n]]
    assert not bad, bad[:3]
    assert sum(1 for g in got if g[0] == "C") > 300


@pytest.mark.parametrize("mode,seg,kw", CASES)
def test_end_to_end_paired(oracle, golden, ox, mode, seg, kw):
    ix, tax = ox
    ref = golden.tsv(f"ref_{mode}_{seg}_pe.tsv")
    hits = oracle.classify(ix, tax, oracle.params(mode, seg=seg), golden.pseqs, golden.poff, paired=True)
    got = util.oracle_records(hits)
    bad = [(n, g, ref[n]) for n, g in zip(golden.pnames, got) if g != ref[n]]
    assert not bad, bad[:3]


def test_parameter_variants(oracle, golden, ox):
    ix, tax = ox
    for name, p in (("ref_greedy_e5_s50.tsv", oracle.params("greedy", mismatches=5, min_score=50, min_evalue=10.0)),
                    ("ref_greedy_e0.tsv", oracle.params("greedy", mismatches=0)),
                    ("ref_mem_m15.tsv", oracle.params("mem", min_fragment_length=15))):
        ref = golden.tsv(name)
        got = util.oracle_records(oracle.classify(ix, tax, p, golden.seqs, golden.off))
        bad = [(n, g, ref[n]) for n, g in zip(golden.names, got) if g != ref[n]]
        assert not bad, (name, bad[:3])


def test_fm_known_answers(oracle, golden, ox):
    ix, _ = ox
    with np.load(os.path.join(golden.dir, "kat_fm.npz")) as z:
        ks, fm, kk, cur, suf = z["ks"], z["fm"], z["kk"], z["cur"], z["suf"]
    L = oracle.lib
    for a, k in enumerate(ks):
        for c in range(21):
            assert L.ko_fmindex(ix, c, int(k)) == fm[a, c]
    for a, k in enumerate(kk):
        cc = C.c_int()
        v = L.ko_fmindex_current(ix, int(k), C.byref(cc))
        assert (v, cc.value) == tuple(cur[a])
        iseq, pos = C.c_int32(), C.c_int64()
        L.ko_get_suffix(ix, int(k), C.byref(iseq), C.byref(pos))
        assert (iseq.value, pos.value) == tuple(suf[a])


def test_seg_known_answers(oracle, golden):
    with open(os.path.join(golden.dir, "kat_seg.json")) as f:
        kat = json.load(f)
    nreg = 0
    for aa, regs in kat:
        got = oracle.seg(aa.encode())
        assert got == [tuple(r) for r in regs], aa
        nreg += len(regs)
    assert nreg > 500


def test_fragment_order(oracle):
    """six-frame translation: emission order and key-descending stable queue (ConsumerThread.cpp:190-270)"""
    p = oracle.params("mem", seg=0)
    read = b"ATGGCTGCTAAAGGTTCTGCTCCTGAAGAACTGTTCAAAGGTACCGCTTAAATGCGTCGTAAACTGGCTGCTCTGGAAGAACGTGGTTCTCCGGCTTGA"
    fr = oracle.fragments(p, read)
    keys = [k for k, _ in fr]
    assert keys == sorted(keys, reverse=True)
    assert all(len(s) == k and k >= 11 for k, s in fr)
    assert any(s.startswith(b"MAAKGSAPEELFKGTA") for _, s in fr)
    # a base that is not ACGTU terminates the frame like a stop codon
    fr2 = oracle.fragments(p, read.replace(b"GGT", b"GNT", 1))
    assert not any(s.startswith(b"MAAKGSAPEELFKGTA") for _, s in fr2)


@pytest.mark.parametrize("mode", ["mem", "greedy"])
def test_kaijux_lines(oracle, golden, ox, mode):
    """kaijux semantics of the oracle (ids = database sequences; MEM searched with maxMatches(.., 1) as
    ConsumerThreadx.cpp:135 does) == the lines of the reference's kaijux binary, single and paired"""
    ix, _ = ox
    for seqs, off, names, pe, tsv in ((golden.seqs, golden.off, golden.names, False, f"refx_{mode}.tsv"),
                                      (golden.pseqs, golden.poff, golden.pnames, True, f"refx_{mode}_pe.tsv")):
        hits = oracle.classify(ix, None, oracle.params(mode, seg=1, kaijux=1), seqs, off, paired=pe)
        lines = {}
        with open(os.path.join(golden.dir, tsv)) as f:
            for line in f:
                q = line.rstrip("\n").split("\t")
                lines[q[1]] = q
        for h, nm in zip(hits, names):
            ref = lines[nm]
            if h["classified"]:
                ids = sorted(int(x) for x in h["taxid"][:h["n_ids"]])
                got = "".join(oracle.lib.ko_seq_name(ix, i).decode() + "," for i in ids)
                assert ref[0] == "C" and int(ref[2]) == int(h["best"]) and ref[3] == got, (mode, pe, nm, ref, got)
            else:
                assert ref[0] == "U", (mode, pe, nm, ref)


@pytest.mark.parametrize("mode,seg,kw", CASES)
def test_protein_input(oracle, golden, ox, mode, seg, kw):
    """protein reads (kaiju -p, ConsumerThread.cpp:640-646,659-696): the oracle's protein mode == the reference's lines"""
    ix, tax = ox
    ref = golden.tsv(f"refp_{mode}_{seg}.tsv")
    hits = oracle.classify(ix, tax, oracle.params(mode, seg=seg, protein=1), golden.prot_seqs, golden.prot_off)
    got = util.oracle_records(hits)
    assert len(ref) == len(golden.prot_names)
    bad = [(n, g, ref[n]) for n, g in zip(golden.prot_names, got) if g != ref[n]]
    assert not bad, bad[:3]
    assert sum(1 for g in got if g[0] == "C") > 150


@pytest.mark.parametrize("mode", ["mem", "greedy"])
def test_kaijup_lines(oracle, golden, ox, mode):
    """kaijup (ConsumerThreadp.cpp): protein reads, database sequences instead of taxa"""
    ix, _ = ox
    hits = oracle.classify(ix, None, oracle.params(mode, seg=1, kaijux=1, protein=1), golden.prot_seqs, golden.prot_off)
    lines = {}
    with open(os.path.join(golden.dir, f"refpx_{mode}.tsv")) as f:
        for line in f:
            q = line.rstrip("\n").split("\t")
            lines[q[1]] = q
    assert len(lines) == len(golden.prot_fullnames)
    for h, nm in zip(hits, golden.prot_fullnames):
        ref = lines[nm]
        if h["classified"]:
            ids = sorted(int(x) for x in h["taxid"][:h["n_ids"]])
            got = "".join(oracle.lib.ko_seq_name(ix, i).decode() + "," for i in ids)
            assert ref[0] == "C" and int(ref[2]) == int(h["best"]) and ref[3] == got, (mode, nm, ref, got)
        else:
            assert ref[0] == "U", (mode, nm, ref)


@pytest.mark.parametrize("kind", ["prot", "nuc"])
@pytest.mark.parametrize("mode", ["mem", "greedy"])
def test_many_seg_regions(oracle, golden, ox, mode, kind):
    """reads whose single fragment holds more than 15 low-complexity regions (the case the kernels' exact pass exists
    for): oracle == the reference's lines"""
    ix, tax = ox
    names, reads = util.read_fasta(os.path.join(golden.dir, f"regions_{kind}.fa"))
    seqs, off = util.pack(reads)
    ref = golden.tsv(f"refr_{kind}_{mode}.tsv")
    got = util.oracle_records(oracle.classify(ix, tax, oracle.params(mode, seg=1, protein=int(kind == "prot")), seqs, off))
    bad = [(n, g, ref[n]) for n, g in zip(names, got) if g != ref[n]]
    assert not bad, bad[:3]
    assert sum(1 for g in got if g[0] == "C") >= 10
