"""Kernel logic (kaiju_amd/csrc/kj_core.h compiled for the host, tests/emu) against the oracle
and the golden vectors: packed rank blocks, SEG, fragment lists, MEM and Greedy lanes incl. the
scratch-overflow retry pass.  These run without a GPU; the same comparisons run against the real
kernels in test_gpu_parity.py."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import util


@pytest.fixture(scope="module")
def handles(oracle, emu, golden):
    return emu.load(golden.fmi), oracle.load_fmi(golden.fmi), oracle.load_nodes(golden.nodes)


def test_packed_rank_matches_fmindex(oracle, emu, golden, handles):
    h, ix, _ = handles
    with np.load(os.path.join(golden.dir, "kat_fm.npz")) as z:
        ks, fm, kk, cur = z["ks"], z["fm"], z["kk"], z["cur"]
    for a, k in enumerate(ks):
        for c in range(1, 21):
            assert emu.lib.emu_rank(h, c, int(k)) == fm[a, c]
    for a, k in enumerate(kk):
        v, c = cur[a]
        assert emu.lib.emu_symbol(h, int(k)) == c
        assert emu.lib.emu_rank(h, int(c), int(k)) == v      # LF step incl. the terminator (c == 0)
    assert emu.lib.emu_index_warnings(h) == 0


@pytest.mark.parametrize("prefix", [True, False])
def test_seg_known_answers(oracle, emu, golden, handles, prefix, monkeypatch):
    """s_Trim's sub-windows from the prefix counts of the raw segment (what k_seg does) and each counting its own letters:
    the reference's known answers, and the oracle's regions on low-complexity strings of 12 .. 150 residues (raw segments
    beyond 63 residues take the generic window function)"""
    if not prefix:
        monkeypatch.setenv("KAIJU_EMU_NO_SEG_PREFIX", "1")
    h = handles[0]
    with open(os.path.join(golden.dir, "kat_seg.json")) as f:
        kat = json.load(f)
    for aa, regs in kat:
        assert emu.seg(h, aa.encode()) == [tuple(r) for r in regs], aa
    rng = np.random.default_rng(77)
    letters = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
    nlow = 0
    for _ in range(1500):
        n = int(rng.integers(12, 151))
        few = rng.choice(letters, size=int(rng.integers(1, 5)), replace=False)
        aa = rng.choice(letters, size=n)
        a = int(rng.integers(0, n))
        b = min(n, a + int(rng.integers(8, 90)))
        aa[a:b] = rng.choice(few, size=b - a)                     # a low-complexity stretch
        aa = aa.tobytes()
        want = oracle.seg(aa)
        nlow += bool(want)
        if len(want) <= 15:
            assert emu.seg(h, aa) == [tuple(r) for r in want], aa
    assert nlow > 1000


def test_fragment_lists(oracle, emu, golden, handles):
    """stage 1 without SEG == getAllFragmentsBits queue order of the oracle"""
    h = handles[0]
    for mode in ("mem", "greedy"):
        _, _, dump = emu.classify(h, util.gp(mode, seg=0), golden.seqs, golden.off, want_frags=True)
        per_read = dump.split("#\n")[1:]
        p = oracle.params(mode, seg=0)
        for i, r in enumerate(golden.reads):
            exp = oracle.fragments(p, r) if len(r) >= 33 else []
            got = [(int(x.split(":")[0]), x.split(":")[1].encode()) for x in per_read[i].split("\n") if x]
            assert got == exp, (mode, i)


CASES = [("mem", 1), ("mem", 0), ("greedy", 1), ("greedy", 0)]


@pytest.mark.parametrize("mode,seg", CASES)
def test_lanes_vs_oracle_single(oracle, emu, golden, handles, mode, seg):
    h, ix, tax = handles
    oh = oracle.classify(ix, tax, oracle.params(mode, seg=seg, use_evalue=0), golden.seqs, golden.off)
    gh, nretry = emu.classify(h, util.gp(mode, seg=seg), golden.seqs, golden.off)
    bad = [i for i in range(len(oh)) if not util.same_hit(oh[i], gh[i])]
    assert not bad, (bad[:5], oh[bad[0]], gh[bad[0]])
    assert not (gh["flags"] & 0xC0000000).any()


@pytest.mark.parametrize("mode,seg", CASES)
def test_lanes_vs_oracle_paired(oracle, emu, golden, handles, mode, seg):
    h, ix, tax = handles
    oh = oracle.classify(ix, tax, oracle.params(mode, seg=seg, use_evalue=0), golden.pseqs, golden.poff, paired=True)
    gh, _ = emu.classify(h, util.gp(mode, seg=seg), golden.pseqs, golden.poff, paired=True)
    bad = [i for i in range(len(oh)) if not util.same_hit(oh[i], gh[i])]
    assert not bad, (bad[:5], oh[bad[0]], gh[bad[0]])


@pytest.mark.parametrize("mode,seg", CASES)
def test_fast_stage1_and_lazy_seg_on_short_reads(oracle, emu, golden, handles, mode, seg, monkeypatch):
    """batches of short reads take the fast stage 1 (build_fragments_fast) and, MEM, the lazy SEG flow; the same batch
    through the general stage 1 (KAIJU_EMU_STAGE1_OLD) must give the same records and the same fragment lists"""
    h, ix, tax = handles
    idx, seqs, off = golden.short()
    assert len(idx) > 600
    oh = oracle.classify(ix, tax, oracle.params(mode, seg=seg, use_evalue=0), seqs, off)
    gh, _, frags_fast = emu.classify(h, util.gp(mode, seg=seg), seqs, off, want_frags=True)
    bad = [i for i in range(len(oh)) if not util.same_hit(oh[i], gh[i])]
    assert not bad, (bad[:5], oh[bad[0]], gh[bad[0]])
    monkeypatch.setenv("KAIJU_EMU_STAGE1_OLD", "1")
    g2, _, frags_old = emu.classify(h, util.gp(mode, seg=seg), seqs, off, want_frags=True)
    assert (g2 == gh).all()
    if not (mode == "mem" and seg):            # (lazy SEG leaves the lists of most reads unsplit: compared only where both split eagerly)
        assert frags_fast == frags_old


@pytest.mark.parametrize("seg", [1, 0])
def test_mem_locate_in_a_pass_of_its_own(oracle, emu, golden, handles, seg, monkeypatch):
    """MEM: the second-generation lanes leave the longest matches of a read in its hit record and mem_locate_read*
    (k_mem_locate* on the device) walk to the ids behind the searches; the first-generation lane, which walks itself
    (KAIJU_EMU_LANE=v1), must give the same records - single reads, pairs, the general set with long reads"""
    h, ix, tax = handles
    _, sseqs, soff = golden.short()
    for seqs, off, pe in ((sseqs, soff, False), (golden.pseqs, golden.poff, True), (golden.seqs, golden.off, False)):
        oh = oracle.classify(ix, tax, oracle.params("mem", seg=seg, use_evalue=0), seqs, off, paired=pe)
        monkeypatch.delenv("KAIJU_EMU_LANE", raising=False)
        gd, _ = emu.classify(h, util.gp("mem", seg=seg), seqs, off, paired=pe)
        monkeypatch.setenv("KAIJU_EMU_LANE", "v1")
        gi, _ = emu.classify(h, util.gp("mem", seg=seg), seqs, off, paired=pe)
        monkeypatch.delenv("KAIJU_EMU_LANE")
        assert (gd == gi).all()
        bad = [i for i in range(len(oh)) if not util.same_hit(oh[i], gd[i])]
        assert not bad, (seg, pe, bad[:5])
        assert not (gd["flags"] & 0x20000000).any()          # kHitLocPending never leaves the library


@pytest.mark.parametrize("mode,seg", CASES)
def test_retry_pass(oracle, emu, golden, handles, mode, seg, monkeypatch):
    """scratch far too small in the main pass: every overflowing read must come out right
    from the retry pass"""
    h, ix, tax = handles
    oh = oracle.classify(ix, tax, oracle.params(mode, seg=seg, use_evalue=0), golden.seqs, golden.off)
    # the second-generation MEM lane keeps two matches in registers; the first-generation lane
    # (which is also what the retry pass itself runs) overflows with a 1-entry buffer
    monkeypatch.setenv("KAIJU_EMU_LANE", "v1")
    gh, nretry = emu.classify(h, util.gp(mode, seg=seg), golden.seqs, golden.off, caps=(1, 12, 2))
    assert nretry > 0
    bad = [i for i in range(len(oh)) if not util.same_hit(oh[i], gh[i])]
    assert not bad, bad[:5]


def test_parameter_variants(oracle, emu, golden, handles):
    h, ix, tax = handles
    for op, g in ((oracle.params("greedy", mismatches=5, min_score=50, use_evalue=0), util.gp("greedy", mismatches=5, min_score=50)),
                  (oracle.params("greedy", mismatches=0, use_evalue=0), util.gp("greedy", mismatches=0)),
                  (oracle.params("greedy", mismatches=1, seed_length=9, use_evalue=0), util.gp("greedy", mismatches=1, seed_length=9)),
                  (oracle.params("mem", min_fragment_length=15), util.gp("mem", m=15)),
                  (oracle.params("mem", min_fragment_length=8), util.gp("mem", m=8))):
        oh = oracle.classify(ix, tax, op, golden.seqs, golden.off)
        gh, _ = emu.classify(h, g, golden.seqs, golden.off)
        bad = [i for i in range(len(oh)) if not util.same_hit(oh[i], gh[i])]
        assert not bad, (g.mode, g.mismatches, bad[:5])


def test_golden_tsv_through_host_seam(emu, golden, handles):
    """device hits -> kaiju_finalize_hits (E-value gate, LCA, C/U) == reference output lines"""
    from kaiju_amd import api
    tax = api.Taxonomy(golden.nodes)
    h = handles[0]
    L = api.lib()
    for mode in ("mem", "greedy"):
        for seg in (1, 0):
            for pe in (False, True):
                seqs, off, names = (golden.pseqs, golden.poff, golden.pnames) if pe else (golden.seqs, golden.off, golden.names)
                gh, _ = emu.classify(h, util.gp(mode, seg=seg), seqs, off, paired=pe)
                p = api.default_params(mode, seg=seg)
                res = np.zeros(len(gh), dtype=api.RESULT_DTYPE)
                # db_length = bwtlen - nseq (Config.cpp:20)
                with open(golden.fmi, "rb") as f:
                    hdr = np.frombuffer(f.read(12), dtype=np.uint8)
                bwtlen = int(hdr[:8].view("<i8")[0]); nseq = int(hdr[8:12].view("<i4")[0])
                rc = L.kaiju_finalize_hits(tax._h, C.byref(p), float(bwtlen - nseq), gh.ctypes.data,
                                           np.ascontiguousarray(off).ctypes.data, len(gh), 1 if pe else 0, res.ctypes.data)
                assert rc == 0
                ref = golden.tsv(f"ref_{mode}_{seg}{'_pe' if pe else ''}.tsv")
                for i, n in enumerate(names):
                    r = ref[n]
                    if r[0] == "C":
                        assert res[i]["classified"] == 1 and int(res[i]["taxon"]) == r[1] and int(res[i]["best"]) == r[2], (mode, seg, pe, n)
                        assert tuple(sorted(int(x) for x in gh[i]["taxid"][:gh[i]["n_ids"]])) == r[3]
                    else:
                        assert res[i]["classified"] == 0, (mode, seg, pe, n)


@pytest.mark.parametrize("lane", ["v1", "wide", None])
def test_mem_lane_generations_agree(oracle, emu, golden, handles, lane, monkeypatch):
    """first-generation (32/64-bit positions) and second-generation MEM lanes give identical records"""
    h, ix, tax = handles
    if lane:
        monkeypatch.setenv("KAIJU_EMU_LANE", lane)
    for seg in (1, 0):
        oh = oracle.classify(ix, tax, oracle.params("mem", seg=seg), golden.seqs, golden.off)
        gh, _ = emu.classify(h, util.gp("mem", seg=seg), golden.seqs, golden.off)
        bad = [i for i in range(len(oh)) if not util.same_hit(oh[i], gh[i])]
        assert not bad, (lane, seg, bad[:5])
        oh = oracle.classify(ix, tax, oracle.params("mem", seg=seg), golden.pseqs, golden.poff, paired=True)
        gh, _ = emu.classify(h, util.gp("mem", seg=seg), golden.pseqs, golden.poff, paired=True)
        bad = [i for i in range(len(oh)) if not util.same_hit(oh[i], gh[i])]
        assert not bad, (lane, seg, "paired", bad[:5])


@pytest.mark.parametrize("lane,gate", [("v1", None), (None, "0"), (None, "1"), (None, None), (None, "7"), ("v3", None), ("v3", "nosplit")])
def test_greedy_lane_generations_agree(oracle, emu, golden, handles, lane, gate, monkeypatch):
    """first-generation Greedy lane, the second-generation lane (any period of its slow part) and the row-pool lane of
    kj_greedy3.h (one slow block per pull, or the whole chain) == oracle"""
    h, ix, tax = handles
    if lane == "v3":
        monkeypatch.setenv("KAIJU_EMU_GREEDY", "3")
        if gate:
            monkeypatch.setenv("KAIJU_EMU_G3_SPLIT", "0")
        lane = gate = None
    if lane:
        monkeypatch.setenv("KAIJU_EMU_LANE", lane)
    if gate:
        monkeypatch.setenv("KAIJU_EMU_GATE", gate)
    for seg in (1, 0):
        for mm in (3, 1, 0):
            g = util.gp("greedy", seg=seg, mismatches=mm)
            oh = oracle.classify(ix, tax, oracle.params("greedy", seg=seg, mismatches=mm), golden.seqs, golden.off)
            gh, _ = emu.classify(h, g, golden.seqs, golden.off)
            bad = [i for i in range(len(oh)) if not util.same_hit(oh[i], gh[i])]
            assert not bad, (lane, gate, seg, mm, bad[:5])
        oh = oracle.classify(ix, tax, oracle.params("greedy", seg=seg), golden.pseqs, golden.poff, paired=True)
        gh, _ = emu.classify(h, util.gp("greedy", seg=seg), golden.pseqs, golden.poff, paired=True)
        bad = [i for i in range(len(oh)) if not util.same_hit(oh[i], gh[i])]
        assert not bad, (lane, gate, seg, "paired", bad[:5])


def test_greedy_chain_pruning_changes_nothing(oracle, emu, golden, handles):
    """greedy_lane2 does not queue a variant on ONE database row whose chain can never hold a match of m letters (kChainPrune,
    kj_chain_hopeless: a mask of differences between fragment and database text): records identical to the lane built without
    it (-DKJ_NO_CHAIN_PRUNE) and to the oracle - golden reads, pairs, long reads; 0 to 5 mismatches, m = 11 and m = 15"""
    import os
    plain = util.Emu(so=os.path.join(util.EMU_DIR, "libkaiju_kernel_emu_noprune.so"), defines=("KJ_NO_CHAIN_PRUNE",))
    h, ix, tax = handles
    h0 = plain.load(golden.fmi)
    lseqs, loff = util.pack(util.long_reads(n=40))
    for seqs, off, pe in ((golden.seqs, golden.off, False), (golden.pseqs, golden.poff, True), (lseqs, loff, False)):
        for kw in (dict(), dict(mismatches=0), dict(mismatches=1), dict(mismatches=5, min_score=50), dict(m=15), dict(seg=0)):
            okw = {("min_fragment_length" if k == "m" else k): v for k, v in kw.items()}
            oh = oracle.classify(ix, tax, oracle.params("greedy", use_evalue=0, **okw), seqs, off, paired=pe)
            a, _ = emu.classify(h, util.gp("greedy", use_evalue=0, **kw), seqs, off, paired=pe)
            b, _ = plain.classify(h0, util.gp("greedy", use_evalue=0, **kw), seqs, off, paired=pe)
            assert (a == b).all(), (pe, kw)
            bad = [i for i in range(len(oh)) if not util.same_hit(oh[i], a[i])]
            assert not bad, (pe, kw, bad[:5])
    # variants on TWO to four rows (kChainRows): a database with mutated copies of its proteins (1 - 15 % substitutions: what
    # bench.py's database holds) - most seeds then lie on a protein and its copy
    import tempfile
    from kaiju_amd import mkfmi, synth
    _, leaves = synth.make_taxonomy(3, 3, 3)
    db = synth.make_db(nseq=4001, seed=31, leaves=leaves, max_len=600)
    with tempfile.TemporaryDirectory() as d:
        synth.write_fasta(db, d + "/db.faa")
        mkfmi.build_fmi(d + "/db.faa", d + "/db.fmi", threads=4, exponent=3)
        seqs, off = synth.pack_reads(synth.make_reads(db, 6000, seed=32))
        oix = oracle.load_fmi(d + "/db.fmi")
        h1, h2 = emu.load(d + "/db.fmi"), plain.load(d + "/db.fmi")
        for kw in (dict(), dict(mismatches=5, min_score=50), dict(m=15)):
            okw = {("min_fragment_length" if k == "m" else k): v for k, v in kw.items()}
            oh = oracle.classify(oix, None, oracle.params("greedy", use_evalue=0, **okw), seqs, off)
            a, _ = emu.classify(h1, util.gp("greedy", use_evalue=0, **kw), seqs, off)
            b, _ = plain.classify(h2, util.gp("greedy", use_evalue=0, **kw), seqs, off)
            assert (a == b).all(), kw
            bad = [i for i in range(len(oh)) if not util.same_hit(oh[i], a[i])]
            assert not bad, (kw, bad[:5])


@pytest.mark.parametrize("g3", [False, True])
def test_greedy_lane2_spill_and_retry(oracle, golden, handles, g3, monkeypatch):
    """the second-generation Greedy lane (and the row-pool lane) built with tiny bounds (-DKJ_G_SMALL): match lengths and queue
    priorities spill from the LDS rows to global scratch, reads beyond the bounds take the retry pass"""
    import os
    if g3:
        monkeypatch.setenv("KAIJU_EMU_GREEDY", "3")
    small = util.Emu(so=os.path.join(util.EMU_DIR, "libkaiju_kernel_emu_small.so"), defines=("KJ_G_SMALL",))
    _, ix, tax = handles
    h = small.load(golden.fmi)
    total_retry = 0
    for seg in (1, 0):
        oh = oracle.classify(ix, tax, oracle.params("greedy", seg=seg), golden.seqs, golden.off)
        gh, nretry = small.classify(h, util.gp("greedy", seg=seg), golden.seqs, golden.off)
        total_retry += nretry
        bad = [i for i in range(len(oh)) if not util.same_hit(oh[i], gh[i])]
        assert not bad, (seg, bad[:5])
    assert total_retry > 0


@pytest.mark.parametrize("mode", ["mem", "greedy"])
def test_long_reads(oracle, emu, handles, mode):
    """reads of 400..3000 nt: fragments longer than the 64-residue window, stage 1 written in place, match
    lists beyond the LDS rows (second-generation Greedy lane spills / retries)"""
    h, ix, tax = handles
    reads = util.long_reads()
    seqs, off = util.pack(reads)
    for seg in (1, 0):
        oh = oracle.classify(ix, tax, oracle.params(mode, seg=seg, use_evalue=0), seqs, off)
        gh, _ = emu.classify(h, util.gp(mode, seg=seg), seqs, off)
        bad = [i for i in range(len(oh)) if not util.same_hit(oh[i], gh[i])]
        assert not bad, (mode, seg, bad[:5], len(reads[bad[0]]))


@pytest.mark.parametrize("mode,lanes", [("mem", "v2"), ("mem", "v1"), ("mem", "wide16"), ("mem", "wide19-walk"),
                                        ("greedy", "v2"), ("greedy", "v1"), ("greedy", "wide16"), ("greedy", "wide19-walk")])
def test_verbose_columns(emu, golden, handles, mode, lanes, monkeypatch):
    """columns 6 (accessions) and 7 (matched peptides) of kaiju -v == the reference's lines (single and paired, SEG on and
    off).  From the VERBOSE instantiations of the second-generation lanes + mem_verbose_read (v2; wide*: the same with
    64-bit positions forced on the golden index) and from the first-generation lanes (v1: what the retry and exact passes
    still run)"""
    import ctypes as C
    import os
    from kaiju_amd import api
    h = handles[0]
    if lanes == "v1":
        monkeypatch.setenv("KAIJU_EMU_VERBOSE_V1", "1")
    if lanes.startswith("wide"):
        monkeypatch.setenv("KAIJU_GPU_FORCE_WIDE", lanes[4:6])
        if lanes.endswith("walk"):
            monkeypatch.setenv("KAIJU_EMU_NO_ROW_TAX", "1")
        h = emu.load(golden.fmi)
    E = emu.lib
    E.emu_set_verbose.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    E.emu_seq_name.restype = C.c_char_p
    E.emu_seq_name.argtypes = [C.c_void_p, C.c_uint32]
    E.emu_alphabet.restype = C.c_char_p
    E.emu_alphabet.argtypes = [C.c_void_p]
    alpha = E.emu_alphabet(h)
    tax = api.Taxonomy(golden.nodes)
    cap = 8192
    try:
        for seg in (1, 0):
            for seqs, off, names, pe, tsv in ((golden.seqs, golden.off, golden.names, False, f"ref_{mode}_{seg}.tsv"),
                                              (golden.pseqs, golden.poff, golden.pnames, True, f"ref_{mode}_{seg}_pe.tsv")):
                n = len(names)
                nacc = np.zeros(n, dtype=np.uint32); acc = np.zeros(n * 20, dtype=np.uint32)
                tlen = np.zeros(n, dtype=np.uint32); text = np.zeros(n * cap, dtype=np.uint8)
                E.emu_set_verbose(nacc.ctypes.data, acc.ctypes.data, tlen.ctypes.data, text.ctypes.data, cap)
                gh, _ = emu.classify(h, util.gp(mode, seg=seg), seqs, off, paired=pe)
                lines = {}
                with open(os.path.join(golden.dir, tsv)) as f:
                    for line in f:
                        p = line.rstrip("\n").split("\t")
                        lines[p[1]] = p
                for r, nm in enumerate(names):
                    ref = lines[nm]
                    if ref[0] != "C":
                        continue
                    accs = set()
                    for q in range(int(nacc[r])):
                        s = E.emu_seq_name(h, int(acc[r * 20 + q]))
                        if s and b"_" in s:
                            accs.add(s[: s.rindex(b"_")].decode())
                    t = "".join("," if c == 255 else chr(alpha[c]) for c in text[r * cap: r * cap + int(tlen[r])])
                    assert ref[5] == "".join(x + "," for x in sorted(accs)) and ref[6] == t, (mode, seg, pe, nm, ref[5:], accs, t)
    finally:
        E.emu_set_verbose(None, None, None, None, 0)


@pytest.mark.parametrize("shift,rowtax", [("16", True), ("19", True), ("19", False)])
def test_wide_mem_lane(oracle, emu, golden, handles, shift, rowtax, monkeypatch):
    """second-generation MEM lane with 64-bit positions (indexes of 2^32 rows and more), forced on the golden index:
    rank counts relative to a base every 2^shift rows, 16-byte k-mer entries; the ids through the row -> taxon table
    (mem_locate_read<true>, <true, true>: where HBM has room for it) and through the walks of a team (without)"""
    monkeypatch.setenv("KAIJU_GPU_FORCE_WIDE", shift)
    if not rowtax:
        monkeypatch.setenv("KAIJU_EMU_NO_ROW_TAX", "1")
    h = emu.load(golden.fmi)
    _, ix, tax = handles
    reads = util.long_reads(n=30)
    lseqs, loff = util.pack(reads)
    for seg in (1, 0):
        for seqs, off, pe in ((golden.seqs, golden.off, False), (golden.pseqs, golden.poff, True), (lseqs, loff, False)):
            oh = oracle.classify(ix, tax, oracle.params("mem", seg=seg), seqs, off, paired=pe)
            gh, _ = emu.classify(h, util.gp("mem", seg=seg), seqs, off, paired=pe)
            bad = [i for i in range(len(oh)) if not util.same_hit(oh[i], gh[i])]
            assert not bad, (shift, seg, pe, bad[:5])


@pytest.mark.parametrize("shift,rowtax", [("16", True), ("19", True), ("19", False), ("18/tv0", True)])
def test_wide_greedy_lane(oracle, emu, golden, handles, shift, rowtax, monkeypatch):
    """second-generation Greedy lane with 64-bit positions (greedy_lane2<COUNT, WIDE = true>), forced on the golden index:
    the k-mer table of 16-byte entries, block counts relative to a base every 2^shift rows, sequence numbers at the sampled rows,
    queue items and match records in their wide packing; single reads, pairs, long reads, parameter variants.  "/tv0": the
    text position of EVERY row is kept - the chain test of the narrow lane (kChainPrune) then runs in the wide one as well"""
    if shift.endswith("/tv0"):
        shift = shift[:-4]
        monkeypatch.setenv("KAIJU_EMU_TV_SHIFT", "0")
    monkeypatch.setenv("KAIJU_GPU_FORCE_WIDE", shift)
    if not rowtax:
        monkeypatch.setenv("KAIJU_EMU_NO_ROW_TAX", "1")
    h = emu.load(golden.fmi)
    _, ix, tax = handles
    reads = util.long_reads(n=30)
    lseqs, loff = util.pack(reads)
    for seg in (1, 0):
        for seqs, off, pe in ((golden.seqs, golden.off, False), (golden.pseqs, golden.poff, True), (lseqs, loff, False)):
            oh = oracle.classify(ix, tax, oracle.params("greedy", seg=seg, use_evalue=0), seqs, off, paired=pe)
            gh, nretry = emu.classify(h, util.gp("greedy", seg=seg), seqs, off, paired=pe)
            bad = [i for i in range(len(oh)) if not util.same_hit(oh[i], gh[i])]
            assert not bad, (shift, seg, pe, bad[:5])
    for kw in (dict(mismatches=0), dict(mismatches=5, min_score=50), dict(m=15)):
        okw = {("min_fragment_length" if k == "m" else k): v for k, v in kw.items()}
        oh = oracle.classify(ix, tax, oracle.params("greedy", seg=1, use_evalue=0, **okw), golden.seqs, golden.off)
        gh, _ = emu.classify(h, util.gp("greedy", seg=1, use_evalue=0, **kw), golden.seqs, golden.off)
        bad = [i for i in range(len(oh)) if not util.same_hit(oh[i], gh[i])]
        assert not bad, (shift, kw, bad[:5])


def test_randomised_databases_and_parameters():
    """a few rounds of tests/tools/fuzz_emu.py: random databases with repeats and low-complexity stretches, reads
    with Ns / lower case / odd lengths / pairs, random parameters (the long hunts run from the command line)"""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("fuzz_emu", os.path.join(util.ROOT, "tests", "tools", "fuzz_emu.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main(rounds=5, seed=99, first=0) == 0


@pytest.mark.parametrize("mode", ["mem", "greedy"])
def test_kaijux_semantics(emu, golden, mode):
    """kaijux (ConsumerThreadx.cpp): hits collect database sequences instead of taxa - the same kernels on an index whose
    "taxon id" of sequence i is i; score, names (in sequence order) and the C/U decision == the reference's kaijux lines"""
    import ctypes as C
    import os
    from kaiju_amd import api
    E = emu.lib
    E.emu_index_load_x.restype = C.c_void_p
    E.emu_index_load_x.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    E.emu_seq_name.restype = C.c_char_p
    E.emu_seq_name.argtypes = [C.c_void_p, C.c_uint32]
    err = C.create_string_buffer(256)
    h = E.emu_index_load_x(golden.fmi.encode(), err, 256)
    assert h, err.value
    with open(golden.fmi, "rb") as f:
        hdr = np.frombuffer(f.read(12), dtype=np.uint8)
    db_length = float(int(hdr[:8].view("<i8")[0]) - int(hdr[8:12].view("<i4")[0]))
    L = api.lib()
    p = api.default_params(mode, seg=1)
    for seqs, off, names, pe, tsv in ((golden.seqs, golden.off, golden.names, False, f"refx_{mode}.tsv"),
                                      (golden.pseqs, golden.poff, golden.pnames, True, f"refx_{mode}_pe.tsv")):
        gh, _ = emu.classify(h, util.gp(mode, seg=1), seqs, off, paired=pe)
        recs = np.zeros(len(gh), dtype=api.COMPACT_DTYPE)
        recs["lca"] = (gh["n_ids"] > 0).astype(np.uint64)
        recs["best"] = gh["best"]
        recs["info"] = gh["n_ids"]
        res = np.zeros(len(gh), dtype=api.RESULT_DTYPE)
        assert L.kaiju_finalize_compact(C.byref(p), db_length, recs.ctypes.data, np.ascontiguousarray(off).ctypes.data, len(gh),
                                        1 if pe else 0, res.ctypes.data) == 0
        lines = {}
        with open(os.path.join(golden.dir, tsv)) as f:
            for line in f:
                q = line.rstrip("\n").split("\t")
                lines[q[1]] = q
        for r, nm in enumerate(names):
            ref = lines[nm]
            if res[r]["classified"]:
                ids = sorted(int(x) for x in gh[r]["taxid"][:gh[r]["n_ids"]])
                got = "".join(E.emu_seq_name(h, i).decode() + "," for i in ids)
                assert ref[0] == "C" and int(ref[2]) == int(gh[r]["best"]) and ref[3] == got, (mode, pe, nm, ref, got)
            else:
                assert ref[0] == "U", (mode, pe, nm, ref)


def _verbose_buffers(E, n, cap=8192):
    import ctypes as C
    E.emu_set_verbose.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    E.emu_seq_name.restype = C.c_char_p
    E.emu_seq_name.argtypes = [C.c_void_p, C.c_uint32]
    E.emu_alphabet.restype = C.c_char_p
    E.emu_alphabet.argtypes = [C.c_void_p]
    nacc = np.zeros(n, dtype=np.uint32); acc = np.zeros(n * 20, dtype=np.uint32)
    tlen = np.zeros(n, dtype=np.uint32); text = np.zeros(n * cap, dtype=np.uint8)
    E.emu_set_verbose(nacc.ctypes.data, acc.ctypes.data, tlen.ctypes.data, text.ctypes.data, cap)
    return nacc, acc, tlen, text, cap


def _tsv_lines(path):
    lines = {}
    with open(path) as f:
        for line in f:
            q = line.rstrip("\n").split("\t")
            lines[q[1]] = q
    return lines


@pytest.mark.parametrize("mode", ["mem", "greedy"])
def test_protein_input(oracle, emu, golden, handles, mode):
    """protein reads (kaiju -p): stage 1 = build_fragments_protein, the rest of the kernel sequence unchanged; hit records
    == the oracle's (whose protein mode is pinned on the reference's lines), verbose columns == the reference's"""
    import os
    h, ix, tax = handles
    E = emu.lib
    for seg in (1, 0):
        oh = oracle.classify(ix, tax, oracle.params(mode, seg=seg, protein=1, use_evalue=0), golden.prot_seqs, golden.prot_off)
        gh, _ = emu.classify(h, util.gp(mode, seg=seg, protein=1), golden.prot_seqs, golden.prot_off)
        bad = [i for i in range(len(oh)) if not util.same_hit(oh[i], gh[i])]
        assert not bad, (mode, seg, bad[:5])
        assert sum(1 for g in gh if g["n_ids"]) > 150
        # first-generation lanes
        for lane in ("v1", "wide"):
            os.environ["KAIJU_EMU_LANE"] = lane
            try:
                g1, _ = emu.classify(h, util.gp(mode, seg=seg, protein=1), golden.prot_seqs, golden.prot_off)
            finally:
                del os.environ["KAIJU_EMU_LANE"]
            assert all(util.same_hit(a, b) for a, b in zip(gh, g1)), (mode, seg, lane)
        # columns 6/7 of kaiju -p -v
        n = len(golden.prot_names)
        nacc, acc, tlen, text, cap = _verbose_buffers(E, n)
        try:
            emu.classify(h, util.gp(mode, seg=seg, protein=1), golden.prot_seqs, golden.prot_off)
        finally:
            E.emu_set_verbose(None, None, None, None, 0)
        alpha = E.emu_alphabet(h)
        lines = _tsv_lines(os.path.join(golden.dir, f"refp_{mode}_{seg}.tsv"))
        for r, nm in enumerate(golden.prot_names):
            ref = lines[nm]
            if ref[0] != "C":
                continue
            accs = set()
            for q in range(int(nacc[r])):
                s = E.emu_seq_name(h, int(acc[r * 20 + q]))
                if s and b"_" in s:
                    accs.add(s[: s.rindex(b"_")].decode())
            t = "".join("," if c == 255 else chr(alpha[c]) for c in text[r * cap: r * cap + int(tlen[r])])
            assert ref[5] == "".join(x + "," for x in sorted(accs)) and ref[6] == t, (mode, seg, nm, ref[5:], accs, t)
    # paired protein input does not exist (kaiju.cpp:201)
    assert E.emu_classify(h, util.gp(mode, protein=1), golden.pseqs.ctypes.data, golden.poff.ctypes.data, 4, 1, None, 16, 192, 64,
                          None, None, 0) != 0


@pytest.mark.parametrize("prot", [False, True])
@pytest.mark.parametrize("mode", ["mem", "greedy"])
def test_kaijux_verbose_and_kaijup(emu, golden, mode, prot):
    """kaijux -v / kaijup -v: score, database sequences and the matched peptides == the reference's lines; in MEM mode the
    reference searches with maxMatches(.., 1) (ConsumerThreadx.cpp:135), whose list starts with the match found first:
    the lanes visit the matches in that order (kParamXOrder)"""
    import ctypes as C
    import os
    E = emu.lib
    E.emu_index_load_x.restype = C.c_void_p
    E.emu_index_load_x.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    err = C.create_string_buffer(256)
    h = E.emu_index_load_x(golden.fmi.encode(), err, 256)
    assert h, err.value
    try:
        if prot:
            cases = [(golden.prot_seqs, golden.prot_off, golden.prot_fullnames, False, f"refpx_{mode}_v.tsv")]
        else:
            cases = [(golden.seqs, golden.off, golden.names, False, f"refx_{mode}_v.tsv"),
                     (golden.pseqs, golden.poff, golden.pnames, True, f"refx_{mode}_pe_v.tsv")]
        for seqs, off, names, pe, tsv in cases:
            n = len(names)
            nacc, acc, tlen, text, cap = _verbose_buffers(E, n)
            try:
                gh, _ = emu.classify(h, util.gp(mode, seg=1, protein=int(prot)), seqs, off, paired=pe)
            finally:
                E.emu_set_verbose(None, None, None, None, 0)
            # the second-generation lanes (no verbose output) agree with what the verbose pass found
            g2, _ = emu.classify(h, util.gp(mode, seg=1, protein=int(prot)), seqs, off, paired=pe)
            assert all(util.same_hit(a, b) for a, b in zip(gh, g2)), (mode, prot, pe)
            alpha = E.emu_alphabet(h)
            lines = _tsv_lines(os.path.join(golden.dir, tsv))
            nc = 0
            for r, nm in enumerate(names):
                ref = lines[nm]
                if ref[0] != "C":
                    continue
                nc += 1
                ids = sorted(int(x) for x in gh[r]["taxid"][:gh[r]["n_ids"]])
                got = "".join(E.emu_seq_name(h, i).decode() + "," for i in ids)
                t = "".join("," if c == 255 else chr(alpha[c]) for c in text[r * cap: r * cap + int(tlen[r])])
                assert int(ref[2]) == int(gh[r]["best"]) and ref[3] == got and ref[4] == t, (mode, prot, pe, nm, ref[2:], got, t)
            assert nc > 100
    finally:
        E.emu_index_free(h)


def test_kaijux_mem_order_under_the_id_cap(oracle, emu, tmp_path):
    """a database of near-identical sequences: the collected sequences depend on the order in which the matches of a
    fragment are visited (the 21-id cap cuts the traversal); == the oracle's kaijux mode (maxMatches_limited), ids in
    traversal order, for both generations of the MEM lane"""
    import ctypes as C
    import os
    from kaiju_amd import mkfmi
    faa, fmi = str(tmp_path / "rep.faa"), str(tmp_path / "rep.fmi")
    reads = util.repetitive_db(faa)
    mkfmi.build_fmi(faa, fmi, threads=2, exponent=3)
    E = emu.lib
    E.emu_index_load_x.restype = C.c_void_p
    E.emu_index_load_x.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    err = C.create_string_buffer(256)
    h = E.emu_index_load_x(fmi.encode(), err, 256)
    assert h, err.value
    assert not E.emu_index_warnings(h)
    seqs, off = util.pack(reads)
    ix = oracle.load_fmi(fmi)
    oh = oracle.classify(ix, None, oracle.params("mem", seg=0, kaijux=1), seqs, off)
    assert sum(1 for o in oh if o["flags"] & 1) > 20
    for lane in (None, "v1", "wide"):
        if lane:
            os.environ["KAIJU_EMU_LANE"] = lane
        try:
            gh, _ = emu.classify(h, util.gp("mem", seg=0), seqs, off)
        finally:
            os.environ.pop("KAIJU_EMU_LANE", None)
        bad = [i for i in range(len(oh)) if not util.same_hit(oh[i], gh[i])]
        assert not bad, (lane, bad[:5])
    # the order matters: the plain kaiju order (greedyExact) gives other sets for some of these reads
    h0 = emu.load(fmi)
    g0, _ = emu.classify(h0, util.gp("mem", seg=0), seqs, off)
    o0 = oracle.classify(ix, None, oracle.params("mem", seg=0), seqs, off)
    assert all(int(a["best"]) == int(b["best"]) for a, b in zip(g0, gh))
    E.emu_index_free(h); E.emu_index_free(h0)


def test_protein_long(oracle, emu, golden, handles):
    """proteins of several thousand residues (windows refill, fragments as long as the read) == oracle"""
    h, ix, tax = handles
    rng = np.random.default_rng(3)
    prots = [s for s in golden.prot_reads if len(s) > 100]
    reads = []
    for _ in range(40):
        parts = [prots[int(rng.integers(0, len(prots)))] for _ in range(int(rng.integers(3, 12)))]
        reads.append(b"X".join(parts) if rng.random() < 0.5 else b"".join(parts))
    assert max(len(r) for r in reads) > 4000
    seqs, off = util.pack(reads)
    for mode in ("mem", "greedy"):
        oh = oracle.classify(ix, tax, oracle.params(mode, seg=1, protein=1, use_evalue=0), seqs, off)
        gh, _ = emu.classify(h, util.gp(mode, seg=1, protein=1), seqs, off)
        bad = [i for i in range(len(oh)) if not util.same_hit(oh[i], gh[i])]
        assert not bad, (mode, bad[:5])


@pytest.mark.parametrize("mode", ["mem", "greedy"])
def test_exact_pass_many_seg_regions(oracle, emu, golden, handles, mode, monkeypatch):
    """a fragment with more low-complexity regions than a record of the SEG pass holds (15): the read is classified
    again by the exact pass (region lists of any length) and equals the oracle; protein and nucleotide reads"""
    h, ix, tax = handles
    prot, nuc = util.many_region_reads()
    assert max(len(oracle.seg(s)) for s in prot) > 18
    for reads, protein in ((prot, 1), (nuc, 0), (list(golden.prot_reads[:60]) + prot, 1)):
        seqs, off = util.pack(reads)
        oh = oracle.classify(ix, tax, oracle.params(mode, seg=1, protein=protein, use_evalue=0), seqs, off)
        gh, _ = emu.classify(h, util.gp(mode, seg=1, protein=protein), seqs, off)
        bad = [i for i in range(len(oh)) if not util.same_hit(oh[i], gh[i])]
        assert not bad, (mode, protein, bad[:5])
        assert not (gh["flags"] & 0x80000000).any()
    # a region pool that is too small is reported, not passed over
    monkeypatch.setenv("KAIJU_EMU_REDO_POOL", "8")
    seqs, off = util.pack(prot)
    gh, _ = emu.classify(h, util.gp(mode, seg=1, protein=1), seqs, off, allow_capacity=True)
    assert gh is None


@pytest.mark.parametrize("kind", ["prot", "nuc"])
@pytest.mark.parametrize("mode", ["mem", "greedy"])
def test_exact_pass_verbose_columns(emu, golden, handles, mode, kind):
    """reads that go through the exact pass, verbose run: columns 4-7 == the reference's lines"""
    import os
    from kaiju_amd import api
    h = handles[0]
    E = emu.lib
    names, reads = util.read_fasta(os.path.join(golden.dir, f"regions_{kind}.fa"))
    seqs, off = util.pack(reads)
    n = len(names)
    nacc, acc, tlen, text, cap = _verbose_buffers(E, n)
    try:
        gh, _ = emu.classify(h, util.gp(mode, seg=1, protein=int(kind == "prot")), seqs, off)
    finally:
        E.emu_set_verbose(None, None, None, None, 0)
    alpha = E.emu_alphabet(h)
    lines = _tsv_lines(os.path.join(golden.dir, f"refr_{kind}_{mode}.tsv"))
    nc = 0
    for r, nm in enumerate(names):
        ref = lines[nm]
        if ref[0] != "C":
            assert gh[r]["n_ids"] == 0 or mode == "greedy"
            continue
        nc += 1
        ids = ",".join(str(x) for x in sorted(int(x) for x in gh[r]["taxid"][:gh[r]["n_ids"]])) + ","
        accs = set()
        for q in range(int(nacc[r])):
            s = E.emu_seq_name(h, int(acc[r * 20 + q]))
            if s and b"_" in s:
                accs.add(s[: s.rindex(b"_")].decode())
        t = "".join("," if c == 255 else chr(alpha[c]) for c in text[r * cap: r * cap + int(tlen[r])])
        assert int(ref[3]) == int(gh[r]["best"]) and ref[4] == ids, (mode, kind, nm, ref[3:5], gh[r])
        assert ref[5] == "".join(x + "," for x in sorted(accs)) and ref[6] == t, (mode, kind, nm, ref[5:], accs, t)
    assert nc >= 10


@pytest.mark.parametrize("shift", ["16", "20"])
def test_wide_index_with_kaijux_order_and_protein(oracle, emu, golden, handles, tmp_path, shift, monkeypatch):
    """the 64-bit MEM lane (indexes of 2^32 rows and more, forced here) combined with what came later: the match order of
    kaijux (mem_lane2<true, true>) and protein reads"""
    import ctypes as C
    from kaiju_amd import mkfmi
    monkeypatch.setenv("KAIJU_GPU_FORCE_WIDE", shift)
    E = emu.lib
    E.emu_index_load_x.restype = C.c_void_p
    E.emu_index_load_x.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    # kaijux order under the id cap
    faa, fmi = str(tmp_path / "rep.faa"), str(tmp_path / "rep.fmi")
    reads = util.repetitive_db(faa)
    mkfmi.build_fmi(faa, fmi, threads=2, exponent=3)
    err = C.create_string_buffer(256)
    h = E.emu_index_load_x(fmi.encode(), err, 256)
    assert h, err.value
    seqs, off = util.pack(reads)
    ix = oracle.load_fmi(fmi)
    oh = oracle.classify(ix, None, oracle.params("mem", seg=0, kaijux=1), seqs, off)
    gh, _ = emu.classify(h, util.gp("mem", seg=0), seqs, off)
    assert all(util.same_hit(a, b) for a, b in zip(oh, gh))
    E.emu_index_free(h)
    # protein reads on the golden index
    hw = emu.load(golden.fmi)
    _, gix, tax = handles
    for mode in ("mem", "greedy"):
        oh = oracle.classify(gix, tax, oracle.params(mode, seg=1, protein=1, use_evalue=0), golden.prot_seqs, golden.prot_off)
        gh, _ = emu.classify(hw, util.gp(mode, seg=1, protein=1), golden.prot_seqs, golden.prot_off)
        bad = [i for i in range(len(oh)) if not util.same_hit(oh[i], gh[i])]
        assert not bad, (shift, mode, bad[:5])
    E.emu_index_free(hw)


@pytest.mark.parametrize("seg", [1, 0])
def test_protein_evalue_gate_and_lca(emu, golden, handles, seg):
    """host seam for protein reads: kaiju_finalize_hits with input_is_protein (E-value with query_len = read length,
    ConsumerThread.cpp:660, LCA, C/U) on the emulated hit records == the reference's `kaiju -p -a greedy` lines"""
    import ctypes as C
    from kaiju_amd import api
    h = handles[0]
    gh, _ = emu.classify(h, util.gp("greedy", seg=seg, protein=1), golden.prot_seqs, golden.prot_off)
    with open(golden.fmi, "rb") as f:
        hdr = np.frombuffer(f.read(12), dtype=np.uint8)
    db_length = float(int(hdr[:8].view("<i8")[0]) - int(hdr[8:12].view("<i4")[0]))
    p = api.default_params("greedy", seg=seg, input_is_protein=1)
    tax = api.Taxonomy(golden.nodes)
    res = np.zeros(len(gh), dtype=api.RESULT_DTYPE)
    hits = np.ascontiguousarray(gh)
    off = np.ascontiguousarray(golden.prot_off, dtype=np.uint64)
    assert api.lib().kaiju_finalize_hits(tax._h, C.byref(p), db_length, hits.ctypes.data, off.ctypes.data, len(gh), 0,
                                        res.ctypes.data) == 0
    ref = golden.tsv(f"refp_greedy_{seg}.tsv")
    for nm, g, r in zip(golden.prot_names, gh, res):
        got = ("C", int(r["taxon"]), int(r["best"]), tuple(sorted(int(x) for x in g["taxid"][:g["n_ids"]]))) if r["classified"] \
            else ("U", 0, None, ())
        assert got == ref[nm], (nm, got, ref[nm])


def test_skip_rules_and_probes(oracle, golden, handles, monkeypatch):
    """kj_core.h kSpanRule / kMemProbe / kGreedyProbe pass end positions whose searches cannot be recorded.  Three builds of the
    lanes - as shipped, with the rules compiled out, and with the wide lane's probes two letters longer than the index size
    asks for (probes of several UpdateSI steps on a small index) - give the oracle's records, narrow and forced wide."""
    _, ix, tax = handles
    reads = util.long_reads(n=30)
    lseqs, loff = util.pack(reads)
    builds = [("plain", ()), ("norules", ("KJ_NO_SPAN_RULE", "KJ_NO_PROBE")), ("longprobe", ("KJ_PROBE_W_ADD=2",))]
    for tag, defs in builds:
        e = util.Emu(so=os.path.join(util.EMU_DIR, f"libkaiju_kernel_emu_{tag}.so"), defines=defs) if defs else util.Emu()
        for wide in (None, "17"):
            if wide:
                monkeypatch.setenv("KAIJU_GPU_FORCE_WIDE", wide)
            else:
                monkeypatch.delenv("KAIJU_GPU_FORCE_WIDE", raising=False)
            h = e.load(golden.fmi)
            for mode, kw, okw in (("mem", dict(), dict()), ("mem", dict(m=20), dict(min_fragment_length=20)),
                                  ("greedy", dict(), dict(use_evalue=0)), ("greedy", dict(seed_length=9), dict(use_evalue=0, seed_length=9))):
                for seqs, off, pe in ((golden.seqs, golden.off, False), (golden.pseqs, golden.poff, True), (lseqs, loff, False)):
                    oh = oracle.classify(ix, tax, oracle.params(mode, seg=1, **okw), seqs, off, paired=pe)
                    gh, _ = e.classify(h, util.gp(mode, seg=1, **({"use_evalue": 0} if mode == "greedy" else {}), **kw), seqs, off, paired=pe)
                    bad = [i for i in range(len(oh)) if not util.same_hit(oh[i], gh[i])]
                    assert not bad, (tag, wide, mode, kw, pe, bad[:5])
            e.lib.emu_index_free(h)


def test_text_arrays_on_an_index_with_the_short_sample_array(tmp_path, monkeypatch):
    """One real database in eight (nseq % 2^e == 0) has the reference's short suffix-array sample (KAIJU_IDX_WARN_SA_SHORT).
    Round 3 built no text arrays for such an index - text verification and the two-load locate silently gone.  Now the few
    rows whose walk runs into the missing sample are resolved through the next one; a locate of such a row stays "no
    sequence" (the reference reads out of bounds there).  The lanes with the text arrays must classify exactly like the lanes
    without them (which walk and step as before)."""
    import ctypes as C
    if os.environ.get("KAIJU_GPU_FORCE_WIDE") or os.environ.get("KAIJU_EMU_NO_TEXT"):
        pytest.skip("the text arrays are a narrow-index feature (and switched off by KAIJU_EMU_NO_TEXT)")
    from kaiju_amd import mkfmi, synth
    _, leaves = synth.make_taxonomy(3, 3, 3)
    db = synth.make_db(nseq=1600, seed=11, leaves=leaves, max_len=700)          # 1600 % 8 == 0
    faa, fmi = str(tmp_path / "db.faa"), str(tmp_path / "db.fmi")
    synth.write_fasta(db, faa)
    mkfmi.build_fmi(faa, fmi, threads=2, exponent=3)
    seqs, off = synth.pack_reads(synth.make_reads(db, 12000, seed=5))
    m1, m2 = synth.make_pairs(db, 3000, seed=6)
    pseqs, poff = synth.pack_reads(m1, m2)
    emu = util.Emu()
    emu.lib.emu_rows_without_sequence.restype = C.c_uint64
    emu.lib.emu_rows_without_sequence.argtypes = [C.c_void_p]
    emu.lib.emu_has_text.argtypes = [C.c_void_p]
    h = emu.load(fmi)
    assert emu.lib.emu_index_warnings(h) & 1                                       # KAIJU_IDX_WARN_SA_SHORT
    assert emu.lib.emu_has_text(h) == 1
    assert 1 <= emu.lib.emu_rows_without_sequence(h) <= 4096
    monkeypatch.setenv("KAIJU_EMU_NO_TEXT", "1")
    h0 = emu.load(fmi)
    assert emu.lib.emu_has_text(h0) == 0
    monkeypatch.delenv("KAIJU_EMU_NO_TEXT")
    differ = 0
    for mode in ("mem", "greedy"):
        for s, o, pe in ((seqs, off, False), (pseqs, poff, True)):
            a, _ = emu.classify(h, util.gp(mode), s, o, paired=pe)
            b, _ = emu.classify(h0, util.gp(mode), s, o, paired=pe)
            # (a match that ends on a row behind the missing sample gets no id from that row - undefined in the reference,
            #  skipped by the walking locate, and skipped just so when the text had grown the match: DevIndex::beyond_lo)
            assert (a == b).all(), (mode, pe, np.nonzero(a != b)[0][:5])
            assert (a["n_ids"] > 0).mean() > 0.4
            # ... and that is the rule's doing: without it the text-grown matches that end there are located through another row
            monkeypatch.setenv("KAIJU_EMU_NO_BEYOND_RULE", "1")
            c, _ = emu.classify(h, util.gp(mode), s, o, paired=pe)
            monkeypatch.delenv("KAIJU_EMU_NO_BEYOND_RULE")
            differ += int((c != b).sum())
    assert differ > 0


def test_text_positions_of_an_index_with_64_bit_rows(oracle, tmp_path, monkeypatch):
    """Wide layout: the database text and the text position of every 2^tv_shift-th row (DevIndex::sa_tpos5, built by walking
    every sequence once: seq_walk_len / seq_walk_fill).  A one-row search steps on until it stands on such a row and then
    compares with the text.  Every sample density gives the oracle's records - and the records of the lanes without the text."""
    import ctypes as C
    from kaiju_amd import mkfmi, synth
    _, leaves = synth.make_taxonomy(3, 3, 3)
    db = synth.make_db(nseq=2500, seed=21, leaves=leaves, max_len=900)
    faa, fmi = str(tmp_path / "db.faa"), str(tmp_path / "db.fmi")
    synth.write_fasta(db, faa)
    mkfmi.build_fmi(faa, fmi, threads=2, exponent=3)
    seqs, off = synth.pack_reads(synth.make_reads(db, 4000, seed=9))
    m1, m2 = synth.make_pairs(db, 600, seed=10)
    pseqs, poff = synth.pack_reads(m1, m2)
    ix = oracle.load_fmi(fmi)
    want = {pe: oracle.classify(ix, None, oracle.params("mem", seg=1), s, o, paired=pe) for s, o, pe in ((seqs, off, False), (pseqs, poff, True))}
    emu = util.Emu()
    emu.lib.emu_has_text.argtypes = [C.c_void_p]
    emu.lib.emu_text.restype = C.POINTER(C.c_uint8)
    emu.lib.emu_text.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    emu.lib.emu_text_pos.restype = C.c_uint64
    emu.lib.emu_text_pos.argtypes = [C.c_void_p, C.c_uint64]

    def arrays(h, rows):
        n = C.c_uint64(0)
        t = emu.lib.emu_text(h, C.byref(n))
        return bytes(np.ctypeslib.as_array(t, shape=(n.value,))), [emu.lib.emu_text_pos(h, r) for r in rows]
    # the narrow layout's arrays (one get_suffix walk per row) are what the wide layout's sequence walks must reproduce
    monkeypatch.delenv("KAIJU_GPU_FORCE_WIDE", raising=False)
    h = emu.load(fmi)
    rows = list(range(0, int(db.total_aa) + db.nseq, 7))          # every seventh row of the index
    text_narrow, pos_narrow = arrays(h, rows)
    emu.lib.emu_index_free(h)
    monkeypatch.setenv("KAIJU_GPU_FORCE_WIDE", "18")
    for tv in ("none", "0", "1", "2", "3", "5"):
        if tv == "none":
            monkeypatch.setenv("KAIJU_EMU_NO_TEXT", "1")
        else:
            monkeypatch.delenv("KAIJU_EMU_NO_TEXT", raising=False)
            monkeypatch.setenv("KAIJU_EMU_TV_SHIFT", tv)
        h = emu.load(fmi)
        assert emu.lib.emu_has_text(h) == (0 if tv == "none" else 1)
        if tv != "none":
            text_wide, pos_wide = arrays(h, rows)
            assert text_wide == text_narrow
            keep = (1 << int(tv)) - 1
            assert all(pw == (pn if r & keep == 0 else 2**64 - 1) for r, pn, pw in zip(rows, pos_narrow, pos_wide)), tv
        for s, o, pe in ((seqs, off, False), (pseqs, poff, True)):
            got, _ = emu.classify(h, util.gp("mem"), s, o, paired=pe)
            bad = [i for i in range(len(got)) if not util.same_hit(want[pe][i], got[i])]
            assert not bad, (tv, pe, bad[:5])
        emu.lib.emu_index_free(h)


def test_redundant_databases_and_the_span_rule_for_any_interval_size(oracle, tmp_path, monkeypatch):
    """kj_core.h kSpanEq (the span rule for intervals of ANY size) and the many-rows locate (DevIndex::row_tax, teams) only act
    on databases with near-identical sequences - the i.i.d. databases of the other tests have no multi-row matches to speak of.
    Here: protein families (synth.make_db_hard, reads with Ns) and an index with every protein several times
    (kaiju_build_fmi_replicated), MEM and Greedy, single and paired, narrow and forced wide, against the oracle - with the rule
    as shipped and compiled out (KJ_NO_SPAN_EQ), which must not change a record."""
    from kaiju_amd import mkfmi, synth
    _, leaves = synth.make_taxonomy(4, 4, 4)
    hdb = synth.make_db_hard(nseq=3001, seed=77, leaves=leaves, fam_lo=20, fam_hi=120)
    faa, fmi = str(tmp_path / "hard.faa"), str(tmp_path / "hard.fmi")
    synth.write_fasta(hdb, faa)
    mkfmi.build_fmi(faa, fmi, threads=2, exponent=3)
    idb = synth.make_db(nseq=400, seed=78, leaves=leaves, max_len=600)
    faa2, fmi2 = str(tmp_path / "base.faa"), str(tmp_path / "rep.fmi")
    synth.write_fasta(idb, faa2)
    mkfmi.build_fmi_replicated(faa2, fmi2, 9, threads=2, exponent=3, copy_taxids=np.asarray(leaves, dtype=np.uint64))
    work = []
    for db, f, n in ((hdb, fmi, 1500), (idb, fmi2, 1200)):
        seqs, off = synth.pack_reads(synth.sprinkle_n(synth.make_reads(db, n, seed=5)))
        m1, m2 = synth.make_pairs(db, n // 3, seed=6)
        pseqs, poff = synth.pack_reads(m1, m2)
        work.append((f, ((seqs, off, False), (pseqs, poff, True))))
    builds = [("plain", ()), ("nospaneq", ("KJ_NO_SPAN_EQ",))]
    for f, sets in work:
        ix = oracle.load_fmi(f)
        want = {(mode, pe): oracle.classify(ix, None, oracle.params(mode, seg=1, **({"use_evalue": 0} if mode == "greedy" else {})), s, o, paired=pe)
                for mode in ("mem", "greedy") for s, o, pe in sets}
        many = 0
        for tag, defs in builds:
            e = util.Emu(so=os.path.join(util.EMU_DIR, f"libkaiju_kernel_emu_{tag}.so"), defines=defs) if defs else util.Emu()
            for wide in (None, "17"):
                if wide:
                    monkeypatch.setenv("KAIJU_GPU_FORCE_WIDE", wide)
                else:
                    monkeypatch.delenv("KAIJU_GPU_FORCE_WIDE", raising=False)
                h = e.load(f)
                for mode in ("mem", "greedy"):
                    for s, o, pe in sets:
                        gh, _ = e.classify(h, util.gp(mode, seg=1, **({"use_evalue": 0} if mode == "greedy" else {})), s, o, paired=pe)
                        oh = want[(mode, pe)]
                        bad = [i for i in range(len(oh)) if not util.same_hit(oh[i], gh[i])]
                        assert not bad, (os.path.basename(f), tag, wide, mode, pe, bad[:5])
                        many += int((gh["n_ids"] >= 5).sum())
                e.lib.emu_index_free(h)
        assert many > 100, many          # (matches of many rows under several taxa: what the i.i.d. databases do not have)


def test_reads_with_many_longest_matches(oracle, tmp_path, monkeypatch):
    """Reads whose longest matches are MANY (three to sixteen equally long ones, in one fragment and in several): since round 5
    they too leave their matches in the hit record - in ids_from_SI's visiting order (:835-845), kaijux in maxMatches' list
    order - and the locate kernels turn them into ids (k_mem_locate_list with the row -> taxon table, teams without it).
    More than sixteen go to the retry pass.  Narrow, forced wide, without the text arrays, kaijux ids - against the oracle."""
    from kaiju_amd import mkfmi
    rng = np.random.default_rng(17)
    aa = list("ACDEFGHIKLMNPQRSTVWY")
    codon = {"A": "GCT", "C": "TGT", "D": "GAT", "E": "GAA", "F": "TTT", "G": "GGT", "H": "CAT", "I": "ATT", "K": "AAA", "L": "CTT",
             "M": "ATG", "N": "AAT", "P": "CCT", "Q": "CAA", "R": "CGT", "S": "TCT", "T": "ACT", "V": "GTT", "W": "TGG", "Y": "TAT"}
    nmotif, mlen = 40, 12
    motifs = ["".join(rng.choice(aa, mlen)) for _ in range(nmotif)]
    faa, fmi = str(tmp_path / "m.faa"), str(tmp_path / "m.fmi")
    with open(faa, "w") as f:
        for i, m in enumerate(motifs):
            for c in range(1 + i % 3):                          # a motif lies in one to three sequences (other taxa)
                f.write(f">M{i}c{c}_{10 + (7 * i + c) % 23}\n" + "".join(rng.choice(aa, int(rng.integers(20, 60)))) + m +
                        "".join(rng.choice(aa, int(rng.integers(20, 60)))) + "\n")
    mkfmi.build_fmi(faa, fmi, threads=2, exponent=3)
    reads = []
    for _ in range(300):
        k = int(rng.integers(3, 19))                            # motifs per read: 3 .. 18 (more than sixteen: the retry pass)
        pep = ""
        for _ in range(k):
            pep += motifs[int(rng.integers(0, nmotif))] + "".join(rng.choice(aa, int(rng.integers(1, 4))))
            if rng.random() < 0.25:
                pep += "*"                                        # a stop: the next motifs lie in another fragment
        nt = "".join("TAA" if c == "*" else codon[c] for c in pep)
        if rng.random() < 0.5:                                    # the other strand
            nt = nt[::-1].translate(str.maketrans("ACGT", "TGCA"))
        reads.append(np.frombuffer(nt.encode(), dtype=np.uint8))
    seqs, off = util.pack(reads)
    ix = oracle.load_fmi(fmi)
    want = oracle.classify(ix, None, oracle.params("mem", seg=0), seqs, off)
    assert (np.array([int(w["n_ids"]) for w in want]) >= 3).mean() > 0.3     # (ids, not matches: most reads have 3+ matches)
    emu = util.Emu()
    for env in ({}, {"KAIJU_GPU_FORCE_WIDE": "17"}, {"KAIJU_GPU_FORCE_WIDE": "17", "KAIJU_EMU_NO_ROW_TAX": "1"}, {"KAIJU_EMU_NO_TEXT": "1"},
                {"KAIJU_EMU_LOCATE_SERIAL": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        h = emu.load(fmi)
        got, nretry = emu.classify(h, util.gp("mem", seg=0), seqs, off)
        bad = [i for i in range(len(want)) if not util.same_hit(want[i], got[i])]
        assert not bad, (env, bad[:5])
        assert nretry > 0                                         # (reads with seventeen and eighteen matches)
        emu.lib.emu_index_free(h)
        for k in env:
            monkeypatch.delenv(k)
    # kaijux ids: the matches in the order maxMatches(.., 1) lists them
    xo = oracle.classify(ix, None, oracle.params("mem", seg=0, kaijux=1), seqs, off)
    import ctypes as C
    emu.lib.emu_index_load_x.restype = C.c_void_p
    emu.lib.emu_index_load_x.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    err = C.create_string_buffer(256)
    hx = emu.lib.emu_index_load_x(fmi.encode(), err, 256)
    assert hx, err.value
    gx, _ = emu.classify(hx, util.gp("mem", seg=0), seqs, off)
    bad = [i for i in range(len(xo)) if not util.same_hit(xo[i], gx[i])]
    assert not bad, bad[:5]


@pytest.mark.parametrize("mode,seg", CASES)
def test_fast_stage1_for_mates_up_to_287_nt(oracle, emu, golden, handles, mode, seg, monkeypatch):
    """build_fragments_fast<.., kS1UnitsLong>: mates of 192 .. 287 nucleotides (250-bp MiSeq reads, 2 x 250 pairs) take the fast
    stage 1 with six units per frame string and 128-bit masks - fragments of up to 95 residues, more than 24 of them per
    pair, SEG trigger windows beyond bit 63, k_trigcheck's scan in pieces.  Against the oracle, and the fragment lists
    against the general stage 1 (KAIJU_EMU_STAGE1_OLD)."""
    from kaiju_amd import synth
    h, ix, tax = handles
    rng = np.random.default_rng(23)
    _, dbseqs = util.read_fasta(os.path.join(util.GOLD, "db.faa"))
    prot = [p.decode() for p in dbseqs if len(p) >= 100]
    aa = "ACDEFGHIKLMNPQRSTVWY"
    codon = {"A": "GCT", "C": "TGT", "D": "GAT", "E": "GAA", "F": "TTT", "G": "GGT", "H": "CAT", "I": "ATT", "K": "AAA", "L": "CTT",
             "M": "ATG", "N": "AAT", "P": "CCT", "Q": "CAA", "R": "CGT", "S": "TCT", "T": "ACT", "V": "GTT", "W": "TGG", "Y": "TAT"}

    def one(L):
        if prot and rng.random() < 0.7:
            p = prot[int(rng.integers(0, len(prot)))]
            k = int(rng.integers(0, len(p) - 96))
            pep = "".join(c if c in codon else "A" for c in p[k:k + 96])
            if rng.random() < 0.3:                               # a low-complexity stretch: SEG trigger windows far into the string
                q = int(rng.integers(40, 80)); pep = pep[:q] + "S" * 14 + pep[q + 14:]
            nt = "".join(codon[c] for c in pep)
            nt = "ACGT"[int(rng.integers(0, 4))] * int(rng.integers(0, 3)) + nt
        else:
            nt = "".join(rng.choice(list("ACGT"), 300))
        nt = nt[:L]
        if rng.random() < 0.2:
            q = int(rng.integers(0, L)); nt = nt[:q] + "N" + nt[q + 1:]
        if rng.random() < 0.5:
            nt = nt[::-1].translate(str.maketrans("ACGTN", "TGCAN"))
        return np.frombuffer(nt.encode(), dtype=np.uint8)

    lens = [192, 193, 239, 240, 241, 250, 251, 286, 287]
    reads = [one(int(rng.choice(lens))) for _ in range(400)] + [one(150) for _ in range(20)]
    m1 = [one(int(rng.choice(lens))) for _ in range(200)]
    m2 = [one(int(rng.choice(lens + [100]))) for _ in range(200)]
    for seqs, off, pe in (util.pack(reads) + (False,), util.pack(m1, m2) + (True,)):
        oh = oracle.classify(ix, tax, oracle.params(mode, seg=seg, use_evalue=0), seqs, off, paired=pe)
        monkeypatch.delenv("KAIJU_EMU_STAGE1_OLD", raising=False)
        gh, _, frags_fast = emu.classify(h, util.gp(mode, seg=seg), seqs, off, paired=pe, want_frags=True)
        bad = [i for i in range(len(oh)) if not util.same_hit(oh[i], gh[i])]
        assert not bad, (pe, bad[:5], oh[bad[0]], gh[bad[0]])
        monkeypatch.setenv("KAIJU_EMU_STAGE1_OLD", "1")
        g2, _, frags_old = emu.classify(h, util.gp(mode, seg=seg), seqs, off, paired=pe, want_frags=True)
        monkeypatch.delenv("KAIJU_EMU_STAGE1_OLD")
        assert (g2 == gh).all()
        if not (mode == "mem" and seg):
            assert frags_fast == frags_old
        assert (gh["n_ids"] > 0).mean() > 0.3


@pytest.mark.parametrize("mode,lanes", [("mem", "v2"), ("mem", "v1"), ("mem", "wide16"), ("greedy", "v2"), ("greedy", "v1"), ("greedy", "wide16")])
def test_verbose_columns_under_the_id_cap(emu, golden, mode, lanes, monkeypatch):
    """kaiju -v when ids_from_SI's limit (more than 20 distinct taxon ids, ConsumerThread.cpp:805-807) ends the traversal before
    the last fragment: a family of 30 identical proteins under 30 taxa, reads whose fragments hold equally long matches in the
    family and elsewhere (tests/golden/idcap, written by make_golden_idcap.py with the reference binary).  The reference has
    pushed the peptide of EVERY such fragment by then (:580-590) - until round 6 the first-generation MEM lane stopped at the
    limit; matches behind it add neither ids nor accessions.  All of columns 4 - 7 == the reference's lines."""
    import ctypes as C
    import os
    d = os.path.join(golden.dir, "idcap")
    if lanes == "v1":
        monkeypatch.setenv("KAIJU_EMU_VERBOSE_V1", "1")
    if lanes.startswith("wide"):
        monkeypatch.setenv("KAIJU_GPU_FORCE_WIDE", lanes[4:6])
    h = emu.load(os.path.join(d, "db.fmi"))
    E = emu.lib
    E.emu_seq_name.restype = C.c_char_p
    E.emu_seq_name.argtypes = [C.c_void_p, C.c_uint32]
    E.emu_alphabet.restype = C.c_char_p
    E.emu_alphabet.argtypes = [C.c_void_p]
    alpha = E.emu_alphabet(h)
    names, reads = util.read_fastq(os.path.join(d, "reads.fq"))
    pnames, p1 = util.read_fastq(os.path.join(d, "pairs_1.fq"))
    _, p2 = util.read_fastq(os.path.join(d, "pairs_2.fq"))
    seqs, off = util.pack(reads)
    pseqs, poff = util.pack(p1, p2)
    ncap = 0
    for seg in (1, 0):
        for sq, of, nms, pe, tsv in ((seqs, off, names, False, f"ref_{mode}_{seg}.tsv"), (pseqs, poff, pnames, True, f"ref_{mode}_{seg}_pe.tsv")):
            n = len(nms)
            nacc, acc, tlen, text, cap = _verbose_buffers(E, n)
            try:
                gh, _ = emu.classify(h, util.gp(mode, seg=seg), sq, of, paired=pe)
            finally:
                E.emu_set_verbose(None, None, None, None, 0)
            lines = _tsv_lines(os.path.join(d, tsv))
            for r, nm in enumerate(nms):
                ref = lines[nm]
                if ref[0] != "C":
                    continue
                accs = set()
                for q in range(int(nacc[r])):
                    s = E.emu_seq_name(h, int(acc[r * 20 + q]))
                    if s and b"_" in s:
                        accs.add(s[: s.rindex(b"_")].decode())
                t = "".join("," if c == 255 else chr(alpha[c]) for c in text[r * cap: r * cap + int(tlen[r])])
                ids = "".join(f"{x}," for x in sorted(int(x) for x in gh[r]["taxid"][:gh[r]["n_ids"]]))
                assert int(ref[3]) == int(gh[r]["best"]) and ref[4] == ids and ref[5] == "".join(x + "," for x in sorted(accs)) and ref[6] == t, \
                    (mode, lanes, seg, pe, nm, ref[3:], ids, sorted(accs), t)
                ncap += int(gh[r]["n_ids"]) == 21
    assert ncap >= 8                                       # (the fixture does what it is for)
