"""Parity of the HIP path (through the C-ABI) with the oracle and the reference's golden output.

Small and medium cases are compared record by record with the oracle; the BASELINE-sized case
(millions of reads on a viruses-scale index) is checked through size-independent properties
(batch-split invariance, permutation invariance, strand symmetry of the MEM length, planted
reads must hit their source) plus a record-by-record comparison of a random sample."""
import ctypes as C
import os

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu

CASES = [("mem", 1), ("mem", 0), ("greedy", 1), ("greedy", 0)]


@pytest.fixture(scope="module")
def gidx(gpu_lib, golden):
    return gpu_lib.Index(golden.fmi)


@pytest.fixture(scope="module")
def ohandles(oracle, golden):
    return oracle.load_fmi(golden.fmi), oracle.load_nodes(golden.nodes)


def finalize_records(api, clf, tax, hits, off, paired=False):
    res = clf.finalize(tax, hits, off, paired)
    out = []
    for h, r in zip(hits, res):
        if r["classified"]:
            out.append(("C", int(r["taxon"]), int(r["best"]), tuple(sorted(int(x) for x in h["taxid"][:h["n_ids"]]))))
        else:
            out.append(("U", 0, None, ()))
    return out


@pytest.mark.parametrize("mode,seg", CASES)
def test_golden_single(gpu_lib, golden, gidx, oracle, ohandles, mode, seg):
    api = gpu_lib
    clf = api.Classifier(gidx, api.default_params(mode, seg=seg))
    hits = clf.classify(golden.seqs, golden.off)
    st = clf.stats()
    assert st.error_flags == 0
    assert not (hits["flags"] & 0xC0000000).any()
    # against the oracle, record by record (ids in traversal order)
    ix, tax = ohandles
    oh = oracle.classify(ix, tax, oracle.params(mode, seg=seg, use_evalue=0), golden.seqs, golden.off)
    bad = [i for i in range(len(oh)) if not util.same_hit(oh[i], hits[i])]
    assert not bad, (bad[:5], oh[bad[0]], hits[bad[0]])
    # against the reference binary's output lines
    got = finalize_records(api, clf, api.Taxonomy(golden.nodes), hits, golden.off)
    ref = golden.tsv(f"ref_{mode}_{seg}.tsv")
    bad = [(n, g, ref[n]) for n, g in zip(golden.names, got) if g != ref[n]]
    assert not bad, bad[:3]


@pytest.mark.parametrize("mode,seg", CASES)
def test_golden_short_reads_fast_path(gpu_lib, golden, gidx, oracle, ohandles, mode, seg, monkeypatch):
    """a batch of short reads only: k_fragments_fast and, MEM, the lazy SEG flow (the whole golden set holds 1000-nt reads
    and takes the general stage 1); vs the oracle, the reference's lines and the general path (KAIJU_GPU_STAGE1=old)"""
    api = gpu_lib
    idx, seqs, off = golden.short()
    clf = api.Classifier(gidx, api.default_params(mode, seg=seg))
    hits = clf.classify(seqs, off)
    assert clf.stats().error_flags == 0
    ix, tax = ohandles
    oh = oracle.classify(ix, tax, oracle.params(mode, seg=seg, use_evalue=0), seqs, off)
    bad = [i for i in range(len(oh)) if not util.same_hit(oh[i], hits[i])]
    assert not bad, (bad[:5], oh[bad[0]], hits[bad[0]])
    got = finalize_records(api, clf, api.Taxonomy(golden.nodes), hits, off)
    ref = golden.tsv(f"ref_{mode}_{seg}.tsv")
    names = [golden.names[i] for i in idx]
    assert not [(n, g) for n, g in zip(names, got) if g != ref[n]]
    monkeypatch.setenv("KAIJU_GPU_STAGE1", "old")
    h2 = api.Classifier(gidx, api.default_params(mode, seg=seg)).classify(seqs, off)
    assert (h2 == hits).all()


@pytest.mark.parametrize("seg", [1, 0])
def test_mem_locate_by_the_search_lanes(gpu_lib, golden, gidx, oracle, ohandles, seg, monkeypatch):
    """KAIJU_GPU_MEM_LANE=v1: the first-generation MEM lanes, which walk to the ids themselves (the second-generation ones leave
    every read's longest matches to k_mem_locate*); short reads and pairs vs the oracle and vs the default flow"""
    api = gpu_lib
    ix, tax = ohandles
    _, sseqs, soff = golden.short()
    for seqs, off, pe in ((sseqs, soff, False), (golden.pseqs, golden.poff, True)):
        oh = oracle.classify(ix, tax, oracle.params("mem", seg=seg, use_evalue=0), seqs, off, paired=pe)
        monkeypatch.delenv("KAIJU_GPU_MEM_LANE", raising=False)
        hd = api.Classifier(gidx, api.default_params("mem", seg=seg)).classify(seqs, off, paired=pe)
        monkeypatch.setenv("KAIJU_GPU_MEM_LANE", "v1")
        hi = api.Classifier(gidx, api.default_params("mem", seg=seg)).classify(seqs, off, paired=pe)
        monkeypatch.delenv("KAIJU_GPU_MEM_LANE")
        for hits in (hd, hi):
            bad = [i for i in range(len(oh)) if not util.same_hit(oh[i], hits[i])]
            assert not bad, (seg, pe, bad[:5])
            assert not (hits["flags"] & 0x20000000).any()        # kHitLocPending never leaves the library


@pytest.mark.parametrize("mode,seg", CASES)
def test_device_lca_compact_records(gpu_lib, golden, gidx, mode, seg):
    """k_lca: hit records -> 16-byte records on the device; finalize_compact == finalize_hits == reference lines"""
    api = gpu_lib
    tax = api.Taxonomy(golden.nodes)
    dtax = api.DeviceTaxonomy(tax, 0)
    clf = api.Classifier(gidx, api.default_params(mode, seg=seg))
    for seqs, off, names, pe, tsv in ((golden.seqs, golden.off, golden.names, False, f"ref_{mode}_{seg}.tsv"),
                                      (golden.pseqs, golden.poff, golden.pnames, True, f"ref_{mode}_{seg}_pe.tsv")):
        hits = clf.classify(seqs, off, paired=pe)
        recs = clf.lca(dtax, hits)
        a = clf.finalize(tax, hits, off, pe)
        b = clf.finalize_compact(recs, off, pe)
        assert (a == b).all()
        assert (recs["best"] == hits["best"]).all() and ((recs["info"] & 255) == hits["n_ids"]).all()
        ref = golden.tsv(tsv)
        for n, r in zip(names, b):
            assert (ref[n][0] == "C") == bool(r["classified"]) and (not r["classified"] or int(r["taxon"]) == ref[n][1]), n


@pytest.mark.parametrize("shift,rowtax", [("16", "1"), ("20", "1"), ("31", "1"), ("20", "0")])
def test_wide_index_path(gpu_lib, golden, oracle, ohandles, shift, rowtax, monkeypatch):
    """indexes of 2^32 rows and more: 64-bit positions, rank counts relative to a base every 2^shift rows, 16-byte
    k-mer entries grown on the device - forced here on the small golden index (KAIJU_GPU_FORCE_WIDE); the ids through the
    row -> taxon table (k_mem_locate<true>, k_mem_locate_list<true>) and, without it, through the walks of k_mem_locate_wide"""
    api = gpu_lib
    monkeypatch.setenv("KAIJU_GPU_FORCE_WIDE", shift)
    monkeypatch.setenv("KAIJU_GPU_ROW_TAX", rowtax)
    monkeypatch.setenv("KAIJU_GPU_KMER", "6")
    idx = api.Index(golden.fmi)
    ix, tax = ohandles
    reads = util.long_reads(n=40)
    lseqs, loff = util.pack(reads)
    for mode in ("mem", "greedy"):
        for seqs, off, pe in ((golden.seqs, golden.off, False), (golden.pseqs, golden.poff, True), (lseqs, loff, False)):
            clf = api.Classifier(idx, api.default_params(mode, seg=1))
            hits = clf.classify(seqs, off, paired=pe)
            oh = oracle.classify(ix, tax, oracle.params(mode, seg=1, use_evalue=0), seqs, off, paired=pe)
            bad = [i for i in range(len(oh)) if not util.same_hit(oh[i], hits[i])]
            assert not bad, (shift, mode, pe, bad[:5])


@pytest.mark.gpu
def test_text_positions_of_an_index_with_64_bit_rows(gpu_lib, oracle, tmp_path, monkeypatch):
    """Wide layout with the database text and the text position of every 2^tv_shift-th row (k_seq_walk_len / k_seq_walk_fill at
    index load, K_SAPOS / K_TEXT in k_mem_wide2): every sample density - and no text at all - gives the oracle's records"""
    api = gpu_lib
    from kaiju_amd import mkfmi, synth
    _, leaves = synth.make_taxonomy(3, 3, 3)
    db = synth.make_db(nseq=6000, seed=21, leaves=leaves, max_len=900)
    faa, fmi = str(tmp_path / "db.faa"), str(tmp_path / "db.fmi")
    synth.write_fasta(db, faa)
    mkfmi.build_fmi(faa, fmi, threads=4, exponent=3)
    seqs, off = synth.pack_reads(synth.make_reads(db, 30000, seed=9))
    m1, m2 = synth.make_pairs(db, 5000, seed=10)
    pseqs, poff = synth.pack_reads(m1, m2)
    ix = oracle.load_fmi(fmi)
    want = {pe: oracle.classify(ix, None, oracle.params("mem", seg=1), s, o, paired=pe) for s, o, pe in ((seqs, off, False), (pseqs, poff, True))}
    monkeypatch.setenv("KAIJU_GPU_FORCE_WIDE", "18")
    # (tv "-1": no text, only the row -> taxon table - what a refseq_ref-class index gets; "1/walks": the text without that table)
    for tv in ("none", "auto", "0", "1", "3", "-1", "1/walks"):
        monkeypatch.delenv("KAIJU_GPU_NO_TEXT", raising=False)
        monkeypatch.delenv("KAIJU_GPU_TV_SHIFT", raising=False)
        monkeypatch.delenv("KAIJU_GPU_ROW_TAX", raising=False)
        if tv == "none":
            monkeypatch.setenv("KAIJU_GPU_NO_TEXT", "1")
        elif tv == "1/walks":
            monkeypatch.setenv("KAIJU_GPU_TV_SHIFT", "1")
            monkeypatch.setenv("KAIJU_GPU_ROW_TAX", "0")
        elif tv != "auto":
            monkeypatch.setenv("KAIJU_GPU_TV_SHIFT", tv)
        idx = api.Index(fmi)
        fp = idx.footprint
        assert fp.wide == 1 and (fp.text > 0) == (tv not in ("none", "-1")) and (fp.sa_full > 0) == (tv != "none")
        rows = int(idx.info.bwtlen)
        assert (fp.sa_full >= 4 * rows) == (tv not in ("none", "1/walks"))          # (the table: 4 B per row)
        clf = api.Classifier(idx, api.default_params("mem", seg=1))
        for s, o, pe in ((seqs, off, False), (pseqs, poff, True)):
            hits = clf.classify(s, o, paired=pe)
            assert clf.stats().error_flags == 0
            bad = [i for i in range(len(hits)) if not util.same_hit(want[pe][i], hits[i])]
            assert not bad, (tv, pe, bad[:5])


def test_index_image_loads_like_the_fmi(gpu_lib, golden, gidx, tmp_path):
    """an index loaded from its device image classifies exactly like the one packed from the .fmi"""
    api = gpu_lib
    img = str(tmp_path / "db.kjimg")
    api.write_index_image(golden.fmi, img)
    # the arrays that grow with the index are streamed from the file to the device in page-locked pieces: default piece size
    # (one piece here), pieces of 16 KB (dozens per array, several reader threads per piece), and the host-copy path
    for env in ({"KAIJU_GPU_STREAM_PIECE_MB": "256"}, {"KAIJU_GPU_STREAM_PIECE_KB": "16"}, {"KAIJU_GPU_IMAGE_HOST_COPY": "1"}, {}):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            idx2 = api.Index(img)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        assert idx2.info.bwtlen == gidx.info.bwtlen and idx2.info.nseq == gidx.info.nseq
        assert idx2.footprint.as_dict() == gidx.footprint.as_dict()
        for mode in ("mem", "greedy"):
            a = api.Classifier(gidx, api.default_params(mode, seg=1)).classify(golden.seqs, golden.off)
            b = api.Classifier(idx2, api.default_params(mode, seg=1)).classify(golden.seqs, golden.off)
            assert (a == b).all(), env
        idx2.close()
    # a damaged image is refused
    data = open(img, "rb").read()
    bad = str(tmp_path / "bad.kjimg")
    open(bad, "wb").write(data[: len(data) // 2])
    with pytest.raises(Exception):
        api.Index(bad)


@pytest.mark.parametrize("mode,seg,lane", [(m, s, "default") for m, s in CASES] + [(m, s, "v1") for m, s in CASES] + [("greedy", 1, "v1pool64")])
def test_verbose_columns(gpu_lib, golden, gidx, mode, seg, lane, monkeypatch):
    """kaiju -v columns 6 (accessions) and 7 (matched peptides) == the reference's lines, single and paired: from the
    second-generation lanes (k_mem_vb / k_greedy2_vb + k_mem_verbose; default) and from the first-generation lanes
    (KAIJU_GPU_VERBOSE_LANE=v1, read when the context is created), which the retry and exact passes still run"""
    api = gpu_lib
    if lane.startswith("v1"):
        monkeypatch.setenv("KAIJU_GPU_VERBOSE_LANE", "v1")
    if lane == "v1pool64":               # (-v gives the first-generation Greedy main pass 512 queue slots per lane: 64 sends reads to the retry pass)
        monkeypatch.setenv("KAIJU_GPU_G1_POOL", "64")
    clf = api.Classifier(gidx, api.default_params(mode, seg=seg))
    tax = api.Taxonomy(golden.nodes)
    for seqs, off, names, pe, tsv in ((golden.seqs, golden.off, golden.names, False, f"ref_{mode}_{seg}.tsv"),
                                      (golden.pseqs, golden.poff, golden.pnames, True, f"ref_{mode}_{seg}_pe.tsv")):
        hits, accs, peps = clf.classify_verbose(seqs, off, paired=pe)
        plain = clf.classify(seqs, off, paired=pe)
        assert all(util.same_hit(a, b) for a, b in zip(plain, hits))          # the verbose pass == the plain one
        res = clf.finalize(tax, hits, off, pe)
        lines = {}
        with open(os.path.join(golden.dir, tsv)) as f:
            for line in f:
                p = line.rstrip("\n").split("\t")
                lines[p[1]] = p
        for n, r, a, t in zip(names, res, accs, peps):
            ref = lines[n]
            if r["classified"]:
                assert ref[0] == "C" and ref[5] == "".join(x + "," for x in a) and ref[6] == t, (n, ref[5:], a, t)
            else:
                assert ref[0] == "U"


@pytest.mark.parametrize("mode", ["mem", "greedy"])
def test_long_reads(gpu_lib, gidx, oracle, ohandles, mode):
    """reads of 400..3000 nt (alone and mixed with short ones): in-place stage 1, window refills, spills"""
    api = gpu_lib
    ix, tax = ohandles
    reads = util.long_reads()
    for batch in (reads, reads[:40] + [b"ACGT" * 40, b"", b"ACG"] + reads[40:80]):
        seqs, off = util.pack(batch)
        for seg in (1, 0):
            clf = api.Classifier(gidx, api.default_params(mode, seg=seg))
            hits = clf.classify(seqs, off)
            assert clf.stats().error_flags == 0
            oh = oracle.classify(ix, tax, oracle.params(mode, seg=seg, use_evalue=0), seqs, off)
            bad = [i for i in range(len(oh)) if not util.same_hit(oh[i], hits[i])]
            assert not bad, (mode, seg, bad[:5])


@pytest.mark.parametrize("k", ["0", "3", "6", "7"])
def test_kmer_table_depths(gpu_lib, golden, oracle, ohandles, k, monkeypatch):
    """the k-mer table that starts every backward search: none, host-built, grown on the device to 6 and 7
    letters (the depth a viruses-size index gets); records identical in all cases"""
    api = gpu_lib
    monkeypatch.setenv("KAIJU_GPU_KMER", k)
    idx = api.Index(golden.fmi)
    ix, tax = ohandles
    for mode, seg in (("mem", 1), ("greedy", 1)):
        clf = api.Classifier(idx, api.default_params(mode, seg=seg))
        hits = clf.classify(golden.seqs, golden.off)
        oh = oracle.classify(ix, tax, oracle.params(mode, seg=seg, use_evalue=0), golden.seqs, golden.off)
        bad = [i for i in range(len(oh)) if not util.same_hit(oh[i], hits[i])]
        assert not bad, (k, mode, bad[:5])
        hits = clf.classify(golden.pseqs, golden.poff, paired=True)
        oh = oracle.classify(ix, tax, oracle.params(mode, seg=seg, use_evalue=0), golden.pseqs, golden.poff, paired=True)
        bad = [i for i in range(len(oh)) if not util.same_hit(oh[i], hits[i])]
        assert not bad, (k, mode, "paired", bad[:5])


@pytest.mark.parametrize("mode,seg", CASES)
def test_golden_paired(gpu_lib, golden, gidx, mode, seg):
    api = gpu_lib
    clf = api.Classifier(gidx, api.default_params(mode, seg=seg))
    hits = clf.classify(golden.pseqs, golden.poff, paired=True)
    got = finalize_records(api, clf, api.Taxonomy(golden.nodes), hits, golden.poff, paired=True)
    ref = golden.tsv(f"ref_{mode}_{seg}_pe.tsv")
    bad = [(n, g, ref[n]) for n, g in zip(golden.pnames, got) if g != ref[n]]
    assert not bad, bad[:3]


def test_parameter_variants(gpu_lib, golden, gidx):
    api = gpu_lib
    tax = api.Taxonomy(golden.nodes)
    for name, p in (("ref_greedy_e5_s50.tsv", api.default_params("greedy", mismatches=5, min_score=50, min_evalue=10.0)),
                    ("ref_greedy_e0.tsv", api.default_params("greedy", mismatches=0)),
                    ("ref_mem_m15.tsv", api.default_params("mem", min_fragment_length=15))):
        clf = api.Classifier(gidx, p)
        hits = clf.classify(golden.seqs, golden.off)
        got = finalize_records(api, clf, tax, hits, golden.off)
        ref = golden.tsv(name)
        bad = [(n, g, ref[n]) for n, g in zip(golden.names, got) if g != ref[n]]
        assert not bad, (name, bad[:3])


def test_empty_and_tiny_batches(gpu_lib, golden, gidx):
    api = gpu_lib
    clf = api.Classifier(gidx, api.default_params("mem"))
    h = clf.classify(np.zeros(0, dtype=np.uint8), np.zeros(1, dtype=np.uint64))
    assert len(h) == 0
    s, o = util.pack([b"", b"ACGT", golden.reads[0]])
    h = clf.classify(s, o)
    assert h["best"][0] == 0 and h["best"][1] == 0 and h["n_ids"][0] == 0
    full = clf.classify(golden.seqs, golden.off)
    assert util.same_hit(full[0], h[2])


def test_unsupported_parameters(gpu_lib, gidx):
    api = gpu_lib
    with pytest.raises(api.KaijuGpuError):
        api.Classifier(gidx, api.default_params("greedy", mismatches=9))
    with pytest.raises(api.KaijuGpuError):
        api.Classifier(gidx, api.default_params("mem", max_match_ids=40))


# ----------------------------------------------------------------------------------------
# viruses-scale workload (BASELINE.json configs 1-3)
# ----------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def big(tmp_path_factory, gpu_lib):
    """viruses-like index + reads; size via KAIJU_TEST_NSEQ / KAIJU_TEST_READS"""
    from kaiju_amd import synth, mkfmi
    W = str(tmp_path_factory.mktemp("big"))
    nseq = int(os.environ.get("KAIJU_TEST_NSEQ", "680001"))
    nreads = int(os.environ.get("KAIJU_TEST_READS", "2000000"))
    lines, leaves = synth.make_taxonomy()
    synth.write_nodes_dmp(f"{W}/nodes.dmp", lines)
    db = synth.make_db(nseq=nseq, seed=12345, leaves=leaves)
    synth.write_fasta(db, f"{W}/db.faa")
    mkfmi.build_fmi(f"{W}/db.faa", f"{W}/db.fmi", threads=os.cpu_count() or 8, exponent=3)
    reads = synth.make_reads(db, nreads, seed=777)
    seqs, off = synth.pack_reads(reads)
    return dict(W=W, db=db, reads=reads, seqs=seqs, off=off, index=gpu_lib.Index(f"{W}/db.fmi"))


@pytest.mark.parametrize("mode,seg", [("mem", 1), ("greedy", 1)])
def test_fullsize_sample_vs_oracle(gpu_lib, oracle, big, mode, seg):
    api = gpu_lib
    clf = api.Classifier(big["index"], api.default_params(mode, seg=seg))
    hits = clf.classify(big["seqs"], big["off"])
    assert clf.stats().error_flags == 0
    assert not (hits["flags"] & 0xC0000000).any()
    rng = np.random.default_rng(11)
    sample = np.sort(rng.choice(len(hits), size=min(20000, len(hits)), replace=False))
    from kaiju_amd import synth
    s2, o2 = synth.pack_reads(big["reads"][sample])
    ix = oracle.load_fmi(f"{big['W']}/db.fmi")
    oh = oracle.classify(ix, None, oracle.params(mode, seg=seg, use_evalue=0), s2, o2)
    bad = [int(sample[i]) for i in range(len(sample)) if not util.same_hit(oh[i], hits[sample[i]])]
    assert not bad, bad[:5]
    frac = float((hits["n_ids"] > 0).mean())
    assert 0.55 < frac < 0.80          # 70 % of the reads come from the database


@pytest.mark.parametrize("mode", ["mem", "greedy"])
def test_fullsize_paired_sample_vs_oracle(gpu_lib, oracle, big, mode):
    """2 x 150-bp pairs at scale (the shape of BASELINE configs[3]): a batch of 500 k pairs, 10 k of them record by record"""
    api = gpu_lib
    from kaiju_amd import synth
    m1, m2 = synth.make_pairs(big["db"], 500000, seed=778)
    seqs, off = synth.pack_reads(m1, m2)
    clf = api.Classifier(big["index"], api.default_params(mode, seg=1))
    hits = clf.classify(seqs, off, paired=True)
    assert clf.stats().error_flags == 0
    assert not (hits["flags"] & 0xC0000000).any()
    sample = np.sort(np.random.default_rng(12).choice(len(hits), size=10000, replace=False))
    s2, o2 = synth.pack_reads(m1[sample], m2[sample])
    ix = oracle.load_fmi(f"{big['W']}/db.fmi")
    oh = oracle.classify(ix, None, oracle.params(mode, seg=1, use_evalue=0), s2, o2, paired=True)
    bad = [int(sample[i]) for i in range(len(sample)) if not util.same_hit(oh[i], hits[sample[i]])]
    assert not bad, bad[:5]
    # the batch split in two gives the same records
    h1 = clf.classify(*synth.pack_reads(m1[:250000], m2[:250000]), paired=True)
    assert (h1 == hits[:250000]).all()


def test_fullsize_invariants(gpu_lib, big):
    api = gpu_lib
    clf = api.Classifier(big["index"], api.default_params("mem", seg=1))
    reads = big["reads"]
    from kaiju_amd import synth
    hits = clf.classify(big["seqs"], big["off"])
    n = len(reads)
    # (1) batch-split invariance: two halves give the same records as the whole
    h1 = clf.classify(*synth.pack_reads(reads[: n // 2]))
    h2 = clf.classify(*synth.pack_reads(reads[n // 2:]))
    assert (np.concatenate([h1, h2]) == hits).all()
    # (2) permutation invariance
    perm = np.random.default_rng(5).permutation(n)
    hp = clf.classify(*synth.pack_reads(reads[perm]))
    assert (hp == hits[perm]).all()
    # (3) strand symmetry: the reverse complement has the same set of peptides, hence the same
    #     longest match (ids may differ only in the rare 21-id cap case)
    comp = np.zeros(256, dtype=np.uint8)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    m = min(n, 500000)
    rc = comp[reads[:m, ::-1]]
    hr = clf.classify(*synth.pack_reads(rc))
    assert (hr["best"] == hits["best"][:m]).all()
    nocap = ((hr["flags"] | hits["flags"][:m]) & 1) == 0
    a = np.sort(hr["taxid"][nocap], axis=1)
    b = np.sort(hits["taxid"][:m][nocap], axis=1)
    assert (a == b).all()
    # (4) idempotence: same input, same output
    assert (clf.classify(big["seqs"], big["off"]) == hits).all()


def test_planted_reads_hit_their_source(gpu_lib, big):
    """exact back-translations of database windows must give a 50-aa match containing the source taxon"""
    api = gpu_lib
    from kaiju_amd import synth
    db = big["db"]
    rng = np.random.default_rng(3)
    lens = np.diff(db.offsets)
    elig = np.nonzero(lens >= 60)[0]
    seq = elig[rng.integers(0, len(elig), size=20000)]
    start = db.offsets[seq] + (rng.random(len(seq)) * (lens[seq] - 50)).astype(np.int64)
    win = db.codes[start[:, None] + np.arange(50)[None, :]]
    cod = synth._CODTAB[win, 0]
    nts = np.stack([(cod >> 4) & 3, (cod >> 2) & 3, cod & 3], axis=2).reshape(len(seq), 150)
    reads = np.frombuffer(b"ACGT", dtype=np.uint8)[nts]
    clf = api.Classifier(big["index"], api.default_params("mem", seg=0))
    hits = clf.classify(*synth.pack_reads(reads))
    assert (hits["best"] == 50).all()
    src = db.taxids[seq].astype(np.uint64)
    found = (hits["taxid"] == src[:, None]).any(axis=1) | ((hits["flags"] & 1) == 1)
    assert found.all()


def test_randomised_databases_and_parameters(gpu_lib):
    """a few rounds of tests/tools/fuzz_gpu.py through the C-ABI (random databases, reads, parameters)"""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(util.ROOT, "tests", "tools", "fuzz_gpu.py"), "12", "5"],
                       capture_output=True, timeout=600)
    assert r.returncode == 0 and b"FUZZ_OK" in r.stdout, r.stdout[-2000:].decode() + r.stderr[-2000:].decode()


class Hip:
    """the few HIP runtime calls the device-buffer test needs, through ctypes on the runtime libkaiju_gpu.so is linked
    against (torch is not used: a second HIP runtime in the test process would not see the GPU)"""

    def __init__(self):
        self.L = C.CDLL("libamdhip64.so.7")

    def ck(self, rc):
        assert rc == 0, f"HIP error {rc}"

    def malloc(self, n):
        p = C.c_void_p()
        self.ck(self.L.hipMalloc(C.byref(p), C.c_size_t(n)))
        return p.value

    def h2d(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        self.ck(self.L.hipMemcpy(C.c_void_p(dptr), C.c_void_p(arr.ctypes.data), C.c_size_t(arr.nbytes), 1))

    def d2h(self, dptr, nbytes):
        out = np.zeros(nbytes, dtype=np.uint8)
        self.ck(self.L.hipMemcpy(C.c_void_p(out.ctypes.data), C.c_void_p(dptr), C.c_size_t(nbytes), 2))
        return out

    def memset(self, dptr, nbytes):
        self.ck(self.L.hipMemset(C.c_void_p(dptr), 0, C.c_size_t(nbytes)))

    def stream(self):
        s = C.c_void_p()
        self.ck(self.L.hipStreamCreateWithFlags(C.byref(s), 1))       # hipStreamNonBlocking
        return s.value

    def free(self, dptr):
        self.L.hipFree(C.c_void_p(dptr))


@pytest.mark.parametrize("mode", ["mem", "greedy"])
def test_device_resident_entry_points(gpu_lib, golden, gidx, mode):
    """kaiju_gpu_classify_batch_device + kaiju_gpu_lca_batch_device on device buffers (what bench.py times): with the NULL
    stream (= the context's own stream for BOTH calls: the LCA must queue behind the search) and with an explicit HIP
    stream; records == the host-buffer entry points'.  Also kaiju_gpu_classify_batch_compact and the counting lanes."""
    api = gpu_lib
    hip = Hip()
    tax = api.Taxonomy(golden.nodes)
    dtax = api.DeviceTaxonomy(tax, 0)
    clf = api.Classifier(gidx, api.default_params(mode, seg=1))
    want_hits = clf.classify(golden.seqs, golden.off)
    want = clf.lca(dtax, want_hits)
    n = len(want_hits)
    seqs = np.ascontiguousarray(golden.seqs, dtype=np.uint8)
    off = np.ascontiguousarray(golden.off, dtype=np.uint64)
    d_seqs, d_off = hip.malloc(seqs.nbytes + 64), hip.malloc(off.nbytes)
    d_hits, d_rec = hip.malloc(n * 184), hip.malloc(n * 16)
    hip.h2d(d_seqs, seqs)
    hip.h2d(d_off, off)
    clf.set_max_read_length(int((off[1:] - off[:-1]).max()))
    assert clf.stream_handle() != 0
    for stream in (0, hip.stream()):
        for counting in (False, True):
            hip.memset(d_hits, n * 184)
            hip.memset(d_rec, n * 16)
            clf.count_ops(counting)
            clf.classify_device(d_seqs, seqs.nbytes, d_off, n, d_hits, stream=stream)
            clf.lca_device(dtax, d_hits, n, d_rec, stream=stream)
            clf.synchronize()
            clf.count_ops(False)
            hits = np.frombuffer(hip.d2h(d_hits, n * 184).tobytes(), dtype=api.HIT_DTYPE)
            rec = np.frombuffer(hip.d2h(d_rec, n * 16).tobytes(), dtype=api.COMPACT_DTYPE)
            assert all(util.same_hit(a, b) for a, b in zip(want_hits, hits)), (stream != 0, counting)
            assert (rec == want).all(), (stream != 0, counting)
            if counting:
                oc = clf.op_counts()
                assert oc["hits"] == n and oc["read_meta"] == n and oc["lane_iterations"] > oc["wave_iterations"] > 0
                assert oc["update_si"] <= oc["update_si_lines"] <= 2 * (oc["update_si"] + oc["multi_letter_steps"])
    got = clf.classify_compact(dtax, seqs, off)
    assert (got == want).all()
    # kaiju_gpu_classify_batch_device_compact: the two calls in one (MEM: the fused post-search pass writes the 16-byte records)
    for stream in (0, hip.stream()):
        hip.memset(d_hits, n * 184)
        hip.memset(d_rec, n * 16)
        clf.classify_device_compact(dtax, d_seqs, seqs.nbytes, d_off, n, d_hits, d_rec, stream=stream)
        clf.synchronize()
        hits = np.frombuffer(hip.d2h(d_hits, n * 184).tobytes(), dtype=api.HIT_DTYPE)
        rec = np.frombuffer(hip.d2h(d_rec, n * 16).tobytes(), dtype=api.COMPACT_DTYPE)
        assert all(util.same_hit(a, b) for a, b in zip(want_hits, hits)), stream != 0
        assert (rec == want).all(), stream != 0
    for p in (d_seqs, d_off, d_hits, d_rec):
        hip.free(p)


@pytest.mark.parametrize("seg", [1, 0])
def test_fused_post_search_pass(gpu_lib, big, seg, monkeypatch):
    """k_mem_post1 / _post2 (the lazy-SEG look, the locate and the LCA in one pass over the records) write what k_trigcheck,
    k_mem_locate, k_mem_locate_list and k_lca wrote as separate passes (KAIJU_GPU_FUSED_POST=0): hit records and 16-byte records
    of 1 M benchmark reads, single and paired"""
    api = gpu_lib
    from kaiju_amd import synth
    tax = api.Taxonomy(f"{big['W']}/nodes.dmp")
    dtax = api.DeviceTaxonomy(tax, 0)
    n = min(1000000, len(big["reads"]))
    for paired in (False, True):
        if paired:
            s2, o2 = synth.pack_reads(big["reads"][:n // 2], big["reads"][n // 2:n])
        else:
            s2, o2 = synth.pack_reads(big["reads"][:n])
        out = {}
        for fused in ("0", "1"):
            monkeypatch.setenv("KAIJU_GPU_FUSED_POST", fused)
            clf = api.Classifier(big["index"], api.default_params("mem", seg=seg))
            hits = clf.classify(s2, o2, paired=paired).copy()
            assert clf.stats().error_flags == 0
            out[fused] = (hits, clf.classify_compact(dtax, s2, o2, paired=paired).copy(), clf.lca(dtax, hits).copy())
        for f in ("best", "n_ids", "flags", "reserved", "taxid"):
            assert (out["0"][0][f] == out["1"][0][f]).all(), (paired, f)
        assert (out["0"][1] == out["1"][1]).all() and (out["1"][1] == out["1"][2]).all(), paired


def test_text_arrays_on_an_index_with_the_short_sample_array(gpu_lib, tmp_path, monkeypatch):
    """an index with the reference's short suffix-array sample (nseq % 8 == 0, KAIJU_IDX_WARN_SA_SHORT) gets its text arrays
    too (the rows behind the missing sample are resolved through the next one) - and the SAME records as without them: a row
    behind the missing sample gives no id (the reference reads out of bounds there), also when the match that ends there was
    grown along the text (DevIndex::beyond_lo), so a result does not depend on whether the arrays found room in HBM.  Forced
    wide: such an index gets no text arrays at all."""
    wide = bool(os.environ.get("KAIJU_GPU_FORCE_WIDE"))
    if os.environ.get("KAIJU_GPU_NO_TEXT"):
        pytest.skip("the suite runs without text arrays altogether (tests/tools/forced_wide_suite.sh): nothing to compare")
    from kaiju_amd import mkfmi, synth
    api = gpu_lib
    _, leaves = synth.make_taxonomy(3, 3, 3)
    db = synth.make_db(nseq=1600, seed=11, leaves=leaves, max_len=700)
    faa, fmi = str(tmp_path / "db.faa"), str(tmp_path / "db.fmi")
    synth.write_fasta(db, faa)
    mkfmi.build_fmi(faa, fmi, threads=2, exponent=3)
    seqs, off = synth.pack_reads(synth.make_reads(db, 20000, seed=5))
    m1, m2 = synth.make_pairs(db, 6000, seed=6)
    pseqs, poff = synth.pack_reads(m1, m2)
    with_text = api.Index(fmi)
    assert with_text.info.warnings & 1
    assert (with_text.footprint.text == 0) if wide else (with_text.footprint.text > 0 and with_text.footprint.sa_full > 0)
    monkeypatch.setenv("KAIJU_GPU_NO_TEXT", "1")
    without = api.Index(fmi)
    assert without.footprint.text == 0
    for mode in ("mem", "greedy"):
        for s, o, pe in ((seqs, off, False), (pseqs, poff, True)):
            a = api.Classifier(with_text, api.default_params(mode, seg=1)).classify(s, o, paired=pe)
            b = api.Classifier(without, api.default_params(mode, seg=1)).classify(s, o, paired=pe)
            assert (a == b).all(), (mode, pe, np.nonzero(a != b)[0][:5])
            assert (a["n_ids"] > 0).mean() > 0.4




@pytest.mark.parametrize("split", ["1", "0"])
def test_greedy_row_pool_lane(gpu_lib, golden, big, split, tmp_path):
    """the experimental row-pool Greedy lane (kj_greedy3.h: the reads of a block as rows of LDS, wavefronts pull rows by class; a
    variant build, -DKJ_GREEDY3 + KAIJU_GPU_GREEDY_LANE=v3 - DESIGN.md 6b) writes the records greedy_lane2 writes: the golden
    reads with and without SEG, 400 k benchmark reads.  The variant library is loaded by a process of its own."""
    import subprocess
    import sys
    api = gpu_lib
    lib = os.path.join(util.ROOT, "kaiju_amd", "variants", "libkaiju_gpu_g3.so")
    deps = [os.path.join(util.ROOT, "kaiju_amd", "csrc", f) for f in ("capi.hip", "kj_core.h", "kj_greedy3.h")]
    if not os.path.exists(lib) or any(os.path.getmtime(d) > os.path.getmtime(lib) for d in deps):
        subprocess.run(["bash", os.path.join(util.ROOT, "tests", "tools", "mem_variants.sh"), "build"], check=True,
                       env=dict(os.environ, VARIANTS="g3"))
    from kaiju_amd import synth
    n = min(400000, len(big["reads"]))
    s2, o2 = synth.pack_reads(big["reads"][:n])
    np.save(tmp_path / "seqs.npy", s2)
    np.save(tmp_path / "off.npy", o2)
    want = {}
    for seg in (1, 0):
        want[("golden", seg)] = api.Classifier(api.Index(golden.fmi), api.default_params("greedy", seg=seg)).classify(golden.seqs, golden.off).copy()
    want[("big", 1)] = api.Classifier(big["index"], api.default_params("greedy", seg=1)).classify(s2, o2).copy()
    code = f"""
import sys, numpy as np
sys.path.insert(0, {util.ROOT!r}); sys.path.insert(0, {os.path.join(util.ROOT, 'tests')!r})
import util
from kaiju_amd import api
g = util.Golden()
for seg in (1, 0):
    clf = api.Classifier(api.Index(g.fmi), api.default_params("greedy", seg=seg))
    np.save({str(tmp_path)!r} + f"/golden_{{seg}}.npy", clf.classify(g.seqs, g.off))
    assert clf.stats().error_flags == 0
clf = api.Classifier(api.Index({big['W']!r} + "/db.fmi"), api.default_params("greedy", seg=1))
np.save({str(tmp_path)!r} + "/big_1.npy", clf.classify(np.load({str(tmp_path)!r} + "/seqs.npy"), np.load({str(tmp_path)!r} + "/off.npy")))
assert clf.stats().error_flags == 0
"""
    subprocess.run([sys.executable, "-c", code], check=True,
                   env=dict(os.environ, KAIJU_GPU_LIB=lib, KAIJU_GPU_GREEDY_LANE="v3", KAIJU_GPU_G3_SPLIT=split))
    for (name, seg), w in want.items():
        got = np.load(tmp_path / f"{name}_{seg}.npy")
        for f in ("best", "n_ids", "flags", "taxid"):
            assert (got[f] == w[f]).all(), (name, seg, f)


@pytest.mark.parametrize("mode,lane", [("mem", "default"), ("mem", "v1"), ("greedy", "default"), ("greedy", "v1")])
def test_verbose_columns_under_the_id_cap(gpu_lib, golden, mode, lane, monkeypatch):
    """kaiju -v when ids_from_SI's limit (more than 20 distinct taxon ids) ends the traversal before the last fragment
    (tests/golden/idcap: a family of 30 identical proteins under 30 taxa; make_golden_idcap.py): columns 4 - 7 == the
    reference's lines - the peptide of EVERY fragment that holds a longest match (ConsumerThread.cpp:580-590), no ids or
    accessions from the matches behind the limit (:805-807)"""
    api = gpu_lib
    d = os.path.join(golden.dir, "idcap")
    if lane == "v1":
        monkeypatch.setenv("KAIJU_GPU_VERBOSE_LANE", "v1")
    idx = api.Index(os.path.join(d, "db.fmi"))
    names, reads = util.read_fastq(os.path.join(d, "reads.fq"))
    pnames, p1 = util.read_fastq(os.path.join(d, "pairs_1.fq"))
    _, p2 = util.read_fastq(os.path.join(d, "pairs_2.fq"))
    ncap = 0
    try:
        for seg in (1, 0):
            clf = api.Classifier(idx, api.default_params(mode, seg=seg))
            for (seqs, off), nms, pe, tsv in ((util.pack(reads), names, False, f"ref_{mode}_{seg}.tsv"),
                                              (util.pack(p1, p2), pnames, True, f"ref_{mode}_{seg}_pe.tsv")):
                hits, accs, peps = clf.classify_verbose(seqs, off, paired=pe)
                lines = {}
                with open(os.path.join(d, tsv)) as f:
                    for line in f:
                        p = line.rstrip("\n").split("\t")
                        lines[p[1]] = p
                for nm, h, a, t in zip(nms, hits, accs, peps):
                    ref = lines[nm]
                    if ref[0] != "C":
                        continue
                    ids = "".join(f"{x}," for x in sorted(int(x) for x in h["taxid"][:h["n_ids"]]))
                    assert int(ref[3]) == int(h["best"]) and ref[4] == ids and ref[5] == "".join(x + "," for x in a) and ref[6] == t, \
                        (mode, lane, seg, pe, nm, ref[3:], ids, a, t)
                    ncap += int(h["n_ids"]) == 21
            clf.close()
    finally:
        idx.close()
    assert ncap >= 8


@pytest.mark.parametrize("mode", ["mem", "greedy"])
def test_verbose_packed_equals_strided(gpu_lib, golden, gidx, mode):
    """kaiju_gpu_classify_batch_verbose_packed (column 7 of the batch as one string: the command line programs) == the entry point
    with a row of text_stride bytes per read, record for record and byte for byte; an empty batch is an empty string"""
    api = gpu_lib
    clf = api.Classifier(gidx, api.default_params(mode, seg=1))
    for seqs, off, pe in ((golden.seqs, golden.off, False), (golden.pseqs, golden.poff, True)):
        hits, v, text, stride = clf.classify_verbose_raw(seqs, off, paired=pe)
        hits, v = hits.copy(), v.copy()
        rows = [bytes(text[r * stride: r * stride + int(v[r]["text_len"])]) for r in range(len(v))]
        h2, v2, pos, s = clf.classify_verbose_packed(seqs, off, paired=pe)
        assert (hits == h2).all() and (v == v2).all()
        assert [s[int(pos[r]): int(pos[r]) + int(v2[r]["text_len"])] for r in range(len(v2))] == rows
        assert sum(len(x) for x in rows) == len(s) and any(rows)
    h0, v0, p0, s0 = clf.classify_verbose_packed(np.zeros(0, dtype=np.uint8), np.zeros(1, dtype=np.uint64))
    assert len(h0) == 0 and s0 == b""
    clf.close()
