"""GPU tests of the paths added last in round 1 - protein input (kaiju -p, kaijup: k_fragments_protein), the list
order of the reference's kaijux MEM search (k_mem_x, kParamXOrder) and the exact pass for fragments with more SEG regions
than a record of the SEG pass holds (exact_pass.hip).  They were developed against the host emulation of
the kernels (tests/test_kernel_emu.py runs the same checks there) after the round's GPU budget was spent, so this file
is named to run after every test that had already been run on the device."""
import os
import subprocess

import numpy as np
import pytest

import util
from kaiju_amd import build

pytestmark = pytest.mark.gpu

CASES = [("mem", 1), ("mem", 0), ("greedy", 1), ("greedy", 0)]


@pytest.fixture(scope="module")
def gidx(gpu_lib, golden):
    return gpu_lib.Index(golden.fmi)


@pytest.mark.parametrize("mode,seg", CASES)
def test_protein_parity(gpu_lib, golden, gidx, oracle, mode, seg):
    """protein reads through the C-ABI == the oracle record by record, and == the reference's `kaiju -p` lines"""
    api = gpu_lib
    clf = api.Classifier(gidx, api.default_params(mode, seg=seg, input_is_protein=1))
    hits = clf.classify(golden.prot_seqs, golden.prot_off)
    assert clf.stats().error_flags == 0
    assert not (hits["flags"] & 0xC0000000).any()
    ix, tax = oracle.load_fmi(golden.fmi), oracle.load_nodes(golden.nodes)
    oh = oracle.classify(ix, tax, oracle.params(mode, seg=seg, protein=1, use_evalue=0), golden.prot_seqs, golden.prot_off)
    bad = [i for i in range(len(oh)) if not util.same_hit(oh[i], hits[i])]
    assert not bad, (bad[:5], oh[bad[0]], hits[bad[0]])
    res = clf.finalize(api.Taxonomy(golden.nodes), hits, golden.prot_off, False)
    ref = golden.tsv(f"refp_{mode}_{seg}.tsv")
    for nm, h, r in zip(golden.prot_names, hits, res):
        got = ("C", int(r["taxon"]), int(r["best"]), tuple(sorted(int(x) for x in h["taxid"][:h["n_ids"]]))) if r["classified"] \
            else ("U", 0, None, ())
        assert got == ref[nm], (nm, got, ref[nm])
    # a pair of protein reads does not exist (kaiju.cpp:201)
    with pytest.raises(api.KaijuGpuError):
        clf.classify(golden.pseqs, golden.poff, paired=True)


def test_protein_long_and_batched(gpu_lib, golden, gidx, oracle):
    """proteins of several thousand residues (windows refill, retry scratch sized for whole-read fragments) and a batch of
    many copies: == oracle"""
    api = gpu_lib
    rng = np.random.default_rng(3)
    prots = [s for s in golden.prot_reads if len(s) > 100]
    reads = []
    for _ in range(40):
        parts = [prots[int(rng.integers(0, len(prots)))] for _ in range(int(rng.integers(3, 12)))]
        reads.append(b"X".join(parts) if rng.random() < 0.5 else b"".join(parts))
    reads += list(golden.prot_reads) * 5
    seqs, off = util.pack(reads)
    ix, tax = oracle.load_fmi(golden.fmi), oracle.load_nodes(golden.nodes)
    for mode in ("mem", "greedy"):
        clf = api.Classifier(gidx, api.default_params(mode, seg=1, input_is_protein=1))
        hits = clf.classify(seqs, off)
        oh = oracle.classify(ix, tax, oracle.params(mode, seg=1, protein=1, use_evalue=0), seqs, off)
        bad = [i for i in range(len(oh)) if not util.same_hit(oh[i], hits[i])]
        assert not bad, (mode, bad[:5])
        assert not (hits["flags"] & 0x80000000).any() and clf.stats().error_flags == 0


def cli(name):
    build.build_cli()
    return os.path.join(os.path.dirname(build.CLI), name)


@pytest.mark.parametrize("mode,seg", CASES)
def test_cli_kaiju_p(gpu_lib, golden, tmp_path, mode, seg):
    """`kaiju -p -v`: all seven columns == the reference's lines"""
    out = str(tmp_path / "p.tsv")
    cmd = [cli("kaiju"), "-p", "-t", golden.nodes, "-f", golden.fmi, "-i", golden.prot_fa, "-a", mode, "-v", "-o", out] + \
          ([] if seg else ["-X"])
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    assert open(out).read() == open(os.path.join(golden.dir, f"refp_{mode}_{seg}.tsv")).read()
    # without -v (LCA on the device, 16-byte records)
    cmd.remove("-v")
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    want = ["\t".join(line.rstrip("\n").split("\t")[:3]) for line in open(os.path.join(golden.dir, f"refp_{mode}_{seg}.tsv"))]
    assert [line.rstrip("\n") for line in open(out)] == want


@pytest.mark.parametrize("mode", ["mem", "greedy"])
def test_cli_kaijup(gpu_lib, golden, tmp_path, mode):
    """kaijup == the reference's kaijup lines (names kept whole, both kinds of U lines), with and without -v"""
    for v in (False, True):
        out = str(tmp_path / "px.tsv")
        cmd = [cli("kaijup"), "-f", golden.fmi, "-i", golden.prot_fa, "-a", mode, "-o", out] + (["-v"] if v else [])
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        assert open(out).read() == open(os.path.join(golden.dir, f"refpx_{mode}{'_v' if v else ''}.tsv")).read(), (mode, v)


def test_cli_kaijux_mem(gpu_lib, golden, tmp_path):
    """kaijux -a mem (the reference searches with maxMatches(.., 1), ConsumerThreadx.cpp:135): lines identical with and
    without -v, single and paired"""
    for pe in (False, True):
        for v in (False, True):
            out = str(tmp_path / "x.tsv")
            cmd = [cli("kaijux"), "-f", golden.fmi, "-a", "mem", "-o", out]
            cmd += ["-i", os.path.join(golden.dir, "pairs_1.fq"), "-j", os.path.join(golden.dir, "pairs_2.fq")] if pe else \
                   ["-i", os.path.join(golden.dir, "reads.fq")]
            if v:
                cmd.append("-v")
            subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
            ref = os.path.join(golden.dir, f"refx_mem{'_pe' if pe else ''}{'_v' if v else ''}.tsv")
            assert open(out).read() == open(ref).read(), (pe, v)


def test_kaijux_mem_order_under_the_id_cap(gpu_lib, oracle, tmp_path):
    """a database of near-identical sequences: every match interval holds more rows than the 21-id cap lets through, so the
    collected sequences depend on the order in which the matches of a fragment are visited; == the oracle's kaijux mode
    (maxMatches_limited), ids in traversal order"""
    api = gpu_lib
    from kaiju_amd import mkfmi
    faa, fmi = str(tmp_path / "rep.faa"), str(tmp_path / "rep.fmi")
    reads = util.repetitive_db(faa)
    mkfmi.build_fmi(faa, fmi, threads=2, exponent=3)
    idx = api.Index(fmi, id_mode=api.IDS_SEQUENCE)
    assert not idx.info.warnings
    seqs, off = util.pack(reads)
    ix = oracle.load_fmi(fmi)
    clf = api.Classifier(idx, api.default_params("mem", seg=0))
    hits = clf.classify(seqs, off)
    oh = oracle.classify(ix, None, oracle.params("mem", seg=0, kaijux=1), seqs, off)
    assert sum(1 for h in oh if h["flags"] & 1) > 20                     # the cap is hit
    bad = [i for i in range(len(oh)) if not util.same_hit(oh[i], hits[i])]
    assert not bad, bad[:5]


@pytest.mark.parametrize("mode", ["mem", "greedy"])
def test_exact_pass_many_seg_regions(gpu_lib, golden, gidx, oracle, mode):
    """a fragment with more low-complexity regions than a record of the SEG pass holds (15): the read is classified again
    by the exact pass and equals the oracle; protein and nucleotide reads, alone and mixed into a batch of ordinary reads"""
    api = gpu_lib
    prot, nuc = util.many_region_reads()
    ix, tax = oracle.load_fmi(golden.fmi), oracle.load_nodes(golden.nodes)
    for reads, protein in ((prot, 1), (nuc, 0), (list(golden.prot_reads) * 3 + prot, 1), (list(golden.reads) * 3 + nuc, 0)):
        seqs, off = util.pack(reads)
        clf = api.Classifier(gidx, api.default_params(mode, seg=1, input_is_protein=protein))
        hits = clf.classify(seqs, off)
        assert clf.stats().error_flags == 0 and not (hits["flags"] & 0xC0000000).any()
        oh = oracle.classify(ix, tax, oracle.params(mode, seg=1, protein=protein, use_evalue=0), seqs, off)
        bad = [i for i in range(len(oh)) if not util.same_hit(oh[i], hits[i])]
        assert not bad, (mode, protein, bad[:5])


@pytest.mark.parametrize("kind", ["prot", "nuc"])
@pytest.mark.parametrize("mode", ["mem", "greedy"])
def test_cli_exact_pass(gpu_lib, golden, tmp_path, mode, kind):
    """the command line on reads that go through the exact pass: all seven columns == the reference's lines, exit status 0"""
    out = str(tmp_path / "r.tsv")
    cmd = [cli("kaiju"), "-t", golden.nodes, "-f", golden.fmi, "-i", os.path.join(golden.dir, f"regions_{kind}.fa"), "-a", mode,
           "-v", "-o", out] + (["-p"] if kind == "prot" else [])
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    assert open(out).read() == open(os.path.join(golden.dir, f"refr_{kind}_{mode}.tsv")).read()
    cmd.remove("-v")
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    want = ["\t".join(line.rstrip("\n").split("\t")[:3]) for line in open(os.path.join(golden.dir, f"refr_{kind}_{mode}.tsv"))]
    assert [line.rstrip("\n") for line in open(out)] == want
