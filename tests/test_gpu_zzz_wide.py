"""Parity on an index that NEEDS the wide path: 2^32 rows and more, no KAIJU_GPU_FORCE_WIDE anywhere - the loader picks the
layout with 64-bit positions because of the size of the index (k_mem_wide2, k_mem_locate_wide, k_greedy2_wide).

Sorting 4.3 G suffixes takes seven minutes on 256 host threads (tests/tools/wide_index.py, the index the wide-path RATES are
measured on).  This test gets a true FM-index of that size in half a minute instead: a database of 3 700 proteins in which
every protein occurs 4 201 times (copy t under another taxon), written by kaiju_build_fmi_replicated - byte for byte the
file the sorter writes for the spelled-out FASTA (tests/test_mkfmi_pin.py pins that).  Every match interval is at least
4 201 rows wide there, so the id cap (21 distinct taxa) and the 64-bit interval arithmetic are at work in every read.
The oracle (the CPU restatement, which follows the reference's 64-bit IndexType) reads the same 4.4 GB .fmi.

Runs last (file name) and only skips when the box cannot hold the file: 24 GB of host memory, 6 GB under the temp dir."""
import os
import shutil
import sys
import time

import numpy as np
import pytest

import util
from kaiju_amd import mkfmi, synth

pytestmark = pytest.mark.gpu

COPIES = 4201


def _resources_ok(tmp):
    try:
        with open("/proc/meminfo") as f:
            avail = [int(l.split()[1]) for l in f if l.startswith("MemAvailable:")][0] * 1024
    except Exception:  # noqa: BLE001
        avail = 0
    free = shutil.disk_usage(tmp).free
    return avail >= 24 << 30 and free >= 6 << 30, f"MemAvailable {avail >> 30} GiB, free under {tmp} {free >> 30} GiB"


@pytest.fixture(scope="module")
def wide(tmp_path_factory, gpu_lib):
    tmp = str(tmp_path_factory.mktemp("wide"))
    ok, why = _resources_ok(tmp)
    if not ok:
        pytest.skip("not enough room for a 2^32-row index: " + why)
    lines, leaves = synth.make_taxonomy(6, 5, 5)
    db = synth.make_db(nseq=3701, seed=4242, leaves=leaves, max_len=900)
    faa, fmi, nodes = f"{tmp}/base.faa", f"{tmp}/wide.fmi", f"{tmp}/nodes.dmp"
    synth.write_fasta(db, faa)
    synth.write_nodes_dmp(nodes, lines)
    t0 = time.time()
    mkfmi.build_fmi_replicated(faa, fmi, COPIES, threads=0, exponent=3, copy_taxids=np.asarray(leaves, dtype=np.uint64))
    t_build = time.time() - t0
    api = gpu_lib
    t0 = time.time()
    index = api.Index(fmi)
    t_load = time.time() - t0
    print(f"[wide test] {db.total_aa} aa x {COPIES}: .fmi {os.path.getsize(fmi)/1e9:.2f} GB in {t_build:.0f}s, bwtlen {index.info.bwtlen} "
          f"= {index.info.bwtlen / 2**32:.3f} x 2^32, {index.footprint.total/1e9:.1f} GB in HBM, loaded in {t_load:.0f}s", file=sys.stderr)
    import pyoracle as po
    po.build_oracle()
    O = po.Oracle()
    oix, otax = O.load_fmi(fmi), O.load_nodes(nodes)
    yield {"api": api, "index": index, "db": db, "O": O, "oix": oix, "otax": otax, "tmp": tmp, "nodes": nodes, "fmi": fmi}
    del index
    shutil.rmtree(tmp, ignore_errors=True)


def test_the_loader_picked_the_wide_layout(wide):
    ix = wide["index"]
    assert ix.info.bwtlen >= 2 ** 32 and ix.info.warnings == 0
    fp = ix.footprint
    assert fp.wide == 1 and fp.count_bases > 0 and fp.sa_taxid == 0 and fp.kmer_lines == 0 and fp.kmer_k >= 5
    # 2 bytes of rank blocks + half a byte of sampled sequence numbers per row (e = 3), nothing else that grows with the index
    per_row = (fp.rank_blocks + fp.sa_seq + fp.count_bases) / ix.info.bwtlen
    assert 2.4 < per_row < 2.6, per_row


@pytest.mark.parametrize("mode,paired", [("mem", False), ("greedy", False), ("mem", True), ("greedy", True)])
def test_wide_index_parity(wide, mode, paired):
    """record by record (best length / score, taxon ids in the reference's traversal order, flags) against the oracle"""
    api, O, db = wide["api"], wide["O"], wide["db"]
    n = 6000 if mode == "mem" else 3000
    if paired:
        m1, m2 = synth.make_pairs(db, n // 2, seed=778)
        seqs, off = synth.pack_reads(m1, m2)
    else:
        seqs, off = synth.pack_reads(synth.make_reads(db, n, seed=777))
    clf = api.Classifier(wide["index"], api.default_params(mode, seg=1))
    hits = clf.classify(seqs, off, paired=paired)
    st = clf.stats()
    assert st.error_flags == 0
    oh = O.classify(wide["oix"], wide["otax"], O.params(mode, seg=1, use_evalue=0), seqs, off, paired=paired)
    bad = [i for i in range(len(oh)) if not util.same_hit(oh[i], hits[i])]
    assert not bad, (mode, paired, len(bad), bad[:5])
    assert (hits["n_ids"] > 0).mean() > 0.5
    assert (hits["n_ids"] > 5).mean() > 0.3          # (the copies carry different taxa: most reads collect many ids)
    clf.close()


def test_wide_index_command_line(wide, tmp_path):
    """the drop-in program on the same index against the oracle's C/U + taxon decisions"""
    import subprocess
    from kaiju_amd import build
    api, O, db = wide["api"], wide["O"], wide["db"]
    reads = synth.make_reads(db, 2000, seed=99)
    fq = str(tmp_path / "r.fq")
    with open(fq, "wb") as f:
        for i, r in enumerate(reads):
            f.write(b"@r%d\n" % i + r.tobytes() + b"\n+\n" + b"I" * len(r) + b"\n")
    out = str(tmp_path / "o.tsv")
    subprocess.run([build.build_cli(), "-t", wide["nodes"], "-f", wide["fmi"], "-i", fq, "-o", out, "-a", "mem"], check=True)
    seqs, off = synth.pack_reads(reads)
    oh = O.classify(wide["oix"], wide["otax"], O.params("mem", seg=1), seqs, off)
    got = [l.rstrip("\n").split("\t") for l in open(out)]
    assert len(got) == len(oh)
    for i, g in enumerate(got):
        assert g[0] == ("C" if oh[i]["classified"] else "U") and int(g[2]) == int(oh[i]["lca"]), (i, g)
