"""kaiju_build_fmi (kaiju_amd/csrc/mkfmi.cpp) writes byte for byte the file that the reference's kaiju-mkbwt +
kaiju-mkfmi write (the benchmark index is built with it): the committed golden database (its .fmi was made by the
reference binaries, tests/golden/make_golden.py) and, where oracle/_ref is present, fresh synthetic databases with
the sequence-count / length corner cases of SURVEY.md 7 and two checkpoint exponents."""
import filecmp
import os

import numpy as np
import pytest

import pyoracle as po
from kaiju_amd import mkfmi, synth


def test_golden_fmi_is_reproduced(golden, tmp_path):
    out = str(tmp_path / "g.fmi")
    mkfmi.build_fmi(os.path.join(golden.dir, "db.faa"), out, threads=2, exponent=3)
    assert filecmp.cmp(out, golden.fmi, shallow=False)


@pytest.mark.parametrize("nseq,seed,exponent,threads", [(257, 1, 3, 1), (1000, 2, 3, 4), (1531, 3, 5, 3), (64, 4, 2, 8)])
def test_same_bytes_as_the_reference_builder(tmp_path, nseq, seed, exponent, threads):
    if not po.have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    _, leaves = synth.make_taxonomy(3, 3, 3)
    db = synth.make_db(nseq=nseq, seed=seed, leaves=leaves, max_len=900)
    faa = str(tmp_path / "db.faa")
    synth.write_fasta(db, faa)
    ref = po.ref_build_index(faa, str(tmp_path / "ref"), threads=2, exponent=exponent)
    ours = str(tmp_path / "ours.fmi")
    mkfmi.build_fmi(faa, ours, threads=threads, exponent=exponent)
    assert os.path.getsize(ours) == os.path.getsize(ref)
    assert filecmp.cmp(ours, ref, shallow=False)


@pytest.mark.parametrize("copies,exponent,with_taxids", [(3, 3, False), (7, 2, True), (2, 5, False)])
def test_replicated_database_without_a_second_sort(golden, tmp_path, copies, exponent, with_taxids):
    """kaiju_build_fmi_replicated(file, copies) == kaiju_build_fmi(file with every record repeated `copies` times): the
    shortcut that makes a 2^32-row index for the wide-path tests in seconds writes the very bytes the sorter would"""
    src = os.path.join(golden.dir, "db.faa")
    recs = []
    with open(src) as f:
        for line in f:
            if line.startswith(">"):
                recs.append([line.strip()[1:].split()[0], []])
            else:
                recs[-1][1].append(line.strip())
    taxids = [11, 222, 3333, 44444, 5] if with_taxids else None
    rep = str(tmp_path / "rep.faa")
    with open(rep, "w") as f:
        for i, (name, seq) in enumerate(recs):
            for t in range(copies):
                nm = name
                if taxids:
                    nm = (name[: name.rfind("_")] if "_" in name else name) + "_" + str(taxids[(i + t) % len(taxids)])
                f.write(f">{nm}\n{''.join(seq)}\n")
    a, b = str(tmp_path / "sorted.fmi"), str(tmp_path / "replicated.fmi")
    mkfmi.build_fmi(rep, a, threads=3, exponent=exponent)
    mkfmi.build_fmi_replicated(src, b, copies, threads=3, exponent=exponent, copy_taxids=taxids)
    assert os.path.getsize(a) == os.path.getsize(b)
    assert filecmp.cmp(a, b, shallow=False)


def test_the_64_bit_instantiation_writes_the_same_bytes(golden, tmp_path, monkeypatch):
    """databases of 4 G symbols and more are sorted with 64-bit suffix positions (refseq-class indexes of bench.py): the same
    template, forced onto the golden database and a many-piece FASTA (the reader cuts the file at record starts and parses the
    pieces side by side)"""
    _, leaves = synth.make_taxonomy(3, 3, 3)
    db = synth.make_db(nseq=4001, seed=9, leaves=leaves, max_len=900)
    faa = str(tmp_path / "db.faa")
    synth.write_fasta(db, faa)
    a, b, c = str(tmp_path / "a.fmi"), str(tmp_path / "b.fmi"), str(tmp_path / "c.fmi")
    mkfmi.build_fmi(faa, a, threads=1, exponent=3)
    mkfmi.build_fmi(faa, c, threads=16, exponent=3)
    monkeypatch.setenv("KAIJU_MKFMI_FORCE64", "1")
    mkfmi.build_fmi(faa, b, threads=5, exponent=3)
    assert filecmp.cmp(a, b, shallow=False) and filecmp.cmp(a, c, shallow=False)
    g = str(tmp_path / "g.fmi")
    mkfmi.build_fmi(os.path.join(golden.dir, "db.faa"), g, threads=3, exponent=3)
    assert filecmp.cmp(g, golden.fmi, shallow=False)
    r32, r64 = str(tmp_path / "r32.fmi"), str(tmp_path / "r64.fmi")
    mkfmi.build_fmi_replicated(faa, r64, 3, threads=4, exponent=3)
    monkeypatch.delenv("KAIJU_MKFMI_FORCE64")
    mkfmi.build_fmi_replicated(faa, r32, 3, threads=4, exponent=3)
    assert filecmp.cmp(r32, r64, shallow=False)
