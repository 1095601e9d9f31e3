"""A .fmi streamed to HBM and packed there (kaiju_amd/csrc/fmi_stream.hip, KAIJU_GPU_FMI_STREAM) against the same file parsed
and packed on the host (host_index.cpp: PackedIndex::build; reference: readIndexes bwt/bwt.c:78-88, read_fmi
fmicommon.h:190-217, read_suffixArray_body suffixArray.c:313-321): every array the index holds in HBM must come out the same,
and so must the records of the reads."""
import os
import struct

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def synth_db(tmp_path_factory):
    from kaiju_amd import mkfmi, synth
    d = tmp_path_factory.mktemp("stream")
    _, leaves = synth.make_taxonomy(3, 3, 3)
    out = []
    for nseq, seed in ((2500, 31), (1600, 32)):           # (1600 % 8 == 0: KAIJU_IDX_WARN_SA_SHORT)
        db = synth.make_db(nseq=nseq, seed=seed, leaves=leaves, max_len=900)
        faa, fmi = str(d / f"db{nseq}.faa"), str(d / f"db{nseq}.fmi")
        synth.write_fasta(db, faa)
        mkfmi.build_fmi(faa, fmi, threads=4, exponent=3)
        out.append((db, fmi))
    return out


def _load(api, fmi, env, id_mode=0):
    old = {k: os.environ.get(k) for k in env}
    for k, v in env.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    try:
        return api.Index(fmi, device=0, id_mode=id_mode)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("wide", [None, "16", "20"])
def test_streamed_fmi_gives_the_arrays_of_the_host_pack(gpu_lib, golden, synth_db, wide):
    api = gpu_lib
    from kaiju_amd import synth
    files = [golden.fmi] + [f for _, f in synth_db]
    for fmi in files:
        base = {"KAIJU_GPU_FORCE_WIDE": wide}
        host = _load(api, fmi, dict(base, KAIJU_GPU_FMI_STREAM="0"))
        want = host.digest()
        assert want["rank_blocks"] and want["term_rows"] and want["kmer_table"]
        assert (want["count_bases"] != 0) == (wide is not None)
        for piece in ("16", "48", None):
            st = _load(api, fmi, dict(base, KAIJU_GPU_FMI_STREAM="1", KAIJU_GPU_STREAM_PIECE_KB=piece))
            got = st.digest()
            assert got == want, (fmi, wide, piece, {k: (got[k], want[k]) for k in got if got[k] != want[k]})
            assert st.info.warnings == host.info.warnings and st.info.device_bytes == host.info.device_bytes
            assert st.footprint.as_dict() == host.footprint.as_dict()
            st.close()
        host.close()
    # the reads classify alike (MEM, Greedy, pairs) on the streamed index
    db, fmi = synth_db[0]
    seqs, off = synth.pack_reads(synth.make_reads(db, 4000, seed=7))
    m1, m2 = synth.make_pairs(db, 1000, seed=8)
    pseqs, poff = synth.pack_reads(m1, m2)
    a = _load(api, fmi, {"KAIJU_GPU_FORCE_WIDE": wide, "KAIJU_GPU_FMI_STREAM": "0"})
    b = _load(api, fmi, {"KAIJU_GPU_FORCE_WIDE": wide, "KAIJU_GPU_FMI_STREAM": "1", "KAIJU_GPU_STREAM_PIECE_KB": "32"})
    for mode in ("mem", "greedy"):
        for s, o, pe in ((seqs, off, False), (pseqs, poff, True)):
            ca, cb = api.Classifier(a, api.default_params(mode)), api.Classifier(b, api.default_params(mode))
            ha, hb = ca.classify(s, o, paired=pe), cb.classify(s, o, paired=pe)
            assert (ha == hb).all(), (mode, pe)
            assert (ha["n_ids"] > 0).mean() > 0.4
            ca.close(); cb.close()
    a.close(); b.close()


def test_streamed_fmi_with_sequence_ids(gpu_lib, synth_db):
    """kaijux / kaijup ids (KAIJU_GPU_IDS_SEQUENCE): the taxon id of a sampled row is its sequence number"""
    api = gpu_lib
    _, fmi = synth_db[0]
    a = _load(api, fmi, {"KAIJU_GPU_FMI_STREAM": "0"}, id_mode=api.IDS_SEQUENCE)
    b = _load(api, fmi, {"KAIJU_GPU_FMI_STREAM": "1", "KAIJU_GPU_STREAM_PIECE_KB": "64"}, id_mode=api.IDS_SEQUENCE)
    assert a.digest() == b.digest()
    a.close(); b.close()


def test_streamed_fmi_refuses_what_the_host_pack_refuses(gpu_lib, golden, tmp_path):
    """a byte outside the code table / a terminator too many in the BWT: the status and message of PackedIndex::build"""
    api = gpu_lib
    data = bytearray(open(golden.fmi, "rb").read())
    length, nseq, alen = struct.unpack_from("<qii", data, 0)
    # the BWT lies behind the FMI header: find it the way the loader does
    at = 16 + alen
    salen, ncheck, chpt_exp, nbytes = struct.unpack_from("<qqii", data, at)
    at += 8 + 8 + 4 + 4 + 4 + 4 + 8 + 8 + 4
    for _ in range(nseq):
        at += 1 + data[at]
    at += nseq * 12 + ncheck * nbytes
    f_alen, bwtlen = struct.unpack_from("<iq", data, at)
    assert f_alen == alen and bwtlen == length
    bwt_at = at + 4 + 8 + 4 + 4
    startl = struct.unpack_from(f"<{alen + 1}i", data, len(data) - 4 * (alen + 1))
    assert startl[0] == 0 and startl[alen] <= 256
    cases = {}
    if startl[alen] < 256:
        d = bytearray(data); d[bwt_at + bwtlen // 2] = 255; cases["stray"] = d
    d = bytearray(data); d[bwt_at + bwtlen // 3] = 0 if d[bwt_at + bwtlen // 3] != 0 else d[bwt_at + bwtlen // 3 + 1]; cases["terminators"] = d
    for name, d in cases.items():
        p = str(tmp_path / f"{name}.fmi")
        open(p, "wb").write(bytes(d))
        errs = []
        for stream in ("0", "1"):
            with pytest.raises(api.KaijuGpuError) as e:
                _load(api, p, {"KAIJU_GPU_FMI_STREAM": stream})
            errs.append(str(e.value))
        assert errs[0] == errs[1], (name, errs)
