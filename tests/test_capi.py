"""The C-ABI library: it builds, loads, exports every symbol declared in include/kaiju_gpu.h,
its host-side helpers work without a GPU, and the compute entry points fail loudly (never fall
back to the CPU) when no HIP device is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import util
from kaiju_amd import api, build

HEADER = os.path.join(util.ROOT, "include", "kaiju_gpu.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(kaiju_[a-z_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = api.lib()
    names = declared_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/kaiju_gpu.h but not exported"
    assert lib.kaiju_gpu_abi_version() == 1


def test_struct_layouts_match_header():
    assert C.sizeof(api.Params) == 48
    assert api.HIT_DTYPE.itemsize == 184
    assert api.RESULT_DTYPE.itemsize == 16


def test_product_library_does_not_link_the_oracle():
    out = os.popen(f"ldd {build.LIB}").read() + os.popen(f"nm -D {build.LIB}").read()
    assert "kaiju_oracle" not in out and "ko_classify" not in out and "libkaijuref" not in out


def test_default_params_mirror_config():
    p = api.default_params("greedy")
    assert (p.mode, p.min_fragment_length, p.mismatches, p.min_score, p.seed_length, p.seg, p.use_evalue) == \
        (1, 11, 3, 65, 7, 1, 1)
    assert abs(p.min_evalue - 0.01) < 1e-15 and p.max_matches_SI == 20 and p.max_match_ids == 20
    p = api.default_params("mem")
    assert p.mode == 0 and p.use_evalue == 0        # "-a mem" clears use_Evalue, kaiju.cpp:77-80


@pytest.mark.skipif(api.device_count() > 0, reason="checks the no-device behaviour")
def test_no_device_fails_loudly(golden):
    with pytest.raises(api.KaijuGpuError) as e:
        api.Index(golden.fmi)
    assert "no usable HIP device" in str(e.value)


def test_bad_arguments():
    L = api.lib()
    assert L.kaiju_gpu_index_load(None, 0, None) < 0
    h = C.c_void_p()
    assert L.kaiju_taxonomy_load(b"/nonexistent/nodes.dmp", C.byref(h)) == -2
    assert b"I/O" in L.kaiju_gpu_strerror(-2)


def test_taxonomy_lca_matches_oracle(oracle, golden):
    tax = api.Taxonomy(golden.nodes)
    otax = oracle.load_nodes(golden.nodes)
    ids = [int(l.split("\t")[0]) for l in open(golden.nodes)]
    rng = np.random.default_rng(3)
    for _ in range(2000):
        k = int(rng.integers(1, 8))
        pick = [int(x) for x in rng.choice(ids + [999999999], size=k, replace=False)]
        arr = (C.c_uint64 * k)(*pick)
        assert tax.lca(pick) == oracle.lib.ko_lca(otax, arr, k), pick


def test_taxonomy_depth_rules(tmp_path):
    """lca_from_ids measures depth up to a self-parent root or an unknown parent (util.cpp:218-224)"""
    p = tmp_path / "nodes.dmp"
    p.write_text("1\t|\t1\t|\n2\t|\t1\t|\n3\t|\t2\t|\n4\t|\t2\t|\n5\t|\t4\t|\n10\t|\t77\t|\n11\t|\t10\t|\nbad line\n\n12\t|\t10\t|\n")
    tax = api.Taxonomy(str(p))
    assert tax.lca([3, 5]) == 2
    assert tax.lca([3, 4, 5]) == 2
    assert tax.lca([5]) == 5
    assert tax.lca([11, 12]) == 10
    assert tax.lca([3, 424242]) == 3          # ids missing from the tree are dropped
    assert tax.lca([424242, 434343]) == 0


def test_device_lca_logic_matches_host_lca(tmp_path):
    """tax_lca of kj_core.h (what k_lca runs) on the hash-table form of the tree == kaiju_taxonomy_lca, including ids
    outside the tree, nodes below an unknown parent and separate roots"""
    import ctypes as C
    import util
    from kaiju_amd import api
    emu = util.Emu()
    emu.lib.emu_lca.restype = C.c_uint64
    emu.lib.emu_lca.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    rng = np.random.default_rng(5)
    # a forest: root 1 with a random tree, a second root 5000, a node whose parent is missing
    lines = ["1\t|\t1\t|\tno rank\t|\n", "5000\t|\t5000\t|\tno rank\t|\n", "6000\t|\t777777\t|\tspecies\t|\n",
             "6001\t|\t6000\t|\tspecies\t|\n"]
    for i in range(2, 3000):
        lines.append(f"{i}\t|\t{int(rng.integers(1, i))}\t|\tclade\t|\n")
    for i in range(5001, 5050):
        lines.append(f"{i}\t|\t{int(rng.integers(5000, i))}\t|\tclade\t|\n")
    path = tmp_path / "nodes.dmp"
    path.write_text("".join(lines))
    tax = api.Taxonomy(str(path))
    pool = list(range(1, 3000)) + list(range(5000, 5050)) + [6000, 6001, 777777, 424242]
    for case in range(20000):
        n = int(rng.integers(1, 22))
        if case % 3 == 0:
            ids = rng.integers(1, 3000, n).astype(np.uint64)       # one tree
        else:
            ids = rng.choice(pool, n).astype(np.uint64)
        assert tax.lca(ids) == emu.lib.emu_lca(tax._h, ids.ctypes.data, n), ids


def test_index_image_roundtrip(golden, tmp_path):
    """device image of an index (kaiju_gpu_index_write_image): every packed array survives the file; a truncated or
    foreign file is refused"""
    import ctypes as C
    import util
    from kaiju_amd import api
    emu = util.Emu()
    emu.lib.emu_image_roundtrip.argtypes = [C.c_char_p, C.c_char_p]
    img = str(tmp_path / "db.kjimg")
    assert emu.lib.emu_image_roundtrip(golden.fmi.encode(), img.encode()) == 0
    L = api.lib()
    L.kaiju_gpu_index_write_image.argtypes = [C.c_char_p, C.c_char_p]
    img2 = str(tmp_path / "db2.kjimg")
    assert L.kaiju_gpu_index_write_image(golden.fmi.encode(), img2.encode()) == 0
    assert open(img, "rb").read() == open(img2, "rb").read()
    assert L.kaiju_gpu_index_write_image(b"/nonexistent.fmi", img2.encode()) != 0


def test_damaged_index_files_return_a_status(golden, tmp_path):
    """header fields of a .fmi are checked against the file size before anything is allocated from them, and no C++
    exception crosses the C ABI: a damaged or truncated file gives a negative status (parsing needs no GPU:
    kaiju_gpu_index_write_image)"""
    import struct
    L = api.lib()
    L.kaiju_gpu_index_write_image.argtypes = [C.c_char_p, C.c_char_p]
    data = bytearray(open(golden.fmi, "rb").read())
    img = str(tmp_path / "x.kjimg")
    assert L.kaiju_gpu_index_write_image(golden.fmi.encode(), img.encode()) == 0
    cases = {}
    d = bytearray(data); struct.pack_into("<i", d, 8, 2 ** 31 - 1); cases["nseq"] = d            # int32 nseq
    d = bytearray(data); struct.pack_into("<q", d, 0, 2 ** 62); cases["len"] = d                  # int64 len
    cases["truncated"] = data[: len(data) // 3]
    cases["empty"] = bytearray()
    for name, d in cases.items():
        p = str(tmp_path / f"{name}.fmi")
        open(p, "wb").write(bytes(d))
        rc = L.kaiju_gpu_index_write_image(p.encode(), img.encode())
        assert rc < 0, name
        assert L.kaiju_gpu_strerror(rc)


def test_image_info_locates_the_streamed_arrays(golden, tmp_path):
    """kaiju_gpu_index_image_info (host only) reads an image the way the loader does - header and small arrays into memory, the
    arrays that grow with the index only located for the streamed upload - and says how many bytes a load streams"""
    L = api.lib()
    L.kaiju_gpu_index_write_image.argtypes = [C.c_char_p, C.c_char_p]
    L.kaiju_gpu_index_image_info.argtypes = [C.c_char_p, C.POINTER(api.IndexInfo), C.POINTER(C.c_uint64)]
    img = str(tmp_path / "db.kjimg")
    assert L.kaiju_gpu_index_write_image(golden.fmi.encode(), img.encode()) == 0
    info, streamed = api.IndexInfo(), C.c_uint64(0)
    assert L.kaiju_gpu_index_image_info(img.encode(), C.byref(info), C.byref(streamed)) == 0
    import struct
    bwtlen, nseq = struct.unpack_from("<qi", open(golden.fmi, "rb").read(12))
    assert (info.bwtlen, info.nseq, info.alen, info.chpt_exp) == (bwtlen, nseq, 21, 3)
    # rank blocks (128 B per 64 rows) + sampled sequence numbers and offsets + terminator rows + the host's k = 5 table
    n_sa = ((bwtlen - 1) >> 3) - (((nseq - 1) >> 3) + 1) + 1
    want = ((bwtlen >> 6) + 1) * 128 + nseq * 8 + 20 ** 5 * 8
    assert want + 8 * (n_sa - 1) <= streamed.value <= want + 8 * (n_sa + 1)
    assert streamed.value < os.path.getsize(img) and streamed.value <= info.device_bytes
    # a truncated image and a .fmi are refused
    open(str(tmp_path / "cut.kjimg"), "wb").write(open(img, "rb").read()[: os.path.getsize(img) // 2])
    assert L.kaiju_gpu_index_image_info(str(tmp_path / "cut.kjimg").encode(), C.byref(info), None) < 0
    assert L.kaiju_gpu_index_image_info(golden.fmi.encode(), C.byref(info), None) < 0


def test_streamed_pack_of_a_fmi_equals_the_host_pack(golden, tmp_path, monkeypatch):
    """fmi_stream.h: a .fmi whose BWT and samples are packed piece by piece with running counts (what the kernels of
    fmi_stream.hip do on the device) gives the arrays of PackedIndex::build word for word - rank blocks, count bases, terminator
    rows, sampled sequence numbers / offsets / taxon ids, C[] - narrow and wide, with count bases that start inside a piece, a
    BWT that ends on a block boundary's far side, pieces of one group and pieces larger than the file"""
    import util
    from kaiju_amd import mkfmi, synth
    emu = util.Emu()
    emu.lib.emu_stream_pack_check.argtypes = [C.c_char_p, C.c_uint64]
    _, leaves = synth.make_taxonomy(3, 3, 3)
    files = [golden.fmi]
    for nseq, seed in ((700, 3), (1600, 4)):                  # (1600 % 8 == 0: the short sample array of KAIJU_IDX_WARN_SA_SHORT)
        db = synth.make_db(nseq=nseq, seed=seed, leaves=leaves, max_len=700)
        faa, fmi = str(tmp_path / f"db{nseq}.faa"), str(tmp_path / f"db{nseq}.fmi")
        synth.write_fasta(db, faa)
        mkfmi.build_fmi(faa, fmi, threads=2, exponent=3)
        files.append(fmi)
    # databases whose BWT ends exactly on a group / block boundary (the block behind the end is a group, or a block, of its own)
    rng = np.random.default_rng(8)
    for rows in (32768, 32768 + 64, 16384 * 3 - 1):
        nseq, lens = 40, []
        left = rows - nseq
        for i in range(nseq):
            l = left - 30 * (nseq - 1 - i) if i == nseq - 1 else int(rng.integers(30, 2 * left // (nseq - i) - 30))
            lens.append(l); left -= l
        assert sum(lens) + nseq == rows and min(lens) >= 30
        faa, fmi = str(tmp_path / f"rows{rows}.faa"), str(tmp_path / f"rows{rows}.fmi")
        with open(faa, "w") as f:
            for i, l in enumerate(lens):
                f.write(f">s{i}_{leaves[i % len(leaves)]}\n" + "".join(rng.choice(list("ACDEFGHIKLMNPQRSTVWY"), l)) + "\n")
        mkfmi.build_fmi(faa, fmi, threads=2, exponent=3)
        import struct
        assert struct.unpack_from("<q", open(fmi, "rb").read(8))[0] == rows
        files.append(fmi)
    for fmi in files:
        for wide in (None, "16", "17", "31"):
            if wide is None:
                monkeypatch.delenv("KAIJU_GPU_FORCE_WIDE", raising=False)
            else:
                monkeypatch.setenv("KAIJU_GPU_FORCE_WIDE", wide)
            for piece in (16384, 49152, 1 << 20, 1 << 26):
                assert emu.lib.emu_stream_pack_check(fmi.encode(), piece) == 0, (fmi, wide, piece)


def test_the_gather_entry_points_without_a_device(tmp_path):
    """kaiju_gpu_comm_create (the product-side RCCL gather): bad arguments and a missing device give a status, nothing is
    loaded or written (librccl is only opened once a device is there)"""
    L = api.lib()
    L.kaiju_gpu_comm_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.kaiju_gpu_comm_last_error.restype = C.c_char_p
    L.kaiju_gpu_gather_compact.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.c_void_p]
    h = C.c_void_p()
    path = str(tmp_path / "comm.id").encode()
    assert L.kaiju_gpu_comm_create(path, 3, 2, 0, C.byref(h)) == -1 and not h.value        # KAIJU_GPU_ERR_ARG
    assert L.kaiju_gpu_comm_create(None, 0, 1, 0, C.byref(h)) == -1
    if api.device_count() == 0:
        assert L.kaiju_gpu_comm_create(path, 0, 1, 0, C.byref(h)) == -4                    # KAIJU_GPU_ERR_NO_DEVICE
        assert not os.path.exists(path.decode())
    assert L.kaiju_gpu_gather_compact(None, None, 0, None, 0, None) == -1
    L.kaiju_gpu_comm_destroy.argtypes = [C.c_void_p]
    L.kaiju_gpu_comm_destroy(None)


def test_comm_rendezvous_ignores_what_an_earlier_job_left(tmp_path):
    """kaiju_gpu_comm_create's file exchange (rccl_gather.cpp: exchange_id), host only: a stale id file and stale nonce files
    of a dead job lie at the path, the ranks start in any order - every rank ends up with THIS job's 128 bytes and nothing is
    left behind.  (a reader used to take any 128-byte file for the id.)"""
    import threading
    import time
    L = api.lib()
    L.kaiju_gpu_comm_exchange_id.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_double]
    path = tmp_path / "comm.id"
    world = 4
    for order in ([0, 1, 2, 3], [3, 2, 1, 0], [2, 0, 3, 1]):
        # leftovers of a job that died: an id file of the right size for this world, a nonce file of rank 2
        path.write_bytes(b"\x31\x30\x4d\x4d\x4f\x43\x4a\x4b" + world.to_bytes(8, "little") + b"\x07" * (8 * world) + b"\xee" * 128)
        (tmp_path / "comm.id.r2").write_bytes(b"\x07" * 8)
        job_id = os.urandom(128)
        bufs = [C.create_string_buffer(job_id if r == 0 else b"\0" * 128, 128) for r in range(world)]
        rcs = [None] * world

        def run(r):
            rcs[r] = L.kaiju_gpu_comm_exchange_id(str(path).encode(), r, world, bufs[r], 30.0)
        th = []
        for r in order:
            t = threading.Thread(target=run, args=(r,))
            t.start()
            th.append(t)
            time.sleep(0.05)
        for t in th:
            t.join()
        assert rcs == [0] * world, rcs
        for r in range(world):
            assert bufs[r].raw == job_id, r
        assert sorted(os.listdir(tmp_path)) == [], os.listdir(tmp_path)
    # a rank whose job never shows up gives up with a status (and removes its nonce file)
    b = C.create_string_buffer(128)
    assert L.kaiju_gpu_comm_exchange_id(str(path).encode(), 1, 2, b, 0.3) == -2
    assert os.listdir(tmp_path) == []
    # one rank: nobody to wait for
    assert L.kaiju_gpu_comm_exchange_id(str(path).encode(), 0, 1, b, 1.0) == 0
