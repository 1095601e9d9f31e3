"""The N>1 layer (kaiju_amd/dist.py) on CPU with the gloo backend, world_size 2: contiguous
sharding, the per-chunk asynchronous gather of hit records to rank 0 and the max-over-ranks
timing reduction.  The per-rank classification itself is replaced by the kernel emulation so that
the gathered records can be compared with a single-process run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import util
from kaiju_amd import dist as kdist


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 100, 101, 713):
        for w in (1, 2, 3, 8):
            b = [kdist.shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, tmpdir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, os.path.join(util.ROOT, "tests"))
    r, lr, w = kdist.init("gloo")
    assert (r, w) == (rank, world)
    g = util.Golden()
    emu = util.Emu()
    h = emu.load(g.fmi)
    n = len(g.reads)
    lo, hi = kdist.shard_bounds(n, rank, world)
    size = (n + world - 1) // world          # equal chunk sizes: pad the shorter shard with empty reads
    reads = g.reads[lo:hi] + [b""] * (size - (hi - lo))
    gath = kdist.HitGatherer(world, rank, keep_results=True)
    chunk = (size + 2) // 3
    for c0 in range(0, size, chunk):
        part = reads[c0:c0 + chunk]
        part = part + [b""] * (chunk - len(part))
        seqs, off = util.pack(part)
        hits, _ = emu.classify(h, util.gp("mem", seg=1), seqs, off)
        gath.gather(torch.from_numpy(hits.view(np.uint8).reshape(-1).copy()))
    gath.wait()
    t = kdist.max_over_ranks(float(rank + 1))
    assert t == float(world)
    kdist.barrier()
    if rank == 0:
        per_rank = [[] for _ in range(world)]
        for bufs in gath.results:
            for rr in range(world):
                per_rank[rr].append(np.frombuffer(bufs[rr].numpy().tobytes(), dtype=util.GPU_HIT))
        out = []
        for rr in range(world):
            a = np.concatenate(per_rank[rr])
            l2, h2 = kdist.shard_bounds(n, rr, world)
            out.append(a[: h2 - l2])
        np.save(os.path.join(tmpdir, "gathered.npy"), np.concatenate(out))
    torch.distributed.destroy_process_group()


def test_two_rank_gather_matches_single_process(tmp_path, emu, golden):
    port = free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(str(tmp_path / "gathered.npy"))
    h = emu.load(golden.fmi)
    ref, _ = emu.classify(h, util.gp("mem", seg=1), golden.seqs, golden.off)
    assert len(got) == len(ref)
    assert (got == ref).all()


def test_bench_refuses_a_launcher_that_disagrees_with_gpus(tmp_path):
    """bench.py --gpus N: a launcher that started another number of ranks is an error, and without a launcher the script only
    starts N ranks itself when N GPUs are visible (none here) - no silent fall-back to fewer GPUs"""
    import subprocess
    import sys
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, bench, "--gpus", "2"], env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode != 0 and b"WORLD_SIZE=1" in r.stderr
    r = subprocess.run([sys.executable, bench, "--gpus", "2"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode != 0 and b"GPU(s) visible" in r.stderr


def _parity_worker(rank, world, port, tmpdir):
    """the per-rank parity plumbing of an N-rank bench line: every rank's sample of reads and of records goes to rank 0"""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    kdist.init("gloo")
    k, L = 50, 150
    reads = np.full((k, L), 65 + rank, dtype=np.uint8)
    recs = np.arange(k * 16, dtype=np.uint8) + rank
    a = kdist.gather_to_root(torch.from_numpy(reads.reshape(-1)), world, rank)
    b = kdist.gather_to_root(torch.from_numpy(recs), world, rank)
    if rank == 0:
        assert len(a) == world and len(b) == world
        for r in range(world):
            assert (a[r].numpy().reshape(k, L) == 65 + r).all()
            assert (b[r].numpy() == (np.arange(k * 16, dtype=np.uint8) + r)).all()
        open(os.path.join(tmpdir, "ok"), "w").write("1")
    else:
        assert a is None and b is None
    kdist.barrier()
    torch.distributed.destroy_process_group()


def test_parity_samples_of_every_rank_reach_rank_0(tmp_path):
    mp.spawn(_parity_worker, args=(2, free_port(), str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(str(tmp_path / "ok"))


def test_bench_parity_comparison_and_strong_scaling_blocks():
    """bench.py's comparison with the reference's lines (C/U, taxon, column 4 for classified reads) and the read blocks of a
    strong-scaling job: the ranks' contiguous shares of ONE workload are the same reads whatever the number of ranks"""
    sys.path.insert(0, util.ROOT)
    import bench
    from kaiju_amd import synth
    ref = (np.array([1, 0, 1, 1], dtype=np.uint8), np.array([7, 0, 9, 9], dtype=np.uint64), np.array([11, 0, 30, 30], dtype=np.int64))
    cls, tax, best = ref[0].copy(), ref[1].copy(), ref[2].copy()
    assert bench.compare_with_reference(cls, tax, best, ref)["mismatches"] == 0
    best[1] = 99                                    # column 4 of an unclassified read is not compared
    assert bench.compare_with_reference(cls, tax, best, ref)["mismatches"] == 0
    best[2] = 31
    tax[3] = 8
    out = bench.compare_with_reference(cls, tax, best, ref)
    assert out["mismatches"] == 2 and out["first_mismatches"] == [2, 3] and out["checked"] == 4
    _, leaves = synth.make_taxonomy(3, 2, 2)
    db = synth.make_db(nseq=301, seed=5, leaves=leaves)
    old = bench.READ_BLOCK
    bench.READ_BLOCK = 64
    try:
        for paired in (False, True):
            whole = bench.reads_of_range(db, 0, 300, paired, 777000)
            assert whole.shape == (300, 300 if paired else 150)
            for world in (2, 3, 4):
                parts = [bench.reads_of_range(db, *kdist.shard_bounds(300, r, world), paired, 777000) for r in range(world)]
                assert (np.concatenate(parts, axis=0) == whole).all()
    finally:
        bench.READ_BLOCK = old
