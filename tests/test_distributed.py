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
