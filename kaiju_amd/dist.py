"""Multi-GPU layer: reads shard embarrassingly across the GPUs of one node (one process per GPU),
the index is replicated, and the per-read hit records are collected on rank 0 with ONE gather per
chunk (RCCL over xGMI when the backend is "nccl").  There is no other exchange on the path
(SURVEY.md §8e): no reduce, no all-to-all.

Everything here works on CPU tensors with the ``gloo`` backend too, which is how the N>1 logic
is tested without GPUs (tests/test_distributed.py).
"""
from __future__ import annotations

import os

import numpy as np

HIT_BYTES = 184


def env_world():
    """(rank, local_rank, world_size) as set by torch.distributed.run; (0, 0, 1) standalone"""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend: str):
    import torch.distributed as dist
    rank, local_rank, world = env_world()
    # KAIJU_DIST_FORCE_INIT=1: a process group of ONE rank, so that barrier / all-reduce / gather of a single-GPU run go
    # through the backend (RCCL) as they do at N > 1 - the readiness check for boxes with one GPU (tests/test_gpu_dist1.py)
    force = os.environ.get("KAIJU_DIST_FORCE_INIT") == "1"
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_bounds(n_items: int, rank: int, world: int):
    """contiguous input-order block of a strong-scaling job (block b -> rank b)"""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class HitGatherer:
    """Collects fixed-size hit records of equally sized chunks on rank 0.

    ``gather(chunk)`` issues one asynchronous ``dist.gather`` (ncclGather-equivalent) and returns
    immediately so that the next chunk's kernels overlap with the transfer; ``wait()`` drains.
    On rank 0 ``results`` holds, per call, a list of ``world`` tensors (rank order)."""

    def __init__(self, world: int, rank: int, keep_results: bool = True):
        self.world, self.rank = world, rank
        self.pending = []
        self.results = []
        self.keep = keep_results
        self.bytes_gathered = 0

    def gather(self, chunk):
        import torch
        import torch.distributed as dist
        if self.world == 1 and not dist.is_initialized():
            if self.keep:
                self.results.append([chunk])
            return
        bufs = [torch.empty_like(chunk) for _ in range(self.world)] if self.rank == 0 else None
        work = dist.gather(chunk, gather_list=bufs, dst=0, async_op=True)
        self.pending.append((work, bufs, chunk))
        self.bytes_gathered += chunk.numel() * chunk.element_size() * (self.world - 1)
        if len(self.pending) > 4:
            self._retire(self.pending.pop(0))

    def _retire(self, item):
        work, bufs, _ = item
        work.wait()
        if self.rank == 0 and self.keep:
            self.results.append(bufs)

    def wait(self):
        while self.pending:
            self._retire(self.pending.pop(0))


class LibGatherer:
    """The same through the LIBRARY's collective (kaiju_gpu_comm_create / kaiju_gpu_gather_compact: librccl opened by the
    library itself, no torch in the data path): one RCCL gather of 16-byte records per chunk to rank 0, asynchronous on the
    stream the chunk's kernels run on.  ``comm``: api.Comm of this rank (make_comm below).
    Without ``keep_results`` rank 0 receives every chunk of a size into ONE buffer: the gathers are stream-ordered, so a later
    chunk overwrites an earlier one - that mode is for timing; with ``keep_results`` every gather gets a buffer of its own.
    What a gather costs on one GPU: 0.04 ms of kernel and 0.3 - 0.5 ms of host-side enqueue per call - one gather
    per step, not one per chunk."""

    def __init__(self, comm, keep_results: bool = False):
        self.comm, self.world, self.rank = comm, comm.world, comm.rank
        self.keep = keep_results
        self.results = []
        self.bytes_gathered = 0
        self._recv = {}

    def gather(self, chunk):
        import torch
        n = chunk.numel() * chunk.element_size() // 16
        recv_ptr = 0
        if self.rank == 0:
            key = (chunk.numel(), len(self.results) if self.keep else 0)
            if key not in self._recv:
                self._recv[key] = torch.empty(chunk.numel() * self.world, dtype=chunk.dtype, device=chunk.device)
            recv_ptr = self._recv[key].data_ptr()
            if self.keep:
                self.results.append(self._recv[key])
        self.comm.gather_compact(chunk.data_ptr(), n, recv_ptr, 0, torch.cuda.current_stream(chunk.device).cuda_stream)
        self.bytes_gathered += chunk.numel() * chunk.element_size() * (self.world - 1)

    def wait(self):
        pass                      # (stream-ordered: the caller synchronises the streams it launched on)


def make_comm(work_dir: str, rank: int, world: int, device: int):
    """api.Comm of this rank; the rendezvous file gets a name no earlier job has used (rank 0 picks it, torch.distributed -
    when it is up - tells the others; a single rank needs nobody)"""
    import time
    import torch.distributed as dist
    from . import api
    name = [f"{work_dir}/kaiju_comm_{os.getpid()}_{time.time_ns()}.id"]
    if dist.is_initialized():
        dist.broadcast_object_list(name, src=0)
    comm = api.Comm(name[0], rank, world, device)
    if dist.is_initialized():
        dist.barrier()
    if rank == 0:
        try:
            os.remove(name[0])
        except OSError:
            pass
    return comm


def gather_to_root(t, world: int, rank: int):
    """one tensor of every rank on rank 0 (a list in rank order; None elsewhere) - the parity samples of an N-rank bench
    line, not the data path"""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return [t]
    bufs = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, gather_list=bufs, dst=0)
    return bufs


def max_over_ranks(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    import torch.distributed as dist
    if dist.is_initialized():
        dist.barrier()


def hits_from_bytes(buf: np.ndarray):
    from .api import HIT_DTYPE
    return np.frombuffer(buf.tobytes(), dtype=HIT_DTYPE)
