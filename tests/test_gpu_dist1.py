"""8-GPU readiness on a box with one GPU: bench.py with a process group of ONE rank on the nccl (= RCCL) backend
(KAIJU_DIST_FORCE_INIT=1), so that init, barrier, the max-over-ranks all-reduce and the gather of the 16-byte records run
through RCCL exactly as at N > 1; the per-rank rate must equal the job's.  (The N = 2 logic itself is covered on CPU with
gloo: tests/test_distributed.py.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_single_rank_through_rccl(gpu_lib, tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, KAIJU_DIST_FORCE_INIT="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port), KAIJU_BENCH_WORK=str(tmp_path / "work"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--reads", "200000",
                        "--nseq", "20001", "--legs", "", "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    out_lines = r.stdout.decode().strip().splitlines()
    assert out_lines[-1].startswith("{"), out_lines[-3:]           # the JSON line is the LAST line of stdout (RCCL's own chatter first)
    line = json.loads(out_lines[-1])
    assert line["n_gpus"] == 1 and line["scaling"] == "weak" and line["value"] > 0
    pg = line["config"]["process_group"]
    assert pg == {"backend": "nccl", "world_size": 1}, pg
    assert abs(line["config"]["per_rank_units_per_s"][0] - line["value"]) < 0.02 * line["value"]


def test_the_librarys_own_gather(gpu_lib, golden, tmp_path):
    """kaiju_gpu_comm_create / kaiju_gpu_gather_compact: the collective of the path in the PRODUCT (librccl opened by
    libkaiju_gpu.so itself, ncclGather of 16-byte records to the root) - a communicator of one rank on this box's GPU: the
    records a batch left on the device arrive unchanged, stream-ordered behind the kernels that wrote them.
    (Device buffers through ctypes on the HIP runtime the library is linked against, as in test_gpu_parity: torch in THIS
    process would bring a second ROCm stack along, whose runtime finds no device; bench.py --gather lib, below, is the
    torch-first order - there the library picks the librccl that lies next to torch's HIP runtime.)"""
    import ctypes as C
    import numpy as np
    from test_gpu_parity import Hip
    api = gpu_lib
    hip = Hip()
    idx = api.Index(golden.fmi)
    tax = api.Taxonomy(golden.nodes)
    dtax = api.DeviceTaxonomy(tax, 0)
    clf = api.Classifier(idx, api.default_params("mem"))
    want = clf.classify_compact(dtax, golden.seqs, golden.off)
    n = len(want)
    comm = api.Comm(str(tmp_path / "comm.id"), 0, 1, 0)
    api.lib().kaiju_gpu_comm_library.restype = C.c_char_p
    assert b"rccl" in api.lib().kaiju_gpu_comm_library()
    seqs = np.ascontiguousarray(golden.seqs, dtype=np.uint8)
    off = np.ascontiguousarray(golden.off, dtype=np.uint64)
    d_seqs, d_off = hip.malloc(seqs.nbytes + 64), hip.malloc(off.nbytes)
    d_hits, d_rec, d_all = hip.malloc(n * 184), hip.malloc(n * 16), hip.malloc(n * 16)
    hip.h2d(d_seqs, seqs)
    hip.h2d(d_off, off)
    hip.memset(d_all, n * 16)
    clf.set_max_read_length(int((off[1:] - off[:-1]).max()))
    stream = clf.stream_handle()
    clf.classify_device(d_seqs, seqs.nbytes, d_off, n, d_hits, stream=0)
    clf.lca_device(dtax, d_hits, n, d_rec, stream=0)
    comm.gather_compact(d_rec, n, d_all, root=0, stream=stream)
    clf.synchronize()
    got = np.frombuffer(hip.d2h(d_all, n * 16).tobytes(), dtype=api.COMPACT_DTYPE)
    assert (got == want).all()
    assert (got["lca"] != 0).mean() > 0.3
    comm.close()
    with pytest.raises(api.KaijuGpuError):
        api.Comm(str(tmp_path / "comm2.id"), 1, 1, 0)               # rank outside the world
    for p in (d_seqs, d_off, d_hits, d_rec, d_all):
        hip.free(p)


def test_single_rank_bench_with_the_librarys_gather(gpu_lib, tmp_path):
    """bench.py --gather lib: the N-rank harness with the library's collective in place of torch.distributed.gather"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, KAIJU_DIST_FORCE_INIT="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port), KAIJU_BENCH_WORK=str(tmp_path / "work"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--reads", "200000",
                        "--nseq", "20001", "--legs", "", "--no-cpu-baseline", "--gather", "lib"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert line["config"]["gather_by"].startswith("library"), line["config"]["gather_by"]
    assert line["value"] > 0


@pytest.mark.parametrize("gather", ["torch", "lib"])
def test_two_halves_on_two_contexts_gather_twice(gpu_lib, tmp_path, gather):
    """the default shape of a MEM step of 4 M reads and more - two halves on two contexts, each half gathered in stream order
    behind its kernels - through RCCL with a process group of one rank (what every rank does at N > 1)"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, KAIJU_DIST_FORCE_INIT="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port), KAIJU_BENCH_WORK=str(tmp_path / "work"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--reads", "4000000",
                        "--nseq", "20001", "--legs", "", "--no-cpu-baseline", "--gather", gather], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    err = r.stderr.decode()
    line = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert line["value"] > 0 and line["config"]["process_group"]["backend"] == "nccl"
    detail = json.loads(err[err.index("[bench] detail: ") + len("[bench] detail: "):].splitlines()[0])
    assert detail["config"]["contexts_in_flight"] == 2 and detail["config"]["chunk"] == 2000000, detail["config"]
