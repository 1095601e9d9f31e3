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
