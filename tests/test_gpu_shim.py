"""The reference's own `kaiju` with ConsumerThread::doWork() replaced by the C-ABI shim
(integration/ConsumerThread_gpu.cpp, built by integration/apply_shim.py into oracle/_ref/kaiju_gpu_shim):
the reference's ingest, queue, Config, lca_from_ids and output code around the HIP kernels.  Its lines
must equal those of the unmodified reference binary — the committed goldens (`kaiju -v -z 1`) and, where
oracle/_ref/kaiju travelled along, a live run.  First caller of kaiju_gpu_index_from_host."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "oracle", "_ref", "kaiju_gpu_shim")
REF = os.path.join(ROOT, "oracle", "_ref", "kaiju")


def need_shim():
    if not os.path.exists(SHIM):
        if os.path.exists("/root/reference/src/kaiju.cpp"):
            subprocess.run(["python3", os.path.join(ROOT, "integration", "apply_shim.py")], check=True)
        else:
            pytest.skip("oracle/_ref/kaiju_gpu_shim was not built (needs /root/reference: integration/apply_shim.py)")


def lines(path):
    return open(path).read().split("\n")


def run(binary, golden, tmp_path, name, args):
    out = str(tmp_path / name)
    subprocess.run([binary, "-t", golden.nodes, "-f", golden.fmi, "-o", out] + args, check=True, stderr=subprocess.DEVNULL)
    return out


@pytest.mark.parametrize("mode,seg", [("mem", 1), ("mem", 0), ("greedy", 1), ("greedy", 0)])
def test_shim_single_and_paired_verbose(gpu_lib, golden, tmp_path, mode, seg):
    need_shim()
    x = [] if seg else ["-X"]
    out = run(SHIM, golden, tmp_path, "se.tsv", ["-i", os.path.join(golden.dir, "reads.fq"), "-a", mode, "-v", "-z", "1"] + x)
    assert lines(out) == lines(os.path.join(golden.dir, f"ref_{mode}_{seg}.tsv"))
    out = run(SHIM, golden, tmp_path, "pe.tsv", ["-i", os.path.join(golden.dir, "pairs_1.fq"), "-j",
                                                 os.path.join(golden.dir, "pairs_2.fq"), "-a", mode, "-v", "-z", "1"] + x)
    assert lines(out) == lines(os.path.join(golden.dir, f"ref_{mode}_{seg}_pe.tsv"))


def test_shim_options_and_plain_output(gpu_lib, golden, tmp_path):
    need_shim()
    rd = os.path.join(golden.dir, "reads.fq")
    out = run(SHIM, golden, tmp_path, "a.tsv", ["-i", rd, "-a", "greedy", "-e", "5", "-s", "50", "-E", "10", "-v", "-z", "1"])
    assert lines(out) == lines(os.path.join(golden.dir, "ref_greedy_e5_s50.tsv"))
    out = run(SHIM, golden, tmp_path, "b.tsv", ["-i", rd, "-a", "mem", "-m", "15", "-v", "-z", "1"])
    assert lines(out) == lines(os.path.join(golden.dir, "ref_mem_m15.tsv"))
    # without -v: three columns, the classify_batch entry point; several small batches
    env = dict(os.environ, KAIJU_GPU_BATCH="97")
    out = str(tmp_path / "c.tsv")
    subprocess.run([SHIM, "-t", golden.nodes, "-f", golden.fmi, "-o", out, "-i", rd, "-a", "greedy", "-z", "1"], check=True,
                   stderr=subprocess.DEVNULL, env=env)
    want = ["\t".join(l.split("\t")[:3]) for l in lines(os.path.join(golden.dir, "ref_greedy_1.tsv"))]
    assert lines(out) == want


def test_shim_protein_input(gpu_lib, golden, tmp_path):
    """kaiju -p through the shim (ConsumerThread.cpp:640-646,659-696)"""
    need_shim()
    for mode in ("mem", "greedy"):
        out = run(SHIM, golden, tmp_path, "p.tsv", ["-i", os.path.join(golden.dir, "prot.fa"), "-p", "-a", mode, "-v", "-z", "1"])
        assert lines(out) == lines(os.path.join(golden.dir, f"refp_{mode}_1.tsv")), mode


def test_shim_vs_live_reference_multithreaded(gpu_lib, golden, tmp_path):
    """-z 3 consumers (three GPU contexts on one index) against the unmodified reference run here: sorted lines"""
    need_shim()
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/kaiju not present")
    rd = os.path.join(golden.dir, "reads.fq")
    for mode in ("mem", "greedy"):
        a = run(SHIM, golden, tmp_path, "s.tsv", ["-i", rd, "-a", mode, "-z", "3"])
        b = run(REF, golden, tmp_path, "r.tsv", ["-i", rd, "-a", mode, "-z", "3"])
        assert sorted(lines(a)) == sorted(lines(b)), mode
