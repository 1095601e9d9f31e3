"""The drop-in `kaiju` command (kaiju_amd/bin/kaiju) against the reference binary's output on
the golden reads: same option letters, same lines (all seven columns of the -v output), input order."""
import gzip
import os
import shutil
import subprocess

import pytest

import util
from kaiju_amd import build

pytestmark = pytest.mark.gpu


def run_cli(tmp_path, golden, args, name):
    out = str(tmp_path / name)
    cmd = [build.build_cli(), "-t", golden.nodes, "-f", golden.fmi, "-o", out, "-v"] + args
    subprocess.run(cmd, check=True)
    return out


def first5(path):
    """all columns of the -v output (1-3 for unclassified reads); kept under its old name"""
    rows = []
    with open(path) as f:
        for line in f:
            p = line.rstrip("\n").split("\t")
            rows.append(tuple(p[:7]) if p[0] == "C" else tuple(p[:3]))
    return rows


@pytest.mark.parametrize("mode,seg", [("mem", 1), ("mem", 0), ("greedy", 1), ("greedy", 0)])
def test_cli_single_end(gpu_lib, golden, tmp_path, mode, seg):
    args = ["-i", os.path.join(golden.dir, "reads.fq"), "-a", mode, "-z", "4"] + ([] if seg else ["-X"])
    out = run_cli(tmp_path, golden, args, "o.tsv")
    assert first5(out) == first5(os.path.join(golden.dir, f"ref_{mode}_{seg}.tsv"))


def test_cli_paired_gz_and_options(gpu_lib, golden, tmp_path):
    p1, p2 = str(tmp_path / "p1.fq.gz"), str(tmp_path / "p2.fq.gz")
    for src, dst in (("pairs_1.fq", p1), ("pairs_2.fq", p2)):
        with open(os.path.join(golden.dir, src), "rb") as fi, gzip.open(dst, "wb") as fo:
            shutil.copyfileobj(fi, fo)
    out = run_cli(tmp_path, golden, ["-i", p1, "-j", p2, "-a", "greedy"], "pe.tsv")
    assert first5(out) == first5(os.path.join(golden.dir, "ref_greedy_1_pe.tsv"))
    out = run_cli(tmp_path, golden, ["-i", os.path.join(golden.dir, "reads.fq"), "-a", "greedy", "-e", "5", "-s", "50",
                                     "-E", "10"], "e5.tsv")
    assert first5(out) == first5(os.path.join(golden.dir, "ref_greedy_e5_s50.tsv"))


def test_cli_fasta_input(gpu_lib, golden, tmp_path):
    fa = str(tmp_path / "reads.fa")
    with open(fa, "w") as f:
        for n, r in zip(golden.names, golden.reads):
            s = r.decode()
            f.write(f">{n} some description/1\n")
            for k in range(0, len(s), 60):          # wrapped sequence lines
                f.write(s[k:k + 60] + "\n")
    out = run_cli(tmp_path, golden, ["-i", fa, "-a", "mem"], "fa.tsv")
    assert first5(out) == first5(os.path.join(golden.dir, "ref_mem_1.tsv"))


def test_cli_non_verbose_uses_compact_records(gpu_lib, golden, tmp_path):
    """without -v the CLI gets 16-byte records (LCA on the device): columns 1-3 must be the reference's"""
    out = str(tmp_path / "nv.tsv")
    subprocess.run([build.build_cli(), "-t", golden.nodes, "-f", golden.fmi, "-o", out, "-i",
                    os.path.join(golden.dir, "reads.fq"), "-a", "greedy"], check=True)
    got = [tuple(l.rstrip("\n").split("\t")) for l in open(out)]
    want = [r[:3] for r in first5(os.path.join(golden.dir, "ref_greedy_1.tsv"))]
    assert got == want


def test_kaiju_multi(gpu_lib, golden, tmp_path):
    """kaiju-multi: several samples, one index load (kaiju-multi.cpp:307-334)"""
    build.build_cli()
    multi = os.path.join(os.path.dirname(build.CLI), "kaiju-multi")
    r1, p1, p2 = (os.path.join(golden.dir, f) for f in ("reads.fq", "pairs_1.fq", "pairs_2.fq"))
    o1, o2 = str(tmp_path / "a.tsv"), str(tmp_path / "b.tsv")
    subprocess.run([multi, "-t", golden.nodes, "-f", golden.fmi, "-a", "mem", "-v", "-i", f"{r1},{r1}", "-o", f"{o1},{o2}"],
                   check=True)
    ref = first5(os.path.join(golden.dir, "ref_mem_1.tsv"))
    assert first5(o1) == ref and first5(o2) == ref
    subprocess.run([multi, "-t", golden.nodes, "-f", golden.fmi, "-a", "mem", "-v", "-i", f"{p1},{p1}", "-j", f"{p2},{p2}",
                    "-o", f"{o1},{o2}"], check=True)
    ref = first5(os.path.join(golden.dir, "ref_mem_1_pe.tsv"))
    assert first5(o1) == ref and first5(o2) == ref


def test_kaijux(gpu_lib, golden, tmp_path):
    """kaijux (database sequences instead of taxa, no nodes.dmp) == the reference's kaijux lines, single and paired, with
    and without -v (Greedy here; MEM, whose match order follows the reference's maxMatches(.., 1), in
    test_gpu_zz_protein_kaijux_mem.py)"""
    mode = "greedy"
    build.build_cli()
    kaijux = os.path.join(os.path.dirname(build.CLI), "kaijux")
    for pe in (False, True):
        for v in (False, True):
            out = str(tmp_path / "x.tsv")
            cmd = [kaijux, "-f", golden.fmi, "-a", mode, "-o", out]
            cmd += ["-i", os.path.join(golden.dir, "pairs_1.fq"), "-j", os.path.join(golden.dir, "pairs_2.fq")] if pe else \
                   ["-i", os.path.join(golden.dir, "reads.fq")]
            if v:
                cmd.append("-v")
            subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
            ref = os.path.join(golden.dir, f"refx_{mode}{'_pe' if pe else ''}{'_v' if v else ''}.tsv")
            assert open(out).read() == open(ref).read(), (mode, pe, v)


def test_cli_exit_status_of_a_capacity_error(gpu_lib, golden, tmp_path):
    """a batch in which a device-side capacity bound was exceeded ends the program with status 3 (not with an abort:
    the index loader thread is joined before that return), status 0 under KAIJU_GPU_ALLOW_INEXACT"""
    # (the switch that fakes an inexact read exists in a test build of the program only: -DKAIJU_CLI_TEST_HOOKS)
    build.build_cli()
    exe = str(tmp_path / "kaiju_testhooks")
    subprocess.run(["g++", "-O1", "-std=c++17", "-DKAIJU_CLI_TEST_HOOKS", "-o", exe, build.CLI_SRC, "-L" + build.HERE, "-lkaiju_gpu", "-lz",
                    "-lpthread", "-Wl,-rpath," + build.HERE], check=True)
    cmd = [exe, "-t", golden.nodes, "-f", golden.fmi, "-o", str(tmp_path / "x.tsv"), "-i",
           os.path.join(golden.dir, "reads.fq"), "-a", "mem"]
    env = dict(os.environ, KAIJU_GPU_TEST_INEXACT="1")
    # the shipped program ignores the switch
    r = subprocess.run([build.CLI] + cmd[1:], env=env, stderr=subprocess.PIPE)
    assert r.returncode == 0
    r = subprocess.run(cmd, env=env, stderr=subprocess.PIPE)
    assert r.returncode == 3 and b"capacity bound" in r.stderr
    r = subprocess.run(cmd, env=dict(env, KAIJU_GPU_ALLOW_INEXACT="1"), stderr=subprocess.PIPE)
    assert r.returncode == 0 and b"Warning" in r.stderr
    want = [r3[:3] for r3 in first5(os.path.join(golden.dir, "ref_mem_1.tsv"))]
    assert [tuple(l.rstrip("\n").split("\t")) for l in open(tmp_path / "x.tsv")] == want


def test_cli_closed_output_pipe_is_a_failure(gpu_lib, golden, tmp_path):
    """`kaiju ... | head -1`: the write error ends the program with a non-zero status (the reference dies of SIGPIPE)"""
    big = str(tmp_path / "big.fq")
    with open(os.path.join(golden.dir, "reads.fq"), "rb") as f:
        one = f.read()
    with open(big, "wb") as f:
        for _ in range(400):                       # ~ 280 000 reads: more output than a pipe buffer holds
            f.write(one)
    cmd = [build.build_cli(), "-t", golden.nodes, "-f", golden.fmi, "-i", big, "-a", "mem"]
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    p.stdout.readline()
    p.stdout.close()
    assert p.wait(timeout=120) != 0


def test_cli_verbose_batches_in_pieces(gpu_lib, golden, tmp_path, monkeypatch):
    """-v: the text of column 7 is collected in pieces of a bounded size (one long read must not size the buffers of a whole
    batch); with a budget of a few reads per piece the lines are still the reference's, all seven columns"""
    monkeypatch.setenv("KAIJU_GPU_VERBOSE_BUDGET", "5000")
    for mode in ("mem", "greedy"):
        out = run_cli(tmp_path, golden, ["-i", os.path.join(golden.dir, "reads.fq"), "-a", mode], f"pieces_{mode}.tsv")
        assert first5(out) == first5(os.path.join(golden.dir, f"ref_{mode}_1.tsv"))
    out = run_cli(tmp_path, golden, ["-i", os.path.join(golden.dir, "pairs_1.fq"), "-j", os.path.join(golden.dir, "pairs_2.fq"), "-a", "greedy"],
                  "pieces_pe.tsv")
    assert first5(out) == first5(os.path.join(golden.dir, "ref_greedy_1_pe.tsv"))


def test_cli_blocks_over_several_gpus(gpu_lib, golden, tmp_path, monkeypatch):
    """KAIJU_GPU_DEVICES: the index is replicated (parsed and packed once, kaiju_gpu_index_load_devices), input block b goes to
    context b mod (2 x GPUs), the lines come out in input order.  On a box with one GPU the list names it twice - two replicas,
    four contexts: the plumbing of an 8-GPU node - and small blocks make sure every context gets work."""
    monkeypatch.setenv("KAIJU_GPU_DEVICES", "0,0")
    r = subprocess.run([build.build_cli(), "-t", golden.nodes, "-f", golden.fmi, "-o", str(tmp_path / "no.tsv"), "-i",
                        os.path.join(golden.dir, "reads.fq")], stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"listed twice" in r.stderr            # (a device named twice is refused unless a test asks for it)
    monkeypatch.setenv("KAIJU_GPU_DEVICES_ALLOW_REPEAT", "1")
    monkeypatch.setenv("KAIJU_GPU_BATCH", "50")
    for mode in ("mem", "greedy"):
        out = run_cli(tmp_path, golden, ["-i", os.path.join(golden.dir, "reads.fq"), "-a", mode], f"multi_{mode}.tsv")
        assert first5(out) == first5(os.path.join(golden.dir, f"ref_{mode}_1.tsv"))
    out = str(tmp_path / "multi_nv.tsv")
    subprocess.run([build.build_cli(), "-t", golden.nodes, "-f", golden.fmi, "-o", out, "-i", os.path.join(golden.dir, "reads.fq"), "-a", "greedy"],
                   check=True)
    got = [tuple(l.rstrip("\n").split("\t")) for l in open(out)]
    assert got == [r[:3] for r in first5(os.path.join(golden.dir, "ref_greedy_1.tsv"))]
