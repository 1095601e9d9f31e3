"""Ingest stages of the drop-in CLI (block reader, parallel parser) against a restatement of the reference's
reading loop (kaiju.cpp:288-386): FASTA/FASTQ, gzip, blank lines, CRLF, multi-line FASTA, missing final
newline, truncated last record, pairs.  Runs the CLI with KAIJU_GPU_PARSE_ONLY (no GPU work at all)."""
import gzip
import os
import subprocess

import numpy as np
import pytest

import util

CLI = os.path.join(util.ROOT, "kaiju_amd", "bin", "kaiju")


def ref_records(text: bytes):
    """the reference's loop on one file: list of (name, stripped sequence)"""
    lines = text.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    i, out, fastq, first = 0, [], False, True

    def strip(s):
        return bytes(c for c in s if (65 <= c <= 90) or (97 <= c <= 122))

    while True:
        while i < len(lines) and lines[i] == b"":
            i += 1
        if i >= len(lines):
            break
        line = lines[i]
        i += 1
        if first:
            fastq = line[:1] == b"@"
            first = False
        name = line[1:]
        for k, c in enumerate(name):
            if c in b" /\t\r":
                name = name[:k]
                break
        if fastq:
            seq = strip(lines[i]) if i < len(lines) else b""
            i += 3
        else:
            seq = b""
            while i < len(lines) and lines[i][:1] != b">":
                seq += strip(lines[i])
                i += 1
        out.append((name, seq))
    return out


def run_cli(tmp_path, f1, f2=None, batch=None, prescan_threads=None, piece=None, gz_threads=None, gz_piece=None):
    env = dict(os.environ, KAIJU_GPU_PARSE_ONLY="1")
    if gz_threads:                            # .gz files of any size through the several-thread inflate (csrc/host/pargz.h)
        env.update(KAIJU_GPU_GZ_MIN="0", KAIJU_GPU_GZ_THREADS=str(gz_threads), KAIJU_GPU_GZ_PIECE=str(gz_piece or 4096))
    if prescan_threads:                       # several boundary-scan threads even on these small files
        env["KAIJU_GPU_PRESCAN"] = "1"
        env["KAIJU_GPU_PRESCAN_MIN"] = "0"
        env["KAIJU_GPU_HOST_THREADS"] = str(prescan_threads)
        if piece:                             # ... and many more pieces than threads, merged while blocks are handed out
            env["KAIJU_GPU_PRESCAN_PIECE"] = str(piece)
    if batch:
        env["KAIJU_GPU_BATCH"] = str(batch)
    cmd = [CLI, "-i", str(f1)] + (["-j", str(f2)] if f2 else [])
    r = subprocess.run(cmd, env=env, capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr.decode()
    return [tuple(l.split(b"\t")) for l in r.stdout.split(b"\n") if l]


def make_fastq(rng, n, crlf=False, blanks=False, final_newline=True):
    recs = []
    for i in range(n):
        L = int(rng.integers(0, 200))
        seq = bytes(rng.choice(list(b"ACGTNacgt-*"), L).tolist())
        name = f"@read{i} extra/1".encode() if i % 3 else f"@r{i}/1".encode()
        qual = b"@" + b"I" * max(L - 1, 0) if i % 5 == 0 else b"+" * L       # quality lines that look like headers / separators
        eol = b"\r\n" if crlf else b"\n"
        recs.append(name + eol + seq + eol + b"+" + eol + qual + eol)
        if blanks and i % 7 == 0:
            recs.append(b"\n\n")
    t = b"".join(recs)
    return t if final_newline else t.rstrip(b"\n")


def make_fasta(rng, n, width=60):
    recs = []
    for i in range(n):
        L = int(rng.integers(0, 400))
        seq = bytes(rng.choice(list(b"ACGTNacgt"), L).tolist())
        recs.append(f">seq{i}\tdescr".encode() + b"\n")
        for k in range(0, L, width):
            recs.append(seq[k:k + width] + b"\n")
        if i % 11 == 0:
            recs.append(b"\n")                       # empty line inside / after a record
    return b"".join(recs)


@pytest.fixture(scope="module")
def have_cli():
    from kaiju_amd import build
    build.build()
    assert os.path.exists(CLI)


@pytest.mark.parametrize("variant", ["crlf", "blanks", "nofinal", "truncated", "plain"])
@pytest.mark.parametrize("kind", ["fastq", "fasta"])
def test_gz_piece_stream_ingest(have_cli, tmp_path, variant, kind):
    """the record walk over the pieces of a .gz file inflated by several threads (BlockReader::next_pieces: newline lists, records
    that straddle pieces) finds the records of the reference's reading loop: CRLF, blank lines, no final newline, a truncated
    last record; pieces of a few kilobytes, batches of 1 .. 1000 records"""
    rng = np.random.default_rng(31)
    if kind == "fastq":
        text = make_fastq(rng, 3000, crlf=variant == "crlf", blanks=variant == "blanks", final_newline=variant != "nofinal")
    else:
        text = make_fasta(rng, 2000)
        if variant == "crlf":
            text = text.replace(b"\n", b"\r\n")
        if variant == "nofinal":
            text = text.rstrip(b"\n")
    if variant == "truncated":
        text = text[: len(text) - 150]
    path = tmp_path / ("r.fq.gz" if kind == "fastq" else "r.fa.gz")
    with gzip.open(path, "wb") as f:
        f.write(text)
    want = ref_records(text)
    for batch, threads, piece in ((None, 3, 4096), (1, 2, 1024), (7, 5, 20000), (1000, 4, 300)):
        got = run_cli(tmp_path, path, batch=batch, gz_threads=threads, gz_piece=piece)
        assert [(g[0], g[1]) for g in got] == want, (batch, threads, piece)


@pytest.mark.parametrize("variant", ["plain", "crlf", "blanks", "nofinal", "gz", "truncated"])
def test_fastq_ingest(have_cli, tmp_path, variant):
    rng = np.random.default_rng(11)
    text = make_fastq(rng, 3000, crlf=variant == "crlf", blanks=variant == "blanks", final_newline=variant != "nofinal")
    if variant == "truncated":
        text = text[: len(text) - 150]              # the last record loses lines
    path = tmp_path / ("r.fq.gz" if variant == "gz" else "r.fq")
    if variant == "gz":
        with gzip.open(path, "wb") as f:
            f.write(text)
    else:
        path.write_bytes(text)
    want = ref_records(text)
    for batch in (None, 1, 7, 1000):
        got = run_cli(tmp_path, path, batch=batch)
        assert len(got) == len(want)
        assert [(g[0], g[1]) for g in got] == want, batch
    for threads in (2, 5, 13):
        got = run_cli(tmp_path, path, batch=97, prescan_threads=threads)
        assert [(g[0], g[1]) for g in got] == want, threads
    if variant != "gz":
        for piece in (300, 5000, 70000):
            got = run_cli(tmp_path, path, batch=97, prescan_threads=4, piece=piece)
            assert [(g[0], g[1]) for g in got] == want, piece


def test_fasta_ingest(have_cli, tmp_path):
    rng = np.random.default_rng(12)
    text = make_fasta(rng, 2000)
    path = tmp_path / "r.fa"
    path.write_bytes(text)
    want = ref_records(text)
    for batch in (None, 3, 500):
        got = run_cli(tmp_path, path, batch=batch)
        assert [(g[0], g[1]) for g in got] == want, batch
    for threads in (3, 8):
        got = run_cli(tmp_path, path, batch=50, prescan_threads=threads)
        assert [(g[0], g[1]) for g in got] == want, threads
    for piece in (200, 9000):
        got = run_cli(tmp_path, path, batch=50, prescan_threads=5, piece=piece)
        assert [(g[0], g[1]) for g in got] == want, piece
    with gzip.open(tmp_path / "r.fa.gz", "wb") as f:
        f.write(text)
    assert [(g[0], g[1]) for g in run_cli(tmp_path, tmp_path / "r.fa.gz", batch=64)] == want


def test_paired_ingest(have_cli, tmp_path):
    rng = np.random.default_rng(13)
    t1 = make_fastq(rng, 1500)
    t2 = make_fastq(np.random.default_rng(14), 1500).replace(b"/1", b"/2")
    (tmp_path / "a.fq").write_bytes(t1)
    (tmp_path / "b.fq").write_bytes(t2)
    w1, w2 = ref_records(t1), ref_records(t2)
    got = run_cli(tmp_path, tmp_path / "a.fq", tmp_path / "b.fq", batch=97)
    assert [(g[0], g[1], g[2]) for g in got] == [(a[0], a[1], b[1]) for a, b in zip(w1, w2)]
    got = run_cli(tmp_path, tmp_path / "a.fq", tmp_path / "b.fq", batch=97, prescan_threads=6)
    assert [(g[0], g[1], g[2]) for g in got] == [(a[0], a[1], b[1]) for a, b in zip(w1, w2)]


def test_multi_file_lists(have_cli, tmp_path):
    """kaiju-multi: comma separated lists of inputs / outputs (kaiju-multi.cpp:221-334), one pass per sample"""
    rng = np.random.default_rng(21)
    texts = [make_fastq(rng, 300), make_fasta(rng, 200), make_fastq(rng, 50)]
    ins, outs = [], []
    for i, t in enumerate(texts):
        f = tmp_path / f"s{i}.txt"
        f.write_bytes(t)
        ins.append(str(f))
        outs.append(str(tmp_path / f"o{i}.tsv"))
    env = dict(os.environ, KAIJU_GPU_PARSE_ONLY="1", KAIJU_GPU_BATCH="64")
    multi = os.path.join(os.path.dirname(CLI), "kaiju-multi")
    r = subprocess.run([multi, "-i", ",".join(ins), "-o", ",".join(outs)], env=env, capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr.decode()
    for t, o in zip(texts, outs):
        got = [tuple(l.split(b"\t")) for l in open(o, "rb").read().split(b"\n") if l]
        assert [(g[0], g[1]) for g in got] == ref_records(t)
    # list lengths must agree
    r = subprocess.run([multi, "-i", ",".join(ins), "-o", outs[0]], env=env, capture_output=True, timeout=120)
    assert r.returncode != 0 and b"Length of input/output file lists differs" in r.stderr


def test_prescan_false_start_is_caught(have_cli, tmp_path):
    """a quality line that starts with '@' in front of a record whose sequence line starts with '+': the boundary guess of a
    scan thread can land on it; the walk of the thread before does not arrive there, everything is redone sequentially"""
    rng = np.random.default_rng(31)
    recs = []
    for i in range(4000):
        L = int(rng.integers(20, 120))
        seq = bytes(rng.choice(list(b"ACGT"), L).tolist())
        if i % 2:
            seq = b"+" + seq[1:]                       # (strip() drops the '+')
        recs.append(b"@r%d\n" % i + seq + b"\n+\n" + b"@" + b"I" * (L - 1) + b"\n")
    text = b"".join(recs)
    path = tmp_path / "adv.fq"
    path.write_bytes(text)
    want = ref_records(text)
    for threads in (2, 7, 16):
        got = run_cli(tmp_path, path, batch=333, prescan_threads=threads)
        assert [(g[0], g[1]) for g in got] == want, threads
    for piece in (150, 4096):                  # (a wrong guess in the middle: the rest is walked sequentially)
        got = run_cli(tmp_path, path, batch=333, prescan_threads=6, piece=piece)
        assert [(g[0], g[1]) for g in got] == want, piece


def test_kaijup_names_and_u_line_decision(tmp_path):
    """kaijup keeps the whole header line as the name (kaijup.cpp:249-262) and prints "U<TAB>name<TAB>0" exactly for reads
    shorter than -m or without any fragment (ConsumerThreadp.cpp:17-21,67-71): the parse-only dump of the kaijup
    personality carries that decision in a fourth column; compared with the reference's kaijup lines"""
    kaijup = os.path.join(util.ROOT, "kaiju_amd", "bin", "kaijup")
    names, reads = util.read_fasta(os.path.join(util.GOLD, "prot.fa"), keep_names=True)
    for mode in ("mem", "greedy"):
        r = subprocess.run([kaijup, "-f", "-", "-i", os.path.join(util.GOLD, "prot.fa"), "-a", mode],
                           env=dict(os.environ, KAIJU_GPU_PARSE_ONLY="1"), capture_output=True, check=True)
        got = [line.split(b"\t") for line in r.stdout.split(b"\n") if line]
        assert [g[0].decode() for g in got] == names and [g[1] for g in got] == reads
        ref = {}
        with open(os.path.join(util.GOLD, f"refpx_{mode}.tsv")) as f:
            for line in f:
                q = line.rstrip("\n").split("\t")
                ref[q[1]] = q
        n3 = 0
        for g, nm, s in zip(got, names, reads):
            q = ref[nm]
            if q[0] == "U":
                three = len(q) == 3
                n3 += three
                assert three == (len(s) < 11 or g[3] == b"0"), (mode, nm, q, g[3])
            else:
                assert g[3] == b"1"
        assert n3 > 20
    # the kaiju personality cuts the names and refuses -j with -p
    r = subprocess.run([CLI, "-p", "-t", "-", "-f", "-", "-i", os.path.join(util.GOLD, "prot.fa")],
                       env=dict(os.environ, KAIJU_GPU_PARSE_ONLY="1"), capture_output=True, check=True)
    assert [line.split(b"\t")[0].decode() for line in r.stdout.split(b"\n") if line] == util.read_fasta(os.path.join(util.GOLD, "prot.fa"))[0]
    r = subprocess.run([CLI, "-p", "-t", "-", "-f", "-", "-i", os.path.join(util.GOLD, "prot.fa"), "-j", os.path.join(util.GOLD, "prot.fa")],
                       env=dict(os.environ, KAIJU_GPU_PARSE_ONLY="1"), capture_output=True)
    assert r.returncode != 0 and b"Protein input only supports one input file" in r.stderr


def test_device_list_is_validated(tmp_path):
    """KAIJU_GPU_DEVICES: every entry a device number, none twice - "a,b" or "0,,1" must not quietly become GPU 0 more than once
    (checked before anything is loaded; with KAIJU_GPU_PARSE_ONLY no GPU is asked how many devices exist)"""
    fq = tmp_path / "r.fq"
    fq.write_bytes(b"@r0\nACGTACGTACGTACGTACGTACGTACGTACGTACGT\n+\nIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIII\n")
    base = [CLI, "-t", "/dev/null", "-f", "/dev/null", "-i", str(fq)]
    for bad, msg in (("a,b", b"not a list"), ("0,,1", b"not a list"), ("0,1,", b"not a list"), ("0,1,0", b"listed twice"), ("-1", b"no such device")):
        r = subprocess.run(base, env=dict(os.environ, KAIJU_GPU_PARSE_ONLY="1", KAIJU_GPU_DEVICES=bad), capture_output=True)
        assert r.returncode != 0 and msg in r.stderr, (bad, r.stderr)
    r = subprocess.run(base, env=dict(os.environ, KAIJU_GPU_PARSE_ONLY="1", KAIJU_GPU_DEVICES="0,3,1"), capture_output=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run(base, env=dict(os.environ, KAIJU_GPU_PARSE_ONLY="1", KAIJU_GPU_DEVICES="0,0", KAIJU_GPU_DEVICES_ALLOW_REPEAT="1"), capture_output=True)
    assert r.returncode == 0, r.stderr


@pytest.mark.parametrize("kind", ["level1", "level9", "members", "small_members", "syncflush", "stored", "fixed", "fasta"])
def test_gzip_inflated_by_several_threads(have_cli, tmp_path, kind):
    """.gz input of some size is inflated by several threads (csrc/host/pargz.h: every thread enters the deflate stream at a
    block start it searched for, with markers for the 32 KB of text it cannot know; CRC-32 and length of every member are
    checked): the records are those of zlib's gzread (KAIJU_GPU_GZ_THREADS=1) - compression levels, several members, members
    of 64 KB (bgzip-like), sync-flush points (pigz), stored blocks, fixed Huffman codes (no block start to find: one thread),
    FASTA; pieces of 256 KB so that a few megabytes make dozens of them"""
    import zlib
    rng = np.random.default_rng(9)
    n = 60000
    if kind == "fasta":
        recs = [b">s%d d\n%s\n" % (i, bytes(rng.choice(list(b"ACGT"), int(rng.integers(50, 400))).tolist())) for i in range(n)]
    else:
        recs = []
        for i in range(n):
            L = int(rng.integers(30, 151))
            recs.append(b"@read%d/1\n%s\n+\n%s\n" % (i, bytes(rng.choice(list(b"ACGTN"), L).tolist()), bytes(rng.integers(35, 74, L).astype(np.uint8).tolist())))
    text = b"".join(recs)

    def zc(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
        co = zlib.compressobj(level, zlib.DEFLATED, 31, 8, strategy)
        return co.compress(data) + co.flush()
    if kind == "level1":
        blob = zc(text, 1)
    elif kind == "level9":
        blob = zc(text, 9)
    elif kind == "members":
        blob = b"".join(zc(text[i:i + 3_000_000]) for i in range(0, len(text), 3_000_000))
    elif kind == "small_members":
        blob = b"".join(zc(text[i:i + 65280]) for i in range(0, len(text), 65280))
    elif kind == "syncflush":
        co = zlib.compressobj(6, zlib.DEFLATED, 31)
        parts = []
        for i in range(0, len(text), 131072):
            parts += [co.compress(text[i:i + 131072]), co.flush(zlib.Z_SYNC_FLUSH)]
        blob = b"".join(parts) + co.flush()
    elif kind == "stored":
        blob = zc(text, 0)
    elif kind == "fixed":
        blob = zc(text, 6, zlib.Z_FIXED)
    else:
        blob = zc(text)
    assert len(blob) > (2 << 20)                       # (smaller files keep gzread)
    path = tmp_path / ("r.fa.gz" if kind == "fasta" else "r.fq.gz")
    path.write_bytes(blob)
    out = {}
    for threads in ("1", "5"):
        env = dict(os.environ, KAIJU_GPU_PARSE_ONLY="1", KAIJU_GPU_GZ_THREADS=threads, KAIJU_GPU_GZ_PIECE=str(256 << 10))
        r = subprocess.run([CLI, "-i", str(path)], env=env, capture_output=True, timeout=300)
        assert r.returncode == 0, r.stderr.decode()[-400:]
        out[threads] = r.stdout
    assert out["1"] == out["5"]
    assert out["5"].count(b"\n") == n


def test_damaged_gzip_is_an_error(have_cli, tmp_path):
    """a byte flipped in the middle of a .gz file: the several-thread inflate ends with an error (a block that does not decode,
    or the CRC-32 of the member), as gzread does - never with silently different reads"""
    import zlib
    rng = np.random.default_rng(10)
    text = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, bytes(rng.choice(list(b"ACGT"), 150).tolist()), b"I" * 150) for i in range(60000))
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    blob = bytearray(co.compress(text) + co.flush())
    assert len(blob) > (2 << 20)
    blob[len(blob) // 2] ^= 0x10
    path = tmp_path / "bad.fq.gz"
    path.write_bytes(bytes(blob))
    for threads in ("1", "5"):
        env = dict(os.environ, KAIJU_GPU_PARSE_ONLY="1", KAIJU_GPU_GZ_THREADS=threads, KAIJU_GPU_GZ_PIECE=str(256 << 10))
        r = subprocess.run([CLI, "-i", str(path)], env=env, capture_output=True, timeout=300)
        assert r.returncode != 0 and (b"gzip" in r.stderr or b"Error while reading" in r.stderr), (threads, r.stderr.decode()[-300:])
