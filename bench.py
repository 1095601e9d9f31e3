#!/usr/bin/env python3
"""Benchmark of the MI355X Kaiju classification path.

    python bench.py --gpus N --steps K --warmup W

One *step* is one pass of the hot path (six-frame translation -> SEG -> MEM search on the protein
FM-index -> locate / taxon ids -> LCA, kaiju_gpu_classify_batch_device + kaiju_gpu_lca_batch_device)
over one batch of synthetic reads that is already resident in HBM.  The headline workload is
BASELINE.json configs[1]: a viruses-like synthetic index (680 001 proteins, 190 M aa)
and 10 M synthetic 150-bp reads per GPU, `-a mem -m 11`, SEG on (the reference's default).

With N > 1 and no launcher in the environment the script re-executes itself under
`python -m torch.distributed.run --nproc-per-node N` (one rank per GPU, RCCL); every rank holds a
replica of the index and classifies its own 10 M reads (weak scaling); the per-read 16-byte records are
collected on rank 0 with one asynchronous gather per chunk.  Rank 0 prints ONE JSON line.

Besides the headline (`value`, MEM) the line carries further legs, each timed the same way with fewer
steps: `greedy` (BASELINE configs[2], the reference's default mode), `paired` (2 x 150-bp pairs, MEM:
the shape of configs[3] on this index), and at N = 1 `host_buffers` (kaiju_gpu_classify_batch_compact
from page-locked host memory: the PCIe-inclusive rate, never `value`).

`roofline` (per leg) is that of the dominant search kernel.  `achieved` = algorithmic bytes of THIS
implementation per launch / the kernel's launch duration (HIP events on its stream), where the
algorithmic bytes are counted on the device by the counting instantiation of the same lane in one extra,
untimed launch (kaiju_gpu_set_count_ops): 128 B for every k-mer table lookup, every distinct rank-block
line of an UpdateSI / LF step and every SA sample, 64 B per peptide window, 16 B per read / fragment
descriptor, 184 B per hit record (+ queue items and match records for Greedy).  `work_rate` is the
reference-equivalent figure of SURVEY.md 8(d) (128 B per UpdateSI the REFERENCE performs ...): a rate of
work, not traffic, so it is not divided by the HBM peak.  `traffic` is the measured HBM traffic of one
launch from committed rocprofv3 PMC passes (profiles/traffic.json names the raw CSVs), only when
the workload matches.

`cpu_baseline` (N = 1) times the unmodified reference binary (oracle/_ref/kaiju -z <cores>) on a bounded
sample of the same reads; its output lines are then compared with the GPU's records for the same reads
(`parity`: reads checked / mismatches, for all three legs).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from kaiju_amd import api, dist as kdist, mkfmi, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X spec (MI355X_MICROARCH.md)
HIT_BYTES = 184
COMPACT_BYTES = 16


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def fastq_bytes(reads: np.ndarray, first: int = 0) -> bytes:
    """fixed-width FASTQ records, fully vectorised; read i is named r<first + i, 8 digits>"""
    n, L = reads.shape
    names = np.char.add("@r", np.char.zfill(np.arange(first, first + n).astype(str), 8)).astype("S10")
    rec = np.empty((n, 10 + 1 + L + 3 + L + 1), dtype=np.uint8)
    rec[:, :10] = np.frombuffer(names.tobytes(), dtype=np.uint8).reshape(n, 10)
    rec[:, 10] = 10
    rec[:, 11:11 + L] = reads
    rec[:, 11 + L:14 + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    rec[:, 14 + L:14 + 2 * L] = ord("I")
    rec[:, 14 + 2 * L] = 10
    return rec.tobytes()


def offsets(m: int, L: int, Lm: int, paired: bool) -> np.ndarray:
    """off[2m+1] of m fixed-width reads (pairs: mate 1 then mate 2 in one row of L = 2 * Lm bytes)"""
    o = np.empty(2 * m + 1, dtype=np.uint64)
    o[0::2] = np.arange(m + 1, dtype=np.uint64) * np.uint64(L)
    o[1::2] = o[2::2] if not paired else o[0:-1:2] + np.uint64(Lm)
    return o


# ----------------------------------------------------------------------------------------------
# accounting
# ----------------------------------------------------------------------------------------------
def algorithmic_bytes(oc: dict, nseq: int) -> float:
    """bytes one launch of the search kernel has to move, by the memory steps it performed"""
    lines = oc["kmer_lookups"] + oc["update_si_lines"] + oc["lf_lines"] + oc["sa_samples"] + oc["items_read"] + oc["items_written"]
    small = 16 * (oc["read_meta"] + oc["frag_desc"] + oc["matches_read"] + oc["matches_written"] + 2 * oc["si_spills"])
    term = 8 * math.ceil(math.log2(max(nseq, 2))) * oc["term_searches"]
    return 128.0 * lines + small + 64.0 * oc["window_fills"] + float(HIT_BYTES) * oc["hits"] + term + oc.get("record_bytes", 0)


def reference_ops(W, fmi, reads_sample, mode, seg, paired, Lm):
    """UpdateSI / LF / SA-decode counts of the REFERENCE's algorithm per read, from the instrumented oracle
    (test infrastructure used for accounting only, never on the measured path)"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    O = po.Oracle()
    oix = O.load_fmi(fmi)
    k = len(reads_sample)
    seqs = np.ascontiguousarray(reads_sample).reshape(-1)
    off = offsets(k, reads_sample.shape[1], Lm, paired)
    O.counters(reset=True)
    t0 = time.time()
    O.classify(oix, None, O.params(mode, seg=seg), seqs, off, paired=paired)
    t = time.time() - t0
    c = O.counters(reset=True)
    return {"update_si": c["update_si"] / k, "lf_steps": c["fmindex_current"] / k, "sa_decodes": c["sa_decode"] / k,
            "sample": k, "oracle_reads_per_s": k / max(t, 1e-9)}


def run_reference(W, fmi, nodes, reads, Lm, paired, mode, seg, sample, tag_suffix="", extra=()):
    """the unmodified reference on `sample` of the reads: (baseline dict, per-read (classified, taxon) arrays)"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    if not po.have_ref():
        return None, None
    import pandas as pd
    cores = os.cpu_count() or 1
    s = min(sample, len(reads))
    tag = f"{mode}{'_pe' if paired else ''}{tag_suffix}"
    files = [f"{W}/cpu_{tag}_1.fq"] + ([f"{W}/cpu_{tag}_2.fq"] if paired else [])
    one = [f"{W}/cpu_{tag}_one_1.fq"] + ([f"{W}/cpu_{tag}_one_2.fq"] if paired else [])
    for k, (fn, fo) in enumerate(zip(files, one)):
        part = reads[:s, k * Lm:(k + 1) * Lm]
        with open(fn, "wb") as f:
            f.write(fastq_bytes(part))
        with open(fo, "wb") as f:
            f.write(fastq_bytes(part[:1]))
    out = f"{W}/cpu_{tag}_out.tsv"
    base = [po.REF_KAIJU, "-t", nodes, "-f", fmi, "-a", mode, "-z", str(cores), "-o", out] + list(extra)
    if not seg:
        base.append("-X")

    def inputs(fl):
        return ["-i", fl[0]] + (["-j", fl[1]] if paired else [])
    outv = f"{W}/cpu_{tag}_out_v.tsv"
    basev = [x if x != out else outv for x in base] + ["-v"]
    single_run = os.path.getsize(fmi) > 8e9 or os.environ.get("KAIJU_BENCH_SINGLE_REF_RUN") == "1"
    if single_run:
        # a refseq-class .fmi takes the reference a minute to read: ONE run serves as baseline and parity source.  With -v the
        # program says "Start classification" on stderr when the index is loaded (kaiju.cpp:286); the time from that line to
        # the end of the process is the classification phase (output with the -v columns included)
        t0 = time.time()
        pr = subprocess.Popen(basev + inputs(files), stderr=subprocess.PIPE, text=True)
        t_start = None
        for line in pr.stderr:
            if "Start classification" in line and t_start is None:
                t_start = time.time()
        if pr.wait() != 0 or t_start is None:
            raise RuntimeError("the reference binary failed")
        t_end = time.time()
        t_load, t_cls = t_start - t0, max(t_end - t_start, 1e-6)
        bl = {"value": s / t_cls, "unit": "reads/s" if not paired else "pairs/s", "cores": cores, "kind": "reference",
              "sample": f"{s} of the benchmark {'pairs' if paired else 'reads'}, kaiju -z {cores} -a {mode}{'' if seg else ' -X'}{' ' + ' '.join(extra) if extra else ''} -v; "
                        f"one run: {t_cls:.1f}s from its 'Start classification' line to its end (index load {t_load:.1f}s before it)"}
        import shutil
        shutil.copyfile(outv, out)          # (the consistency check below then compares the run with itself)
    else:
        t0 = time.time()
        subprocess.run(base + inputs(one), check=True, stderr=subprocess.DEVNULL)
        t_load = time.time() - t0
        t0 = time.time()
        subprocess.run(base + inputs(files), check=True, stderr=subprocess.DEVNULL)
        t_all = time.time() - t0
        t_cls = max(t_all - t_load, 1e-6)
        bl = {"value": s / t_cls, "unit": "reads/s" if not paired else "pairs/s", "cores": cores, "kind": "reference",
              "sample": f"{s} of the benchmark {'pairs' if paired else 'reads'}, kaiju -z {cores} -a {mode}{'' if seg else ' -X'}{' ' + ' '.join(extra) if extra else ''}; "
                        f"wall {t_all:.1f}s minus index load {t_load:.1f}s"}
        # parity lines: the same reads once more with -v (column 4 = match length / score of the best match,
        # ConsumerThread.cpp:724-739 + extraoutput), untimed - the baseline above is the reference's plain run
        subprocess.run(basev + inputs(files), check=True, stderr=subprocess.DEVNULL)
    df = pd.read_csv(outv, sep="\t", header=None, names=["c", "name", "tax", "best"],
                     dtype={"c": str, "name": str, "tax": np.uint64, "best": np.float64}, usecols=[0, 1, 2, 3])
    idx = df["name"].str.slice(1).astype(np.int64).to_numpy()
    cls = np.zeros(s, dtype=np.uint8)
    tax = np.zeros(s, dtype=np.uint64)
    best = np.zeros(s, dtype=np.int64)
    seen = np.zeros(s, dtype=np.uint8)
    cls[idx] = (df["c"].to_numpy() == "C").astype(np.uint8)
    tax[idx] = df["tax"].to_numpy()
    best[idx] = np.nan_to_num(df["best"].to_numpy(), nan=0.0).astype(np.int64)
    seen[idx] = 1
    if not seen.all():
        raise RuntimeError("the reference printed fewer lines than reads")
    # the plain run's lines must say the same as the -v run's first three columns
    dp = pd.read_csv(out, sep="\t", header=None, names=["c", "name", "tax"], dtype={"c": str, "name": str, "tax": np.uint64},
                     usecols=[0, 1, 2])
    ip = dp["name"].str.slice(1).astype(np.int64).to_numpy()
    if len(ip) != s or not (np.array_equal(cls[ip], (dp["c"].to_numpy() == "C").astype(np.uint8)) and
                            np.array_equal(tax[ip], dp["tax"].to_numpy())):
        raise RuntimeError("the reference's -v run disagrees with its plain run")
    return bl, (cls, tax, best)


def compare_with_reference(cls, tax, best, ref):
    """records of the device (finalised C/U, taxon, best) against the reference's lines (run_reference) for the same reads"""
    k = len(ref[0])
    bad = np.nonzero((cls[:k] != ref[0]) | (tax[:k] != ref[1]) | ((ref[0] != 0) & (best[:k] != ref[2])))[0]
    return {"checked": int(k), "mismatches": int(len(bad)), "first_mismatches": [int(x) for x in bad[:5]]}


def big_database(W, nseq, rank):
    """refseq-class databases (>= 2 M proteins): tests/tools/gen_db.c - the recipe of synth.make_db on all host cores - writes
    the FASTA and the code / offset / taxon arrays under W; every rank maps them (reads are drawn from the database)"""
    base = f"{W}/db_{nseq}"
    if rank == 0 and not (os.path.exists(base + ".taxids") and os.path.exists(base + ".codes")):
        import tempfile
        exe = os.path.join(tempfile.gettempdir(), f"kaiju_gen_db_{os.getpid()}")       # (not under W: /dev/shm may be mounted noexec)
        subprocess.run(["gcc", "-O2", "-fopenmp", "-o", exe, os.path.join(ROOT, "tests", "tools", "gen_db.c"), "-lm"], check=True)
        subprocess.run([exe, str(nseq), "20260926", base + ".faa", base + ".codes", base + ".offsets", base + ".taxids.tmp"], check=True)
        os.replace(base + ".taxids.tmp", base + ".taxids")
    kdist.barrier()
    return synth.SynthDB(codes=np.memmap(base + ".codes", dtype=np.uint8, mode="r"), offsets=np.fromfile(base + ".offsets", dtype=np.int64),
                         taxids=np.fromfile(base + ".taxids", dtype=np.int64), names=None)


READ_BLOCK = 1 << 20


def reads_of_range(db, lo, hi, paired, seed):
    """reads [lo, hi) of the ONE workload whose block b (READ_BLOCK reads) is generated with seed + b: what a rank of a
    strong-scaling job classifies does not depend on the number of ranks"""
    parts = []
    for b in range(lo // READ_BLOCK, (hi + READ_BLOCK - 1) // READ_BLOCK):
        b0 = b * READ_BLOCK
        if paired:
            m1, m2 = synth.make_pairs(db, READ_BLOCK, seed=seed + b)
            blk = np.concatenate([m1, m2], axis=1)
        else:
            blk = synth.make_reads(db, READ_BLOCK, seed=seed + b)
        parts.append(blk[max(lo, b0) - b0: min(hi, b0 + READ_BLOCK) - b0])
    return np.ascontiguousarray(np.concatenate(parts, axis=0))


def vm_hwm_bytes():
    try:
        with open("/proc/self/status") as f:
            for line in f:
                if line.startswith("VmHWM:"):
                    return int(line.split()[1]) * 1024
    except Exception:  # noqa: BLE001
        pass
    return None


def image_is_fresh(img, fmi):
    """an image belongs to this .fmi when its header remembers the .fmi's size (time stamps survive cp -p / rsync -t, and an
    image of another format version must be rebuilt, not loaded)"""
    if not os.path.exists(img):
        return False
    import ctypes
    L = api.lib()
    L.kaiju_gpu_index_image_source_bytes.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64)]
    v = ctypes.c_uint64(0)
    return L.kaiju_gpu_index_image_source_bytes(img.encode(), ctypes.byref(v)) == 0 and int(v.value) == os.path.getsize(fmi)


# ----------------------------------------------------------------------------------------------
# one leg = one workload timed like the headline
# ----------------------------------------------------------------------------------------------
class Leg:
    comm = None          # --gather lib: this rank's kaiju_gpu_comm (the library's own RCCL gather instead of torch.distributed's)

    def __init__(self, name, mode, paired, reads, Lm, index, dtax, dev, rank, world, seg, chunk, nctx, protein=False):
        import torch
        self.torch = torch
        self.name, self.mode, self.paired, self.reads, self.Lm = name, mode, paired, reads, Lm
        self.index, self.dtax, self.dev, self.rank, self.world = index, dtax, dev, rank, world
        self.n, self.L = reads.shape
        self.protein = protein
        self.params = api.default_params(mode, seg=seg, input_is_protein=1 if protein else 0)
        # chunk 0 / nctx 0 = chosen here: a MEM leg of 4 M reads (pairs) and more runs as TWO chunks on two contexts - while one
        # chunk is in k_mem the other's stage 1 and post-search kernels run next to it (profiles/r06_l31: 395.8 -> 426.2 M
        # reads/s, pairs 216.0 -> 229.7; 2 M-read legs + 1.5 %, Greedy 61.5 -> 55.2: its persistent kernel leaves no room) -,
        # everything else as ONE launch on one context
        if nctx <= 0:
            nctx = 2 if (mode == "mem" and self.n >= 4_000_000 and chunk <= 0) else 1
        if chunk <= 0:
            chunk = (self.n + nctx - 1) // nctx
        if os.environ.get("KAIJU_BENCH_SERIAL"):      # (PMC passes: the launches of the default line, one after the other on one context)
            nctx = 1
        self.nctx = max(1, nctx)
        self.clfs = [api.Classifier(index, self.params) for _ in range(self.nctx)]
        for c in self.clfs:
            c.set_max_read_length(Lm)
        self.d_seqs = torch.from_numpy(reads.reshape(-1)).to(dev)
        self.chunk = min(chunk, self.n)
        self.bounds = [(lo, min(self.n, lo + self.chunk)) for lo in range(0, self.n, self.chunk)]
        self.d_offs = [torch.from_numpy(offsets(hi - lo, self.L, Lm, paired).view(np.int64)).to(dev) for lo, hi in self.bounds]
        self.d_out = torch.zeros(self.n * HIT_BYTES, dtype=torch.uint8, device=dev)
        # what leaves the GPU: 16-byte records (LCA computed on the device), not the 184-byte id lists
        self.d_compact = torch.zeros(self.n * COMPACT_BYTES, dtype=torch.uint8, device=dev)
        # one HIP stream per context: kernels, the LCA and the gather of a chunk queue on the same stream (the contexts' own
        # streams are wrapped for torch: its collectives must be ordered behind the kernels).  Which pairs of streams the
        # runtime maps to different hardware queues decides how much of stage 1 / SEG of one chunk hides behind the tail of
        # the other chunk's search kernel; measured on one box: ctx 139.9, torch 134.1, prio 138.1, mixed 146.6 M reads/s
        kind = os.environ.get("KAIJU_BENCH_STREAMS", "mixed")
        if kind == "torch":
            self.streams = [torch.cuda.Stream(dev) for _ in range(self.nctx)]
        elif kind == "prio":
            self.streams = [torch.cuda.Stream(dev, priority=-(k % 2)) for k in range(self.nctx)]
        elif kind == "mixed":
            self.streams = [torch.cuda.ExternalStream(self.clfs[0].stream_handle(), device=dev)] + [torch.cuda.Stream(dev) for _ in range(self.nctx - 1)]
        else:
            self.streams = [torch.cuda.ExternalStream(c.stream_handle(), device=dev) for c in self.clfs]
        self.kern_ms, self.stage_ms, self.retries, self.gathered = [], {"translate": 0.0, "seg": 0.0, "search": 0.0, "post_search": 0.0}, 0, 0

    def _collect(self, c, m, record):
        st = c.stats()                  # HIP events of that context's last chunk (blocks until its kernels are done)
        if st.error_flags:
            raise SystemExit(f"device-side capacity error flags {st.error_flags}")
        if record:
            self.kern_ms.append((st.ms_search, m))
            for k, v in (("translate", st.ms_translate), ("seg", st.ms_seg), ("search", st.ms_search), ("post_search", st.ms_retry)):
                self.stage_ms[k] += v
            self.retries += st.n_overflow_retries

    def step(self, record, nctx=None, only_chunk=None):
        """Classification contexts ping-pong the chunks on their HIP streams (the "two host threads per GPU on separate
        streams" of SURVEY.md 8b): while one chunk is in its search kernel the next one runs stage 1 and the SEG pass."""
        torch = self.torch
        nctx = nctx or self.nctx
        g = kdist.LibGatherer(Leg.comm) if Leg.comm is not None else kdist.HitGatherer(self.world, self.rank, keep_results=False)
        pending = [None] * nctx
        main = torch.cuda.current_stream(self.dev)
        for s in self.streams:
            s.wait_stream(main)
        for k, ((lo, hi), d_off) in enumerate(zip(self.bounds, self.d_offs)):
            if only_chunk is not None and k != only_chunk:
                continue
            c, s = self.clfs[k % nctx], self.streams[k % nctx]
            if pending[k % nctx] is not None:
                self._collect(c, pending[k % nctx], record)
            m = hi - lo
            out_view = self.d_out[lo * HIT_BYTES: hi * HIT_BYTES]
            cview = self.d_compact[lo * COMPACT_BYTES: hi * COMPACT_BYTES]
            with torch.cuda.stream(s):
                if os.environ.get("KAIJU_BENCH_TWO_CALLS"):       # (A/B: the two entry points of rounds 1-5)
                    c.classify_device(self.d_seqs.data_ptr() + lo * self.L, m * self.L, d_off.data_ptr(), m, out_view.data_ptr(),
                                      paired=self.paired, stream=s.cuda_stream)
                    c.lca_device(self.dtax, out_view.data_ptr(), m, cview.data_ptr(), stream=s.cuda_stream)
                else:
                    c.classify_device_compact(self.dtax, self.d_seqs.data_ptr() + lo * self.L, m * self.L, d_off.data_ptr(), m,
                                              out_view.data_ptr(), cview.data_ptr(), paired=self.paired, stream=s.cuda_stream)
                g.gather(cview)
            pending[k % nctx] = m
        for k in range(nctx):
            if pending[k] is not None:
                self._collect(self.clfs[k], pending[k], record)
        for s in self.streams:
            main.wait_stream(s)
        g.wait()
        self.gathered += g.bytes_gathered

    def run(self, steps, warmup):
        torch = self.torch
        for _ in range(warmup):
            self.step(False)
        kdist.barrier()
        torch.cuda.synchronize(self.dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step(True)
        kdist.barrier()
        torch.cuda.synchronize(self.dev)
        mine = time.perf_counter() - t0
        self.elapsed_rank = mine
        self.elapsed = kdist.max_over_ranks(mine, device=self.dev)
        self.steps = steps
        self.gathered_timed = self.gathered
        # one more, untimed, pass with the chunks strictly one after the other: kernel durations free of the other
        # chunk's kernels (what rocprofv3 --kernel-trace shows for `--contexts 1`)
        live = (list(self.kern_ms), dict(self.stage_ms))
        passes = []
        for _ in range(2):                 # (twice, the smaller figure of every stage: one pass now and then sits behind a hiccup)
            self.kern_ms, self.stage_ms = [], {k: 0.0 for k in self.stage_ms}
            self.step(True, 1)
            torch.cuda.synchronize(self.dev)
            passes.append((self.kern_ms, self.stage_ms))
        self.excl_kern = [min(a, b) for a, b in zip(passes[0][0], passes[1][0])]
        self.excl_stage = {k: min(passes[0][1][k], passes[1][1][k]) for k in passes[0][1]}
        self.kern_ms, self.stage_ms = live
        # the records the TIMED kernels wrote (the pass above runs exactly the launches of a timed step): the parity leg
        # reads this snapshot, never what the counting pass below leaves in d_compact
        self.timed_compact = self.d_compact.clone()
        # accounting: the counting instantiation of the lane on the first chunk (untimed)
        c = self.clfs[0]
        c.count_ops(True)
        self.step(False, 1, only_chunk=0)
        torch.cuda.synchronize(self.dev)
        self.op_counts = c.op_counts()
        c.count_ops(False)
        # the counting lane is another instantiation of the same source: its records must equal the timed lanes'
        first = (self.bounds[0][1] - self.bounds[0][0]) * COMPACT_BYTES
        self.count_equals_timed = bool(torch.equal(self.timed_compact[:first], self.d_compact[:first]))
        if not self.count_equals_timed:
            raise SystemExit(f"leg {self.name}: the counting instantiation wrote records that differ from the timed kernels'")
        return self

    def result(self, world, ref_ops, traffic, nseq):
        tot_units = self.n * world * self.steps
        value = tot_units / self.elapsed
        excl_ms = sum(ms for ms, _ in self.excl_kern) / max(len(self.excl_kern), 1)
        live_ms = sum(ms for ms, _ in self.kern_ms) / max(len(self.kern_ms), 1)
        per_launch = float(sum(m for _, m in self.excl_kern)) / max(len(self.excl_kern), 1)
        first = self.bounds[0][1] - self.bounds[0][0]
        oc = self.op_counts
        alg = algorithmic_bytes(oc, nseq) * (per_launch / first)          # counted on chunk 0, scaled to the mean launch
        achieved = alg / (excl_ms * 1e-3) / 1e9 if excl_ms > 0 else 0.0
        wide_ix = self.index.info.bwtlen >= 2 ** 32
        roof = {"bound": "hbm", "kernel": ("k_mem_wide2" if wide_ix else "k_mem") if self.mode == "mem" else ("k_greedy2_wide" if wide_ix else ("k_greedy3" if os.environ.get("KAIJU_GPU_GREEDY_LANE") == "v3" else "k_greedy2")),
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                **({"traffic_note": _TRAFFIC_NOTES[(self.mode, bool(self.paired))]}
                   if _TRAFFIC_NOTES.get((self.mode, bool(self.paired))) else {}),
                "algorithmic_bytes_per_launch": alg, "units_per_launch": per_launch,
                "algorithmic_bytes_per_unit": alg / max(per_launch, 1.0),
                "avg_launch_ms": excl_ms,
                "avg_launch_ms_note": ("HIP events around the kernel, chunks strictly one after the other (two extra untimed passes, the smaller figure; "
                                       "rocprofv3 --kernel-trace of `bench.py --contexts 1 --chunk %d` shows the same average)" % self.chunk)
                                      + ("; live_avg_launch_ms = the same launches in the timed steps, where the two contexts' kernels run next to "
                                         "each other and a launch lasts longer than it would alone" if self.nctx > 1 and len(self.bounds) > 1 else ""),
                "live_avg_launch_ms": live_ms,
                "ops_per_unit": {k: v / first for k, v in oc.items()},
                "stage_ms_per_step_exclusive": dict(self.excl_stage),
                "stage_ms_per_step_live": {k: v / max(self.steps, 1) for k, v in self.stage_ms.items()}}
        if ref_ops is not None:
            ref_bytes = 128.0 * ref_ops["update_si"] + 64.0 * ref_ops["lf_steps"] + 8.0 * ref_ops["sa_decodes"] + self.L + HIT_BYTES
            roof["work_rate"] = {"GBps_reference_equivalent": ref_bytes * per_launch / (excl_ms * 1e-3) / 1e9 if excl_ms > 0 else 0.0,
                                 "reference_bytes_per_unit": ref_bytes, "reference_ops_per_unit": ref_ops,
                                 "note": "SURVEY.md 8(d): bytes the REFERENCE's algorithm would touch for this work (128 B per "
                                         "UpdateSI ...); the k-mer table replaces most of those steps, so this is a rate of "
                                         "work, not traffic, and is not compared with the HBM peak"}
        return {"value": value, "unit": "pairs/s" if self.paired else "reads/s", "steps": self.steps,
                "ms_per_step": self.elapsed / self.steps * 1e3, "units_per_gpu_per_step": self.n,
                "contexts_in_flight": self.nctx, "chunk": self.chunk,
                "overflow_retries_per_step": self.retries / max(self.steps, 1), "roofline": roof}

    def host_records(self, k):
        """finalised (classified, taxon) of the first k reads from the device's compact records"""
        rec = np.frombuffer(self.timed_compact[: k * COMPACT_BYTES].cpu().numpy().tobytes(), dtype=api.COMPACT_DTYPE)
        off = offsets(k, self.L, self.Lm, self.paired)
        res = self.clfs[0].finalize_compact(rec, off, paired=self.paired)
        return res["classified"].astype(np.uint8), res["taxon"].astype(np.uint64), rec

    def close(self):
        for c in self.clfs:
            c.close()
        del self.d_seqs, self.d_out, self.d_compact, self.d_offs, self.timed_compact


def host_buffers_leg(index, dtax, reads, Lm, seg, dev, calls=4, chunk=2_500_000, nthreads=2):
    """PCIe-inclusive rate of the entry point a host caller uses: kaiju_gpu_classify_batch_compact from page-locked host
    buffers, two contexts in two host threads (H2D / D2H of one overlap with the kernels of the other)."""
    import torch
    n, L = reads.shape
    chunk = min(chunk, n)
    params = api.default_params("mem", seg=seg)
    clfs = [api.Classifier(index, params) for _ in range(nthreads)]
    bufs = []
    for t in range(nthreads):
        seqs = torch.empty(chunk * L, dtype=torch.uint8, pin_memory=True).numpy()
        off = torch.empty(2 * chunk + 1, dtype=torch.int64, pin_memory=True).numpy().view(np.uint64)
        out = torch.empty(chunk * COMPACT_BYTES, dtype=torch.uint8, pin_memory=True).numpy().view(api.COMPACT_DTYPE)
        off[:] = offsets(chunk, L, Lm, False)
        bufs.append((seqs, off, out))
    starts = [(k * chunk) % max(n - chunk + 1, 1) for k in range(calls * nthreads)]

    def worker(t, ks, timed):
        torch.cuda.set_device(dev)
        seqs, off, out = bufs[t]
        for k in ks:
            if timed:
                seqs[:] = reads[starts[k]: starts[k] + chunk].reshape(-1)      # the caller's own packing of a batch
            clfs[t].classify_compact(dtax, seqs, off, out=out)

    for t in range(nthreads):                       # warm-up: buffers of the contexts get allocated
        seqs = bufs[t][0]
        seqs[:] = reads[:chunk].reshape(-1)
        worker(t, [0], False)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(t, list(range(t, calls * nthreads, nthreads)), True)) for t in range(nthreads)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    el = time.perf_counter() - t0
    # the same without the packing memcpy on the host (buffers already filled): the library's share
    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(t, list(range(t, calls * nthreads, nthreads)), False)) for t in range(nthreads)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    el2 = time.perf_counter() - t0
    for c in clfs:
        c.close()
    tot = calls * nthreads * chunk
    return {"value": tot / el2, "unit": "reads/s", "reads_per_call": chunk, "calls": calls * nthreads, "host_threads": nthreads,
            "with_host_packing_memcpy": tot / el,
            "bytes_per_read": {"h2d": L + 16, "d2h": COMPACT_BYTES},
            "entry_point": "kaiju_gpu_classify_batch_compact (page-locked host buffers in, 16-B records out, blocking); "
                           "never the headline value"}


def verbose_leg(index, reads, Lm, seg, W, n=400_000, calls=3, mode="mem"):
    """kaiju -v in MEM mode through the entry point a host caller uses (kaiju_gpu_classify_batch_verbose: host buffers in; 184-byte
    records, accessions and the matched peptides out - PCIe inclusive, never the headline value), next to the plain call on the
    same reads (kaiju_gpu_classify_batch), and columns 4 - 7 of its reads against the lines of the reference's -v run that the
    headline's CPU baseline left behind (ConsumerThread.cpp:614-623: match length, ids, accessions, peptides)"""
    n = min(n, len(reads))
    rd = np.ascontiguousarray(reads[:n, :Lm])
    seqs = rd.reshape(-1)
    off = offsets(n, Lm, Lm, False)
    clf = api.Classifier(index, api.default_params(mode, seg=seg))
    try:
        hits, v, text, stride = clf.classify_verbose_raw(seqs, off)          # (warm-up: the context's buffers)
        t0 = time.perf_counter()
        for _ in range(calls):
            hits, v, text, stride = clf.classify_verbose_raw(seqs, off)
        el = (time.perf_counter() - t0) / calls
        clf.classify(seqs, off)
        t0 = time.perf_counter()
        for _ in range(calls):
            plain = clf.classify(seqs, off)
        elp = (time.perf_counter() - t0) / calls
        out = {"value": n / el, "unit": "reads/s", "reads_per_call": n, "calls": calls, "plain_call_reads_per_s": n / elp,
               "cost_over_plain_call": round(el / elp, 3), "records_equal_plain_call": bool((hits == plain).all()),
               "entry_point": "kaiju_gpu_classify_batch_verbose (pageable host buffers in; records, accessions and peptides out; blocking) "
                              "against kaiju_gpu_classify_batch on the same reads; " +
                              ("MEM mode: k_mem_vb + k_mem_verbose + k_vb_pack" if mode == "mem" else "Greedy mode: k_greedy2_vb + k_mem_verbose<.., false> + k_vb_pack"),
               "parity": None}
        refv = f"{W}/cpu_{mode}_out_v.tsv"
        if os.path.exists(refv):
            accs, peps = clf.verbose_columns(v, text, stride)
            checked = bad = 0
            first_bad = None
            with open(refv) as f:
                for line in f:
                    c = line.rstrip("\n").split("\t")
                    r = int(c[1][1:])
                    if r >= n:
                        continue
                    if c[0] != "C":
                        continue                               # (U lines have no columns 4 - 7; C/U itself is the headline's parity)
                    checked += 1
                    h = hits[r]
                    ids = "".join(f"{int(x)}," for x in sorted(int(t) for t in h["taxid"][:int(h["n_ids"])]))
                    # (Greedy: a read the reference classifies passed its E-value gate; the columns are those of its best matches either way)
                    ok = (len(c) >= 7 and int(c[3]) == int(h["best"]) and c[4] == ids and c[5] == "".join(a + "," for a in accs[r]) and c[6] == peps[r])
                    if not ok:
                        bad += 1
                        first_bad = first_bad or (c[1], c[3:], int(h["best"]), accs[r], peps[r])
            out["parity"] = {"checked": checked, "mismatches": bad, "first": repr(first_bad)[:300] if first_bad else None,
                             "against": "columns 4 - 7 of the reference binary's -v lines for the same reads (classified reads)"}
        return out
    finally:
        clf.close()


def load_traffic(mode, paired, seg, nseq, per_launch, leg=None):
    """HBM bytes per launch of the leg's search kernel from the committed PMC passes (profiles/traffic.json); records of the
    legs other than headline / greedy / paired carry the leg's name (tests/tools/pmc_legs.sh)"""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            for rec in json.load(f)["measurements"]:
                if leg is not None:
                    if rec.get("leg") == leg and rec["reads_per_launch"] == int(per_launch):
                        return rec["hbm_bytes_per_launch"]
                    continue
                if rec.get("leg"):
                    continue
                if (rec["mode"] == mode and int(rec["seg"]) == int(seg) and rec["nseq"] == nseq and
                        bool(rec.get("paired", False)) == bool(paired) and rec["reads_per_launch"] == int(per_launch)):
                    _TRAFFIC_NOTES[(mode, bool(paired))] = rec.get("note")
                    return rec["hbm_bytes_per_launch"]
    except Exception:  # noqa: BLE001
        pass
    return None


_TRAFFIC_NOTES = {}          # what the matching record of profiles/traffic.json says about when it was collected


LINE_LIMIT = 4096            # the driver extracts the LAST stdout line; round 5's 28 KB line was not parsed


def _finite(x):
    """JSON has no Infinity / NaN: such a figure becomes null (json.dumps(.., allow_nan=False) then never raises)"""
    if isinstance(x, float):
        return x if math.isfinite(x) else None
    if isinstance(x, dict):
        return {k: _finite(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_finite(v) for v in x]
    if isinstance(x, np.generic):
        return _finite(x.item())
    return x


def _sig(x, digits=5):
    if isinstance(x, float) and math.isfinite(x) and x != 0.0:
        return float(f"{x:.{digits}g}")
    return x


LEG_NAMES = ("greedy", "paired", "hard", "hard_greedy", "wide", "wide_greedy", "long", "protein", "host_buffers", "verbose")


def summary_line(result: dict, detail_path) -> str:
    """The ONE line the driver parses: the contract's keys, the roofline of the dominant kernel, the CPU baseline, the parity
    totals and one figure per further leg - at most LINE_LIMIT bytes.  Everything else (op counts, stage tables, per-leg
    rooflines and baselines, index footprint, notes) is in the file `detail` names (and on stderr)."""
    r = _finite(result)
    cfg, roof = r.get("config", {}), r.get("roofline", {})
    out = {k: r.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                 "vs_baseline", "dtype", "data")}
    out["value"], out["ms_per_step"] = _sig(out["value"], 7), _sig(out["ms_per_step"], 6)
    out["config"] = {"workload": str(cfg.get("workload", ""))[:600], "reads_per_gpu_per_step": cfg.get("reads_per_gpu_per_step"),
                     "ranks": cfg.get("ranks"), "gather": str(cfg.get("gather") or "none")[:160], "gather_by": cfg.get("gather_by"),
                     "process_group": cfg.get("process_group"),
                     "per_rank_units_per_s": [_sig(v) for v in (cfg.get("per_rank_units_per_s") or [])][:16]}
    out["roofline"] = {k: _sig(roof.get(k)) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic",
                                                      "algorithmic_bytes_per_unit", "units_per_launch", "avg_launch_ms")}
    st = roof.get("stage_ms_per_step_exclusive")
    if st:
        out["stage_ms"] = {k: _sig(v, 4) for k, v in st.items()}
    cb = r.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {"value": _sig(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                               "sample": str(cb.get("sample", ""))[:200]}
    if "parity_checked_reads" in r:
        out["parity"] = {"checked": r.get("parity_checked_reads"), "mismatches": r.get("mismatches")}
        if r.get("parity_errors"):
            out["parity"]["errors"] = r["parity_errors"]
    legs = {}
    for nm in LEG_NAMES:
        if isinstance(r.get(nm), dict) and "value" in r[nm]:
            lg = {"value": _sig(r[nm]["value"]), "unit": r[nm].get("unit")}
            lroof = r[nm].get("roofline")
            if lroof:
                lg["kernel"], lg["frac"], lg["avg_launch_ms"] = lroof.get("kernel"), _sig(lroof.get("frac"), 3), _sig(lroof.get("avg_launch_ms"), 4)
                if lroof.get("traffic") is not None:
                    lg["traffic"] = _sig(lroof.get("traffic"), 4)
            legs[nm] = lg
    if legs:
        out["legs"] = legs
    out["detail"] = detail_path
    line = json.dumps(out, allow_nan=False, separators=(",", ":"))
    if len(line) >= LINE_LIMIT:                    # (cannot happen with the caps above; the contract matters more than the extras)
        for k in ("stage_ms", "legs"):
            out.pop(k, None)
        out["config"]["workload"] = out["config"]["workload"][:200]
        line = json.dumps(out, allow_nan=False, separators=(",", ":"))
    assert len(line) < LINE_LIMIT
    return line


def write_detail(result: dict, work: str, world: int):
    """the full structure of the run: into a file (repo-relative gpurun_out/ when writable, else --work) and onto stderr"""
    text = json.dumps(_finite(result), allow_nan=False, indent=1)
    print("[bench] detail:", json.dumps(_finite(result), allow_nan=False), file=sys.stderr, flush=True)
    for d in (os.path.join(ROOT, "gpurun_out"), work):
        try:
            os.makedirs(d, exist_ok=True)
            path = os.path.join(d, f"bench_detail_n{world}.json")
            with open(path, "w") as f:
                f.write(text)
            return os.path.relpath(path, ROOT) if path.startswith(ROOT + os.sep) else path
        except OSError:
            continue
    return None


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=int(os.environ.get("KAIJU_BENCH_READS", 10_000_000)),
                    help="reads per GPU per step")
    ap.add_argument("--contexts", type=int, default=int(os.environ.get("KAIJU_BENCH_CONTEXTS", 0)),
                    help="classification contexts that ping-pong the chunks of a step on their HIP streams; default 0 = per leg: two "
                         "for a MEM leg of 4 M reads and more (each context one half of the step: + 7.7 %% in round 6 - rounds 2 and 3 had "
                         "measured + 1.6 %% and nothing, the step had fewer kernels next to k_mem then), one otherwise (Greedy loses "
                         "10 %% with two: its persistent kernel leaves no room; DESIGN.md 5d)")
    ap.add_argument("--chunk", type=int, default=int(os.environ.get("KAIJU_BENCH_CHUNK", 0)),
                    help="reads (pairs) per launch; default 0: the step divided by the contexts (a persistent kernel ends in a tail in "
                         "which its lanes run dry - more launches per step than contexts only cost: 2 x 5 M 426, 4 x 2.5 M on two "
                         "contexts 396 M reads/s)")
    ap.add_argument("--nseq", type=int, default=int(os.environ.get("KAIJU_BENCH_NSEQ", 680_001)))
    ap.add_argument("--mode", default=os.environ.get("KAIJU_BENCH_MODE", "mem"), choices=["mem", "greedy"],
                    help="mode of the headline leg (the metric of BASELINE.json is quoted on mem)")
    ap.add_argument("--no-seg", action="store_true")
    ap.add_argument("--paired", action="store_true", help="headline leg on 2 x 150-bp pairs instead of single reads")
    ap.add_argument("--legs", default=os.environ.get("KAIJU_BENCH_LEGS", "greedy,paired,host,hard,wide,long,protein,verbose"),
                    help="further legs in the same line: greedy, paired, host, hard, wide, long, protein, verbose (comma separated; '' = none).  verbose: "
                         "kaiju -v in MEM mode through the host entry point, 400 000 reads, columns 4 - 7 against the reference's -v lines.  long / "
                         "protein: what users also feed it - 250-bp reads and protein reads (kaiju -p), MEM, --other-reads per step.  hard: MEM and Greedy on "
                         "a database that is NOT i.i.d. (synth.make_db_hard: families of 50-500 near-identical proteins, low-complexity "
                         "inserts; reads with Ns) - retries, inexact reads and the rate next to the i.i.d. legs.  wide: MEM and Greedy on "
                         "an index of 2^32 rows and more (the benchmark database with every protein --wide-copies times, written by "
                         "kaiju_build_fmi_replicated without a second sort; the .fmi is streamed to HBM and packed there): the kernels "
                         "of the layout with 64-bit positions - k_mem_wide2, k_mem_locate_wide / _team, k_greedy2_wide")
    ap.add_argument("--other-reads", type=int, default=2_000_000,
                    help="reads per step of the legs `long` (250-bp reads: mates of 192 - 287 nt take k_fragments_fast<.., 6>) and "
                         "`protein` (100-residue protein reads, kaiju -p: k_fragments_protein)")
    ap.add_argument("--wide-copies", type=int, default=23, help="copies of every protein in the index of the `wide` leg (23 x 191 M rows = 4.39 G > 2^32)")
    ap.add_argument("--wide-reads", type=int, default=2_000_000)
    ap.add_argument("--hard-nseq", type=int, default=200_001)
    ap.add_argument("--hard-reads", type=int, default=2_000_000)
    ap.add_argument("--leg-steps", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=2_000_000)
    ap.add_argument("--cpu-sample-legs", type=int, default=400_000)
    ap.add_argument("--work", default=os.environ.get("KAIJU_BENCH_WORK", "/tmp/kaiju_amd_bench"))
    ap.add_argument("--copies", type=int, default=1,
                    help="index of the database in which every protein occurs this many times (kaiju_build_fmi_replicated: no second "
                         "sort) - a refseq-class number of rows from a database a sixth the size")
    ap.add_argument("--image", action="store_true",
                    help="load the index through its device image (written once next to the .fmi, streamed from the file to HBM in "
                         "pieces) also at N = 1; N > 1 always does")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: --reads is the size of ONE workload per step, split over the ranks in contiguous blocks "
                         "(BASELINE configs[4]: 400 M reads sharded over 8 GPUs); default: weak scaling, --reads per GPU")
    ap.add_argument("--no-ref-ops", action="store_true",
                    help="skip the op counts of the reference's algorithm (roofline.work_rate): the instrumented oracle reads the "
                         ".fmi once more, a minute for a refseq-class index")
    ap.add_argument("--prepare-only", action="store_true", help="build the database, the .fmi (and the image with --image) and exit; no GPU needed")
    ap.add_argument("--gather", default=os.environ.get("KAIJU_BENCH_GATHER", "torch"), choices=["torch", "lib"],
                    help="who gathers the 16-byte records on rank 0: torch.distributed (backend nccl = RCCL; the default, what the "
                         "driver's launcher sets up) or the library's own collective (kaiju_gpu_comm_create / "
                         "kaiju_gpu_gather_compact: librccl opened by libkaiju_gpu.so, one ncclGather per chunk, no torch in the data path)")
    ap.add_argument("--parity-sample", type=int, default=200_000,
                    help="N > 1: reads (pairs) of EVERY rank whose gathered records rank 0 compares with the reference binary")
    args = ap.parse_args()

    # ---------------- N ranks: launch them ourselves when nobody did ----------------
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        print("[bench] starting", args.gpus, "ranks:", " ".join(cmd), file=sys.stderr, flush=True)
        os.execvp(cmd[0], cmd)

    import torch
    if args.prepare_only:
        rank, local_rank, world = 0, 0, 1                  # host only: database, .fmi (and image) under --work, then exit
    else:
        rank, local_rank, world = kdist.init("nccl")
        if world != args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the Kaiju HIP path has no CPU fallback")
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        assert dist.get_world_size() == world and dist.get_backend() == "nccl"
    seg = 0 if args.no_seg else 1
    W = args.work
    os.makedirs(W, exist_ok=True)
    legs_wanted = [x for x in args.legs.split(",") if x]

    # ---------------- workload (untimed) ----------------
    t0 = time.time()
    lines, leaves = synth.make_taxonomy()
    # (--nseq 15500001: 4.33 G residues, an index of 2^32 rows and more - the layout with 64-bit positions, BASELINE configs[3]
    #  class; the database generator without per-sequence Python loops serves from a few million sequences on, and sorting the
    #  suffixes of such a database takes about seven minutes of the host's cores before the first step is timed)
    big_db = args.nseq > 2_000_000
    db = big_database(W, args.nseq, rank) if big_db else synth.make_db(nseq=args.nseq, seed=12345, leaves=leaves)
    log(rank, f"database {db.nseq} seqs / {db.total_aa} aa ({time.time()-t0:.1f}s)")
    copies = max(1, args.copies)
    fmi, nodes = f"{W}/db_{args.nseq}{'' if copies == 1 else f'_x{copies}'}.fmi", f"{W}/nodes.dmp"
    if rank == 0:
        synth.write_nodes_dmp(nodes, lines)
        if not os.path.exists(fmi):
            t1 = time.time()
            faa = f"{W}/db_{args.nseq}.faa"
            if not big_db:
                synth.write_fasta(db, faa)
            if copies == 1:
                mkfmi.build_fmi(faa, fmi + ".tmp", threads=0, exponent=3)
            else:
                mkfmi.build_fmi_replicated(faa, fmi + ".tmp", copies, threads=0, exponent=3, copy_taxids=np.asarray(leaves, dtype=np.uint64))
            os.replace(fmi + ".tmp", fmi)
            if big_db:
                os.remove(faa)
            log(rank, f".fmi built ({time.time()-t1:.1f}s)")
        log(rank, f"index {os.path.getsize(fmi)/1e6:.0f} MB file{'' if copies == 1 else f' ({copies} copies of every protein)'} "
                  f"({time.time()-t0:.1f}s)")
    kdist.barrier()
    load_path = fmi
    load_info = {"path": "fmi: parsed, packed and uploaded from host memory"}
    # (N ranks, one host: a .fmi of 1 GiB and more is streamed to HBM by every rank for itself - it sits in the page cache, a rank
    #  needs the names and two page-locked pieces -, a smaller one is parsed and packed per rank, four at a time; the device image
    #  of round 4 only on request)
    if args.image:
        img = fmi + ".kjimg"
        if rank == 0 and not image_is_fresh(img, fmi):
            t1 = time.time()
            api.write_index_image(fmi, img + ".tmp")
            os.replace(img + ".tmp", img)
            load_info["image_write_s"] = time.time() - t1
            log(rank, f"device image written: {os.path.getsize(img)/1e9:.2f} GB ({time.time()-t1:.1f}s)")
        kdist.barrier()
        load_path = img
        load_info["path"] = ("device image: small arrays read, the arrays that grow with the index streamed file -> page-locked "
                             "pieces -> HBM (no host copy)")
        load_info["image_bytes"] = os.path.getsize(img)
    if args.prepare_only:
        log(rank, "prepared:", load_path)
        return
    # (the image pieces go through page-locked buffers, 0.5 GB per rank: all ranks may load at once; a .fmi is parsed on the host)
    conc = max(1, int(os.environ.get("KAIJU_BENCH_LOAD_CONCURRENCY", "8" if load_path != fmi or os.path.getsize(fmi) >= (1 << 30) else "4")))
    index = None
    t1 = time.time()
    for g in range(0, world, conc):
        if g <= rank < g + conc:
            index = api.Index(load_path, device=local_rank)
        kdist.barrier()
    load_info["seconds"] = time.time() - t1
    load_info["file_bytes"] = os.path.getsize(load_path)
    # what the load cost this process in host memory (VmHWM: its peak resident set so far - database arrays included; run the
    # .fmi build as a process of its own, --prepare-only, for this to be the loader's figure)
    load_info["host_rss_peak_bytes_after_load"] = vm_hwm_bytes()
    if load_path == fmi and os.path.getsize(fmi) >= (1 << 30) and os.environ.get("KAIJU_GPU_FMI_STREAM", "1") != "0":
        load_info["path"] = (".fmi streamed to HBM in page-locked pieces and packed on the device (fmi_stream.hip): no host copy of the "
                             "BWT, the samples or the packed arrays")
    log(rank, f"index in HBM: {index.footprint.total/1e9:.2f} GB, loaded in {load_info['seconds']:.1f}s ({load_info['path'].split(':')[0]})")
    dtax = api.DeviceTaxonomy(api.Taxonomy(nodes), local_rank)
    if args.gather == "lib" and (world > 1 or os.environ.get("KAIJU_DIST_FORCE_INIT") == "1"):
        Leg.comm = kdist.make_comm(W, rank, world, local_rank)
        log(rank, f"gather through the library's RCCL communicator ({world} rank(s))")
    n = args.reads
    Lm = 150
    if args.strong:
        if args.reads % world:
            raise SystemExit(f"--strong: --reads {args.reads} is not a multiple of the {world} ranks (equal blocks: one gather size)")
        n = args.reads // world

    def make(paired, count, seed):
        if args.strong:
            # rank r classifies the contiguous block [r * count, (r + 1) * count) of the one workload
            lo, hi = kdist.shard_bounds(count * world, rank, world)
            return reads_of_range(db, lo, hi, paired, seed * 1000)
        if paired:
            m1, m2 = synth.make_pairs(db, count, seed=seed + rank)
            return np.concatenate([m1, m2], axis=1)           # pair r = mate 1 followed by mate 2 in the sequence buffer
        return synth.make_reads(db, count, seed=seed + rank)

    t0 = time.time()
    reads = make(args.paired, n, 778 if args.paired else 777)
    log(rank, f"{n} {'pairs' if args.paired else 'reads'}/GPU generated ({time.time()-t0:.1f}s), index {index.info.device_bytes/1e6:.0f} MB in HBM")

    def do_leg(name, mode, paired, rd, steps, warmup):
        leg = Leg(name, mode, paired, rd, Lm, index, dtax, dev, rank, world, seg, args.chunk, args.contexts)
        leg.run(steps, warmup)
        log(rank, f"leg {name}: {leg.n * world * steps / leg.elapsed / 1e6:.1f} M {'pairs' if paired else 'reads'}/s")
        return leg

    head = do_leg("headline", args.mode, args.paired, reads, args.steps, args.warmup)
    # per-rank rates and what was gathered (rank 0 needs them)
    per_rank = [n * args.steps / head.elapsed_rank]
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([per_rank[0]], dtype=torch.float64, device=dev)
        outl = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(outl, t)
        per_rank = [float(x.item()) for x in outl]
    hits = np.frombuffer(head.d_out[: min(n, 1_000_000) * HIT_BYTES].cpu().numpy().tobytes(), dtype=api.HIT_DTYPE)
    frac_hit = float((hits["n_ids"] > 0).mean())
    del hits

    # ---------------- further legs ----------------
    extra = {}
    keep = {"headline": (head, reads)}
    if "greedy" in legs_wanted and not (args.mode == "greedy" and not args.paired):
        leg = do_leg("greedy", "greedy", False, reads if not args.paired else make(False, n, 777), args.leg_steps, 1)
        extra["greedy"] = leg
        keep["greedy"] = (leg, leg.reads)
    if "paired" in legs_wanted and not args.paired:
        rd = make(True, max(n // 2, 1), 778)
        leg = do_leg("paired", "mem", True, rd, args.leg_steps, 1)
        extra["paired"] = leg
        keep["paired"] = (leg, rd)
    host = None
    if "host" in legs_wanted and world == 1 and not args.paired:
        try:
            host = host_buffers_leg(index, dtax, reads, Lm, seg, dev)
            log(rank, f"leg host_buffers: {host['value']/1e6:.1f} M reads/s")
        except Exception as e:  # noqa: BLE001
            log(rank, "host_buffers leg failed:", repr(e))

    # ---------------- reads that are not 150 bp: 250-bp reads (six-unit stage 1) and protein reads (kaiju -p) ----------------
    other = {}
    if world == 1 and not args.paired and not big_db:
        try:
            if "long" in legs_wanted:
                rd = synth.make_reads(db, min(args.other_reads, n), seed=781, read_len=250)
                leg = Leg("long", "mem", False, rd, 250, index, dtax, dev, rank, world, seg, args.chunk, args.contexts)
                leg.run(args.leg_steps, 1)
                log(rank, f"leg long (250-bp reads): {leg.n * args.leg_steps / leg.elapsed / 1e6:.1f} M reads/s")
                other["long"] = leg
            if "protein" in legs_wanted:
                rd = synth.make_protein_reads(db, min(args.other_reads, n), seed=782, read_len=100)
                leg = Leg("protein", "mem", False, rd, 100, index, dtax, dev, rank, world, seg, args.chunk, args.contexts, protein=True)
                leg.run(args.leg_steps, 1)
                log(rank, f"leg protein (100-residue protein reads, kaiju -p): {leg.n * args.leg_steps / leg.elapsed / 1e6:.1f} M reads/s")
                other["protein"] = leg
        except Exception as e:  # noqa: BLE001
            log(rank, "long / protein leg failed:", repr(e))

    # ---------------- the hostile leg: a database that is not i.i.d. (its own index next to the benchmark's) ----------------
    hard = None
    if "hard" in legs_wanted and world == 1 and not big_db and copies == 1:
        try:
            t1 = time.time()
            hdb = synth.make_db_hard(nseq=args.hard_nseq, seed=4321, leaves=leaves)
            hfmi = f"{W}/db_hard_{args.hard_nseq}.fmi"
            if not os.path.exists(hfmi):
                (synth.write_fasta_large if hdb.nseq > 1_000_000 else synth.write_fasta)(hdb, f"{W}/db_hard_{args.hard_nseq}.faa")
                mkfmi.build_fmi(f"{W}/db_hard_{args.hard_nseq}.faa", hfmi + ".tmp", threads=0, exponent=3)
                os.replace(hfmi + ".tmp", hfmi)
            hindex = api.Index(hfmi, device=local_rank)
            hreads = synth.sprinkle_n(synth.make_reads(hdb, args.hard_reads, seed=779))
            log(rank, f"hard database {hdb.nseq} seqs / {hdb.total_aa} aa, index {hindex.footprint.total/1e9:.2f} GB in HBM ({time.time()-t1:.1f}s)")
            hard = {"db": hdb, "fmi": hfmi, "reads": hreads, "index": hindex, "legs": {}}
            for nm, mode in (("hard", "mem"), ("hard_greedy", "greedy")):
                leg = Leg(nm, mode, False, hreads, Lm, hindex, dtax, dev, rank, world, seg, args.chunk, args.contexts)
                leg.run(args.leg_steps, 1)
                log(rank, f"leg {nm}: {leg.n * args.leg_steps / leg.elapsed / 1e6:.1f} M reads/s, {leg.retries / max(args.leg_steps, 1):.0f} reads per step in the retry pass")
                hard["legs"][nm] = leg
        except Exception as e:  # noqa: BLE001
            log(rank, "hard leg failed:", repr(e))
            hard = None

    # ---------------- the wide leg: 2^32 rows and more on the benchmark database (its own index; 64-bit positions) ----------------
    wide = None
    if "wide" in legs_wanted and world == 1 and not big_db and copies == 1 and not args.paired:
        try:
            t1 = time.time()
            wc = max(2, args.wide_copies)
            wfmi = f"{W}/db_{args.nseq}_x{wc}.fmi"
            if not os.path.exists(wfmi):
                faa = f"{W}/db_{args.nseq}.faa"
                if not os.path.exists(faa):
                    synth.write_fasta(db, faa)
                mkfmi.build_fmi_replicated(faa, wfmi + ".tmp", wc, threads=0, exponent=3, copy_taxids=np.asarray(leaves, dtype=np.uint64))
                os.replace(wfmi + ".tmp", wfmi)
            t_build = time.time() - t1
            t2 = time.time()
            windex = api.Index(wfmi, device=local_rank)
            t_load = time.time() - t2
            wreads = np.ascontiguousarray(reads[: min(args.wide_reads, n)])
            log(rank, f"wide index: every protein x {wc} = {windex.info.bwtlen} rows, .fmi {os.path.getsize(wfmi)/1e9:.2f} GB ({t_build:.1f}s), "
                      f"{windex.footprint.total/1e9:.2f} GB in HBM, loaded in {t_load:.1f}s")
            wide = {"fmi": wfmi, "reads": wreads, "index": windex, "legs": {}, "copies": wc, "load_s": t_load, "build_s": t_build}
            for nm, mode in (("wide", "mem"), ("wide_greedy", "greedy")):
                leg = Leg(nm, mode, False, wreads, Lm, windex, dtax, dev, rank, world, seg, args.chunk, args.contexts)
                leg.run(args.leg_steps, 1)
                log(rank, f"leg {nm}: {leg.n * args.leg_steps / leg.elapsed / 1e6:.1f} M reads/s, {leg.retries / max(args.leg_steps, 1):.0f} reads per step in the retry pass")
                wide["legs"][nm] = leg
        except Exception as e:  # noqa: BLE001
            log(rank, "wide leg failed:", repr(e))
            wide = None

    # N > 1: a sample of EVERY rank's reads and of the records its timed kernels wrote goes to rank 0, which checks each against
    # the reference binary (an N-GPU line must not state a rate with nothing looking at the gathered records)
    rank_samples = None
    if world > 1 and not args.no_cpu_baseline and args.parity_sample > 0:
        k = min(args.parity_sample, n)
        rd_t = torch.from_numpy(np.ascontiguousarray(reads[:k]).reshape(-1)).to(dev)
        rec_t = head.timed_compact[: k * COMPACT_BYTES].contiguous()
        rank_samples = (kdist.gather_to_root(rd_t, world, rank), kdist.gather_to_root(rec_t, world, rank), k)

    if rank != 0:
        return

    # ---------------- accounting legs on the host: reference baseline + parity, reference op counts ----------------
    def cpu_leg(leg, rd, sample, oracle_sample, fmi=fmi, tag_suffix=""):
        out = {"ref_ops": None, "baseline": None, "parity": None}
        if args.no_cpu_baseline or world != 1:
            return out
        try:
            if not args.no_ref_ops and oracle_sample > 0:
                out["ref_ops"] = reference_ops(W, fmi, rd[:oracle_sample], leg.mode, seg, leg.paired, leg.Lm)
        except Exception as e:  # noqa: BLE001 - the accounting legs must never kill the measurement
            log(rank, f"reference op counts ({leg.name}) failed:", repr(e))
        try:
            bl, ref = run_reference(W, fmi, nodes, rd, leg.Lm, leg.paired, leg.mode, seg, sample, tag_suffix,
                                    extra=("-p",) if leg.protein else ())
            out["baseline"] = bl
            if ref is not None:
                k = len(ref[0])
                cls, tax, rec = leg.host_records(k)
                out["parity"] = {**compare_with_reference(cls, tax, rec["best"].astype(np.int64), ref),
                                 "records": "timed launch (snapshot of the records written by the same kernels as the timed steps, "
                                            "taken before the counting pass)",
                                 "counting_instantiation_equals_timed": bool(leg.count_equals_timed),
                                 "against": "output lines of the unmodified reference binary (kaiju -v) on the same reads: C/U, "
                                            "taxon id and, for classified reads, column 4 (match length in MEM, score in Greedy)"}
        except Exception as e:  # noqa: BLE001
            log(rank, f"cpu baseline / parity ({leg.name}) failed:", repr(e))
        return out

    acc = {"headline": cpu_leg(head, reads, args.cpu_sample, 30000)}
    for nm in ("greedy", "paired"):
        if nm in extra:
            acc[nm] = cpu_leg(extra[nm], keep[nm][1], args.cpu_sample_legs, 10000)

    if hard is not None:
        for nm, leg in hard["legs"].items():
            acc[nm] = cpu_leg(leg, hard["reads"], min(args.cpu_sample_legs, 200_000), 5000, fmi=hard["fmi"], tag_suffix="_hard")
    for nm, leg in other.items():
        acc[nm] = cpu_leg(leg, leg.reads, min(args.cpu_sample_legs, 200_000), 0, tag_suffix="_" + nm)
    if wide is not None:
        # (one run of the reference per mode: it reads the 8 GB .fmi each time; no op counts of the instrumented oracle there)
        for nm, leg in wide["legs"].items():
            acc[nm] = cpu_leg(leg, wide["reads"], min(args.cpu_sample_legs, 100_000), 0, fmi=wide["fmi"], tag_suffix="_wide")
    per_rank_parity = None
    if rank_samples is not None:
        rds, recs, k = rank_samples
        per_rank_parity = []
        for r in range(world):
            try:
                rd = rds[r].cpu().numpy().reshape(k, head.L)
                rec = np.frombuffer(recs[r].cpu().numpy().tobytes(), dtype=api.COMPACT_DTYPE)
                bl, ref = run_reference(W, fmi, nodes, rd, Lm, head.paired, head.mode, seg, k)
                res = head.clfs[0].finalize_compact(rec, offsets(k, head.L, Lm, head.paired), paired=head.paired)
                pr = compare_with_reference(res["classified"].astype(np.uint8), res["taxon"].astype(np.uint64), rec["best"].astype(np.int64), ref)
                pr["rank"] = r
                if r == 0 and bl is not None:
                    acc["headline"]["baseline"] = bl
                per_rank_parity.append(pr)
                log(rank, f"parity of rank {r}'s gathered sample: {pr['checked']} checked, {pr['mismatches']} mismatches")
            except Exception as e:  # noqa: BLE001
                log(rank, f"parity of rank {r} failed:", repr(e))
                per_rank_parity.append({"rank": r, "error": repr(e)})

    hr = head.result(world, acc["headline"]["ref_ops"],
                     load_traffic(head.mode, head.paired, seg, db.nseq, head.bounds[0][1] - head.bounds[0][0]), db.nseq)
    result = {
        "metric": "classified reads/sec (150 bp)",
        "value": hr["value"], "unit": hr["unit"], "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": hr["ms_per_step"], "higher_is_better": True, "scaling": "strong" if args.strong else "weak",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": f"{('refseq-class' if index.info.bwtlen >= 40e9 else 'refseq_ref-class') + ' (2^32 rows and more: 64-bit positions)' if index.info.bwtlen >= 2 ** 32 else 'viruses-like'} "
                               f"synthetic index ({db.nseq} proteins, {db.total_aa} aa{'' if copies == 1 else f', every protein x {copies}: {index.info.bwtlen} rows'}, .fmi "
                               f"{os.path.getsize(fmi)/1e6:.0f} MB, e=3); {n} synthetic {'2x' if args.paired else ''}{Lm}-bp reads{' (pairs)' if args.paired else ''} per GPU per step{' (one workload of %d split over the ranks)' % args.reads if args.strong else ''} "
                               f"(70% back-translated DB windows, 30% random); kaiju -a {args.mode} -m 11"
                               f"{'' if seg else ' -X'} (SEG {'on' if seg else 'off'})",
                   "reads_per_gpu_per_step": n, "chunk": head.chunk, "contexts_in_flight": head.nctx, "index_replicated": True,
                   "index_hbm_bytes": index.footprint.as_dict(), "index_load": load_info,
                   "timed_region": "reads resident in HBM before the timed region, 16-byte records left in HBM (gathered to rank 0 at "
                                   "N > 1); the PCIe-inclusive rate is the host_buffers leg, never `value`",
                   "ranks": world, "per_rank_units_per_s": per_rank,
                   "process_group": ({"backend": torch.distributed.get_backend(), "world_size": torch.distributed.get_world_size()}
                                     if torch.distributed.is_initialized() else None),
                   "gather_by": ("library: kaiju_gpu_gather_compact (librccl, ncclGather)" if Leg.comm is not None else
                                 "torch.distributed.gather (backend nccl = RCCL)" if torch.distributed.is_initialized() else None),
                   "gather": (f"one async RCCL gather of 16-B records (device LCA of the 184-B hit records) per chunk to rank 0; "
                              f"{head.gathered_timed / max(args.steps, 1):.0f} B into rank 0 per step"
                              if world > 1 else "none (1 GPU); device LCA to 16-B records still runs"),
                   "fraction_reads_with_hit": round(frac_hit, 4), "overflow_retries_per_step": hr["overflow_retries_per_step"],
                   "design_note": "one READ per lane following the reference's sequential pruning (DESIGN.md 3.3), rank blocks "
                                  "read straight from HBM as 128-byte lines - not one fragment per lane with LDS-staged Occ blocks"},
        "roofline": hr["roofline"],
    }
    if acc["headline"]["baseline"] is not None:
        result["cpu_baseline"] = acc["headline"]["baseline"]
    parity = {}
    if acc["headline"]["parity"] is not None:
        parity["headline"] = acc["headline"]["parity"]
    for nm in ("greedy", "paired"):
        if nm in extra:
            lr = extra[nm].result(world, acc[nm]["ref_ops"],
                                  load_traffic(extra[nm].mode, extra[nm].paired, seg, db.nseq,
                                               extra[nm].bounds[0][1] - extra[nm].bounds[0][0]), db.nseq)
            lr["workload"] = ("the same index and reads, kaiju -a greedy -e 3 (BASELINE configs[2])" if nm == "greedy" else
                              f"the same index, {extra[nm].n} synthetic 2x{Lm}-bp pairs per GPU per step, kaiju -a mem (shape of BASELINE configs[3])")
            if acc[nm]["baseline"] is not None:
                lr["cpu_baseline"] = acc[nm]["baseline"]
            if acc[nm]["parity"] is not None:
                parity[nm] = acc[nm]["parity"]
            result[nm] = lr
    if hard is not None:
        for nm, leg in hard["legs"].items():
            lr = leg.result(world, acc[nm]["ref_ops"], load_traffic(None, None, None, None, leg.bounds[0][1] - leg.bounds[0][0], leg=nm), hard["db"].nseq)
            rec = np.frombuffer(leg.timed_compact.cpu().numpy().tobytes(), dtype=api.COMPACT_DTYPE)
            lr["workload"] = (f"NOT i.i.d.: {hard['db'].nseq} proteins / {hard['db'].total_aa} aa in families of 50-500 near-identical members "
                              f"(0.5-3 % substitutions), low-complexity inserts in 5 % of them; {leg.n} 150-bp reads per step, 5 % of them with "
                              f"1-4 Ns; kaiju -a {leg.mode}")
            lr["inexact_reads_per_step"] = int((rec["info"] >> 31).sum())
            lr["id_cap_reads_per_step"] = int(((rec["info"] >> 8) & 1).sum())
            lr["fraction_reads_with_hit"] = round(float(((rec["info"] & 0xff) > 0).mean()), 4)
            if acc[nm]["baseline"] is not None:
                lr["cpu_baseline"] = acc[nm]["baseline"]
            if acc[nm]["parity"] is not None:
                parity[nm] = acc[nm]["parity"]
            result[nm] = lr
    for nm, leg in other.items():
        lr = leg.result(world, None, load_traffic(None, None, None, None, leg.bounds[0][1] - leg.bounds[0][0], leg=nm), db.nseq)
        rec = np.frombuffer(leg.timed_compact.cpu().numpy().tobytes(), dtype=api.COMPACT_DTYPE)
        lr["workload"] = (f"the same index, {leg.n} synthetic 250-bp reads per step (mates of 192 - 287 nt: the six-unit stage 1, k_fragments_fast<.., 6>), kaiju -a mem"
                          if nm == "long" else
                          f"the same index, {leg.n} synthetic protein reads of 100 residues per step (70% database windows with 0-5 substitutions), "
                          f"kaiju -a mem -p (k_fragments_protein)")
        lr["fraction_reads_with_hit"] = round(float(((rec["info"] & 0xff) > 0).mean()), 4)
        if acc.get(nm, {}).get("baseline") is not None:
            lr["cpu_baseline"] = acc[nm]["baseline"]
        if acc.get(nm, {}).get("parity") is not None:
            parity[nm] = acc[nm]["parity"]
        result[nm] = lr
    if wide is not None:
        wix = wide["index"]
        for nm, leg in wide["legs"].items():
            lr = leg.result(world, None, load_traffic(None, None, None, None, leg.bounds[0][1] - leg.bounds[0][0], leg=nm), db.nseq * wide["copies"])
            rec = np.frombuffer(leg.timed_compact.cpu().numpy().tobytes(), dtype=api.COMPACT_DTYPE)
            lr["workload"] = (f"index of 2^32 rows and more: the benchmark database with every protein x {wide['copies']} (copy t under another "
                              f"taxon) = {wix.info.bwtlen} rows, .fmi {os.path.getsize(wide['fmi'])/1e9:.2f} GB written by kaiju_build_fmi_replicated "
                              f"(no second sort); the first {leg.n} reads of the headline; kaiju -a {leg.mode}")
            lr["index_hbm_bytes"] = wix.footprint.as_dict()
            lr["index_load"] = {"path": ".fmi streamed to HBM in page-locked pieces and packed on the device (fmi_stream.hip): no host copy "
                                        "of the BWT, the samples or the packed arrays" if os.path.getsize(wide["fmi"]) >= (1 << 30) and
                                        os.environ.get("KAIJU_GPU_FMI_STREAM", "1") != "0" else "fmi: parsed, packed and uploaded from host memory",
                                "seconds": wide["load_s"], "file_bytes": os.path.getsize(wide["fmi"]), "fmi_build_seconds": wide["build_s"]}
            lr["id_cap_reads_per_step"] = int(((rec["info"] >> 8) & 1).sum())
            lr["fraction_reads_with_hit"] = round(float(((rec["info"] & 0xff) > 0).mean()), 4)
            if acc.get(nm, {}).get("baseline") is not None:
                lr["cpu_baseline"] = acc[nm]["baseline"]
            if acc.get(nm, {}).get("parity") is not None:
                parity[nm] = acc[nm]["parity"]
                lr["parity"] = acc[nm]["parity"]
            result[nm] = lr
    if host is not None:
        result["host_buffers"] = host
    if "verbose" in legs_wanted and world == 1 and args.mode == "mem" and not args.paired and not big_db:
        try:
            vl = verbose_leg(index, reads, Lm, seg, W)
            result["verbose"] = vl
            log(rank, f"leg verbose: {vl['value']/1e6:.1f} M reads/s ({vl['cost_over_plain_call']} x the plain host call), parity {vl['parity']}")
            if vl.get("parity"):
                parity["verbose"] = {"checked": vl["parity"]["checked"], "mismatches": vl["parity"]["mismatches"]}
            if "greedy" in extra:                          # (the Greedy leg's CPU baseline left the reference's -v lines of its sample behind)
                vg = verbose_leg(index, keep["greedy"][1], Lm, seg, W, mode="greedy")
                vl["greedy"] = vg
                log(rank, f"leg verbose, Greedy: {vg['value']/1e6:.1f} M reads/s ({vg['cost_over_plain_call']} x the plain host call), parity {vg['parity']}")
                if vg.get("parity"):
                    parity["verbose_greedy"] = {"checked": vg["parity"]["checked"], "mismatches": vg["parity"]["mismatches"]}
        except Exception as e:  # noqa: BLE001
            log(rank, "verbose leg failed:", repr(e))
    if per_rank_parity is not None:
        parity["per_rank"] = per_rank_parity
        result["parity"] = parity
        ok = [p for p in per_rank_parity if "checked" in p]
        result["parity_checked_reads"] = sum(p["checked"] for p in ok)
        # (a rank whose comparison could not run is an error of the line, not a rank without mismatches)
        result["parity_errors"] = len(per_rank_parity) - len(ok)
        result["mismatches"] = sum(p["mismatches"] for p in ok) if len(ok) == len(per_rank_parity) else None
        parity = {}
    if parity:
        result["parity"] = parity
        result["parity_checked_reads"] = sum(p["checked"] for p in parity.values())
        result["mismatches"] = sum(p["mismatches"] for p in parity.values())
    # (RCCL leaves a line about its library path in the C library's stdout buffer: out with it BEFORE the result, so that the
    #  JSON line is the last thing this process prints)
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass
    detail = write_detail(result, W, world)
    print(summary_line(result, detail), flush=True)


if __name__ == "__main__":
    main()
