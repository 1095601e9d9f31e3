#!/usr/bin/env python3
"""Benchmark of the MI355X Kaiju classification path.

    python bench.py --gpus N --steps K --warmup W

One *step* is one pass of the hot path (six-frame translation -> SEG -> MEM search on the protein
FM-index -> locate/taxon ids, kaiju_gpu_classify_batch_device) over one batch of synthetic reads
that is already resident in HBM.  The workload is BASELINE.json configs[1]: a viruses-like
synthetic index (680 001 proteins, 190 M aa) and 10 M synthetic 150-bp reads per
GPU, `-a mem -m 11`, SEG on (the reference's default).  With N > 1 (launched by
torch.distributed.run, one rank per GPU) every rank holds a replica of the index and classifies
its own 10 M reads (weak scaling); the per-read hit records are collected on rank 0 with one
asynchronous gather per chunk (RCCL over xGMI).  Rank 0 prints ONE JSON line.

`roofline` is computed for the dominant search kernel (k_mem): algorithmic bytes (128 B per
reference-equivalent UpdateSI + 64 B per LF step + 8 B per SA sample + 150 B read + 184 B hit
record, op counts from the instrumented oracle on a sample of the same reads) divided by the
kernel's average duration measured with HIP events on the launch stream.  `cpu_baseline` times
the unmodified reference binary (oracle/_ref/kaiju -z <cores>) on a bounded sample of the same
reads on this host (classification phase only).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
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


def fastq_bytes(reads: np.ndarray) -> bytes:
    """fixed-width FASTQ records, fully vectorised"""
    n, L = reads.shape
    names = np.char.add("@r", np.char.zfill(np.arange(n).astype(str), 8)).astype("S10")
    rec = np.empty((n, 10 + 1 + L + 3 + L + 1), dtype=np.uint8)
    rec[:, :10] = np.frombuffer(names.tobytes(), dtype=np.uint8).reshape(n, 10)
    rec[:, 10] = 10
    rec[:, 11:11 + L] = reads
    rec[:, 11 + L:14 + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    rec[:, 14 + L:14 + 2 * L] = ord("I")
    rec[:, 14 + 2 * L] = 10
    return rec.tobytes()


def cpu_baseline(W, fmi, nodes, reads, mode, seg, sample_reads, oracle_reads):
    """reference binary on a bounded sample + instrumented op counts from the oracle (test
    infrastructure used as the CPU baseline / accounting only, never on the measured path)"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    out = {"ops": None, "baseline": None}
    O = po.Oracle()
    oix = O.load_fmi(fmi)
    k = min(oracle_reads, len(reads))
    seqs, off = synth.pack_reads(reads[:k])
    O.counters(reset=True)
    t0 = time.time()
    O.classify(oix, None, O.params(mode, seg=seg), seqs, off)
    t_or = time.time() - t0
    c = O.counters(reset=True)
    out["ops"] = {"update_si": c["update_si"] / k, "lf_steps": c["fmindex_current"] / k,
                  "sa_decodes": c["sa_decode"] / k, "sample": k}
    if po.have_ref():
        cores = os.cpu_count() or 1
        s = min(sample_reads, len(reads))
        fq = f"{W}/cpu_sample.fq"
        with open(fq, "wb") as f:
            f.write(fastq_bytes(reads[:s]))
        with open(f"{W}/cpu_one.fq", "wb") as f:
            f.write(fastq_bytes(reads[:1]))
        base = [po.REF_KAIJU, "-t", nodes, "-f", fmi, "-a", mode, "-z", str(cores), "-o", f"{W}/cpu_out.tsv"]
        if not seg:
            base.append("-X")
        t0 = time.time()
        subprocess.run(base + ["-i", f"{W}/cpu_one.fq"], check=True, stderr=subprocess.DEVNULL)
        t_load = time.time() - t0
        t0 = time.time()
        subprocess.run(base + ["-i", fq], check=True, stderr=subprocess.DEVNULL)
        t_all = time.time() - t0
        t_cls = max(t_all - t_load, 1e-6)
        out["baseline"] = {"value": s / t_cls, "unit": "reads/s", "cores": cores, "kind": "reference",
                           "sample": f"{s} of the benchmark reads, kaiju -z {cores} -a {mode}"
                                     f"{'' if seg else ' -X'}; wall {t_all:.1f}s minus index load {t_load:.1f}s"}
    else:
        out["baseline"] = {"value": k / t_or, "unit": "reads/s", "cores": 1, "kind": "port",
                           "sample": f"{k} of the benchmark reads through oracle/libkaiju_oracle.so"}
    return out


def tot_reads_per_launch(kern_ms):
    return sum(m for _, m in kern_ms) / max(len(kern_ms), 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=int(os.environ.get("KAIJU_BENCH_READS", 10_000_000)),
                    help="reads per GPU per step")
    ap.add_argument("--contexts", type=int, default=0,
                    help="classification contexts that ping-pong the chunks (default: 2 for mem, 1 for greedy)")
    ap.add_argument("--chunk", type=int, default=int(os.environ.get("KAIJU_BENCH_CHUNK", 5_000_000)))
    ap.add_argument("--nseq", type=int, default=int(os.environ.get("KAIJU_BENCH_NSEQ", 680_001)))
    ap.add_argument("--mode", default=os.environ.get("KAIJU_BENCH_MODE", "mem"), choices=["mem", "greedy"])
    ap.add_argument("--no-seg", action="store_true")
    ap.add_argument("--paired", action="store_true", help="2 x 150-bp pairs (BASELINE config 4 shape) instead of single reads")
    ap.add_argument("--protein", type=int, default=0, metavar="LEN",
                    help="protein reads of LEN residues (kaiju -p workload; not the headline metric) instead of 150-bp reads")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=2_000_000)
    ap.add_argument("--work", default=os.environ.get("KAIJU_BENCH_WORK", "/tmp/kaiju_amd_bench"))
    args = ap.parse_args()

    import torch
    rank, local_rank, world = kdist.init("nccl")
    if world != args.gpus:
        if rank == 0:
            print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: using {world}", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the Kaiju HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    seg = 0 if args.no_seg else 1
    W = args.work
    os.makedirs(W, exist_ok=True)

    # ---------------- workload (untimed) ----------------
    t0 = time.time()
    lines, leaves = synth.make_taxonomy()
    db = synth.make_db(nseq=args.nseq, seed=12345, leaves=leaves)
    fmi, nodes = f"{W}/db_{args.nseq}.fmi", f"{W}/nodes.dmp"
    if rank == 0:
        synth.write_nodes_dmp(nodes, lines)
        if not os.path.exists(fmi):
            synth.write_fasta(db, f"{W}/db_{args.nseq}.faa")
            mkfmi.build_fmi(f"{W}/db_{args.nseq}.faa", fmi + ".tmp", threads=0, exponent=3)
            os.replace(fmi + ".tmp", fmi)
        log(rank, f"database {db.nseq} seqs / {db.total_aa} aa, index {os.path.getsize(fmi)/1e6:.0f} MB file "
                  f"({time.time()-t0:.1f}s)")
    kdist.barrier()
    index = api.Index(fmi, device=local_rank)
    if args.protein and args.paired:
        raise SystemExit("--protein has no paired mode")
    params = api.default_params(args.mode, seg=seg, input_is_protein=1 if args.protein else 0)
    clf = api.Classifier(index, params)
    t0 = time.time()
    n = args.reads
    if args.paired:
        # pair r = mate 1 followed by mate 2 in the sequence buffer
        m1, m2 = synth.make_pairs(db, n, seed=778 + rank)
        reads = np.concatenate([m1, m2], axis=1)
        Lm = m1.shape[1]
    elif args.protein:
        reads = synth.make_protein_reads(db, n, seed=779 + rank, read_len=args.protein)
        Lm = reads.shape[1]
    else:
        reads = synth.make_reads(db, n, seed=777 + rank)
        Lm = reads.shape[1]
    L = reads.shape[1]                       # bytes per read (pair) in the buffer
    clf.set_max_read_length(Lm)
    d_seqs = torch.from_numpy(reads.reshape(-1)).to(dev)
    chunk = min(args.chunk, n)
    bounds = [(lo, min(n, lo + chunk)) for lo in range(0, n, chunk)]
    # off[2r] = off[2r+1]... unpaired: mate 2 empty
    d_offs = []
    for lo, hi in bounds:
        m = hi - lo
        o = np.empty(2 * m + 1, dtype=np.int64)
        o[0::2] = np.arange(m + 1, dtype=np.int64) * L
        o[1::2] = o[2::2] if not args.paired else o[0:-1:2] + Lm
        d_offs.append(torch.from_numpy(o).to(dev))
    d_out = torch.zeros(n * HIT_BYTES, dtype=torch.uint8, device=dev)
    # what leaves the GPU: 16-byte records (LCA computed on the device), not the 184-byte id lists
    dtax = api.DeviceTaxonomy(api.Taxonomy(nodes), local_rank)
    d_compact = torch.zeros(n * COMPACT_BYTES, dtype=torch.uint8, device=dev)
    log(rank, f"{n} reads/GPU resident in HBM ({time.time()-t0:.1f}s), index {index.info.device_bytes/1e6:.0f} MB in HBM")
    # Two classification contexts ping-pong the chunks on two HIP streams (the "two host threads per GPU
    # on separate streams" of SURVEY.md 8b): while one chunk is in its HBM-bound search kernel the next one
    # runs its ALU/latency-bound stage 1 and SEG pass.  --contexts 1 = strictly one chunk after the other.
    # (two overlapping Greedy kernels only slow each other down: both are bound by instruction issue)
    nctx = args.contexts if args.contexts > 0 else (2 if args.mode == "mem" else 1)
    clfs = [clf] + [api.Classifier(index, params) for _ in range(nctx - 1)]
    for c in clfs:
        c.set_max_read_length(L)
    streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(nctx - 1)]

    kern_ms = []          # k_mem / k_greedy2 main-pass durations (HIP events on the launch stream)
    stage_ms = {"translate": 0.0, "seg": 0.0, "search": 0.0, "retry": 0.0}
    retries = 0

    def collect(c, m, record):
        nonlocal retries
        st = c.stats()                  # HIP events of that context's last chunk (blocks until its kernels are done)
        if st.error_flags:
            raise SystemExit(f"device-side capacity error flags {st.error_flags}")
        if record:
            kern_ms.append((st.ms_search, m))
            stage_ms["translate"] += st.ms_translate
            stage_ms["seg"] += st.ms_seg
            stage_ms["search"] += st.ms_search
            stage_ms["retry"] += st.ms_retry
            retries += st.n_overflow_retries

    def one_step(record, nctx=nctx):
        g = kdist.HitGatherer(world, rank, keep_results=False)
        pending = [None] * nctx
        main = torch.cuda.current_stream(dev)
        for s in streams[1:]:
            s.wait_stream(main)
        for k, ((lo, hi), d_off) in enumerate(zip(bounds, d_offs)):
            c, s = clfs[k % nctx], streams[k % nctx]
            if pending[k % nctx] is not None:
                collect(c, pending[k % nctx], record)
            m = hi - lo
            out_view = d_out[lo * HIT_BYTES: hi * HIT_BYTES]
            cview = d_compact[lo * COMPACT_BYTES: hi * COMPACT_BYTES]
            with torch.cuda.stream(s):
                c.classify_device(d_seqs.data_ptr() + lo * L, m * L, d_off.data_ptr(), m, out_view.data_ptr(),
                                  paired=args.paired, stream=s.cuda_stream)
                c.lca_device(dtax, out_view.data_ptr(), m, cview.data_ptr(), stream=s.cuda_stream)
                g.gather(cview)
            pending[k % nctx] = m
        for k in range(nctx):
            if pending[k] is not None:
                collect(clfs[k], pending[k], record)
        for s in streams[1:]:
            main.wait_stream(s)
        g.wait()

    for _ in range(args.warmup):
        one_step(False)
    kdist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step(True)
    kdist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    elapsed = kdist.max_over_ranks(elapsed, device=dev)

    # one more, untimed, pass with the chunks strictly one after the other: the kernel durations without
    # the other chunk's kernels competing for the CUs (reported next to the live figures)
    live = (list(kern_ms), dict(stage_ms))
    kern_ms.clear()
    for k in stage_ms:
        stage_ms[k] = 0.0
    one_step(True, 1)
    torch.cuda.synchronize(dev)
    excl_kern = list(kern_ms)
    kern_ms[:] = live[0]
    stage_ms.update(live[1])

    # sanity: the fraction of reads with a hit must be what the generator plants (70 % DB reads)
    hits = np.frombuffer(d_out[: min(n, 1_000_000) * HIT_BYTES].cpu().numpy().tobytes(), dtype=api.HIT_DTYPE)
    frac_hit = float((hits["n_ids"] > 0).mean())

    if rank != 0:
        return
    total_reads = n * world * args.steps
    value = total_reads / elapsed
    result = {
        "metric": "classified reads/sec (150 bp)" if not args.protein else f"classified protein reads/sec ({args.protein} aa)",
        "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": f"viruses-like synthetic index ({db.nseq} proteins, {db.total_aa} aa, .fmi "
                               f"{os.path.getsize(fmi)/1e6:.0f} MB, e=3); {n} synthetic {'2x' if args.paired else ''}{Lm}-{'aa protein' if args.protein else 'bp'} reads{' (pairs)' if args.paired else ''} per GPU per step "
                               f"(70% {'DB windows' if args.protein else 'back-translated DB windows'}, 30% random); kaiju {'-p ' if args.protein else ''}-a {args.mode} -m 11"
                               f"{'' if seg else ' -X'} (SEG {'on' if seg else 'off'})",
                   "reads_per_gpu_per_step": n, "chunk": chunk, "contexts_in_flight": nctx, "index_replicated": True,
                   "gather": ("one async RCCL gather of 16-B records (device LCA of the 184-B hit records) per chunk to rank 0"
                              if world > 1 else "none (1 GPU); device LCA to 16-B records still runs"),
                   "fraction_reads_with_hit": round(frac_hit, 4), "overflow_retries_per_step": retries / max(args.steps, 1)},
    }
    # ---------------- roofline of the dominant kernel + CPU baseline ----------------
    ops = None
    cb = None
    if not args.no_cpu_baseline and world == 1 and not args.paired and not args.protein:   # (the CPU leg and the op counts are for single reads)
        try:
            r = cpu_baseline(W, fmi, nodes, reads, args.mode, seg, args.cpu_sample, 30000)
            ops, cb = r["ops"], r["baseline"]
        except Exception as e:  # noqa: BLE001 - the baseline leg must never kill the measurement
            log(rank, "cpu baseline failed:", repr(e))
    if ops is None:
        # accounting figures measured with the instrumented oracle on this workload
        ops = ({"update_si": 461.0, "lf_steps": 5.9, "sa_decodes": 0.85, "sample": 0} if args.mode == "mem"
               else {"update_si": 1059.0, "lf_steps": 5.5, "sa_decodes": 0.8, "sample": 0})
    bytes_per_read = 128.0 * ops["update_si"] + 64.0 * ops["lf_steps"] + 8.0 * ops["sa_decodes"] + L + HIT_BYTES
    tot_ms = sum(ms for ms, _ in kern_ms)
    tot_reads = sum(m for _, m in kern_ms)
    avg_ms = tot_ms / max(len(kern_ms), 1)
    achieved = (bytes_per_read * tot_reads / max(len(kern_ms), 1)) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    # HBM bytes of one launch of that kernel as measured with rocprofv3 --pmc TCC_EA0_RDREQ_sum (x 128 B per
    # request, the gfx950 correction of the guide confirmed in profiles/r01_randbench_calibration.txt) and
    # TCC_EA0_WRREQ_sum (x 64 B) on this very workload; PMC passes cannot run inside the timed bench, so the
    # figure comes from the committed measurement and is only reported when the workload matches it
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            for rec in json.load(f)["measurements"]:
                if (rec["mode"] == args.mode and int(rec["seg"]) == int(seg) and rec["nseq"] == db.nseq and
                        rec["reads_per_launch"] == int(tot_reads_per_launch(kern_ms))):
                    traffic = rec["hbm_bytes_per_launch"]
    except Exception:  # noqa: BLE001
        traffic = None
    result["roofline"] = {"bound": "hbm", "kernel": "k_mem" if args.mode == "mem" else "k_greedy2",
                          "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                          "traffic": traffic,
                          "algorithmic_bytes_per_read": bytes_per_read,
                          "ops_per_read": ops, "reads_per_launch": tot_reads / max(len(kern_ms), 1),
                          "avg_launch_ms": avg_ms,
                          "exclusive": (lambda e: {"avg_launch_ms": e, "achieved": bytes_per_read * tot_reads_per_launch(excl_kern) / (e * 1e-3) / 1e9,
                                                   "frac": bytes_per_read * tot_reads_per_launch(excl_kern) / (e * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                   "note": "same kernel, chunks strictly one after the other (untimed extra pass)"})(
                              sum(ms for ms, _ in excl_kern) / max(len(excl_kern), 1)) if excl_kern else None,
                          "stage_ms_per_step": {k: v / max(args.steps, 1) for k, v in stage_ms.items()}}
    if args.protein:
        # no op counts exist for this workload yet: the accounting figures above are those of the 150-bp reads
        result["roofline"].update({"traffic": None, "achieved": None, "frac": None, "exclusive": None,
                                   "note": "op counts per read not measured for protein reads: stage times only"})
    if cb is not None:
        result["cpu_baseline"] = cb
    print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
