"""Developer check run on the GPU box: build a synthetic index with the reference tools in
oracle/_ref, classify on the GPU through the C-ABI and compare with the CPU oracle.

  python tests/tools/gpu_check.py --nseq 20001 --reads 20000 --check 20000
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from kaiju_amd import api, synth  # noqa: E402
import pyoracle as po  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nseq", type=int, default=20001)
    ap.add_argument("--reads", type=int, default=20000)
    ap.add_argument("--check", type=int, default=20000)
    ap.add_argument("--modes", default="mem,greedy")
    ap.add_argument("--segs", default="1,0")
    ap.add_argument("--work", default="/tmp/kaiju_gpu_check")
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    W = args.work
    os.makedirs(W, exist_ok=True)
    t = time.time()
    lines, leaves = synth.make_taxonomy()
    synth.write_nodes_dmp(f"{W}/nodes.dmp", lines)
    db = synth.make_db(nseq=args.nseq, seed=12345, leaves=leaves)
    print(f"db: {db.nseq} seqs, {db.total_aa} aa  ({time.time()-t:.1f}s)", flush=True)
    fmi = f"{W}/db_{args.nseq}.fmi"
    if not os.path.exists(fmi):
        t = time.time()
        synth.write_fasta(db, f"{W}/db.faa")
        po.ref_build_index(f"{W}/db.faa", f"{W}/db_{args.nseq}", threads=os.cpu_count() or 8)
        print(f"reference mkbwt+mkfmi: {time.time()-t:.1f}s", flush=True)
    t = time.time()
    reads = synth.make_reads(db, args.reads, seed=777)
    seqs, off = synth.pack_reads(reads)
    print(f"reads: {len(reads)} ({time.time()-t:.1f}s)", flush=True)
    t = time.time()
    index = api.Index(fmi)
    print(f"index on GPU: {index.info.device_bytes/1e6:.1f} MB, warnings={index.info.warnings} ({time.time()-t:.1f}s)", flush=True)
    O = po.Oracle()
    oix = O.load_fmi(fmi)
    otax = O.load_nodes(f"{W}/nodes.dmp")
    tax = api.Taxonomy(f"{W}/nodes.dmp")
    nchk = min(args.check, len(reads))
    ok = True
    for mode in args.modes.split(","):
        for seg in [int(x) for x in args.segs.split(",")]:
            p = api.default_params(mode, seg=seg)
            clf = api.Classifier(index, p)
            hits = clf.classify(seqs, off)
            best_ms = 1e9
            for _ in range(args.reps):
                hits = clf.classify(seqs, off)
                st = clf.stats()
                best_ms = min(best_ms, st.ms_total)
            st = clf.stats()
            res = clf.finalize(tax, hits, off)
            op = O.params(mode, seg=seg)
            t = time.time()
            s2, o2 = synth.pack_reads(reads[:nchk])
            O.counters(reset=True)
            oh = O.classify(oix, otax, op, s2, o2)
            cnt = O.counters(reset=True)
            dt = time.time() - t
            bad = 0
            for i in range(nchk):
                a, b, r = oh[i], hits[i], res[i]
                same = (int(a["classified"]) == int(r["classified"]) and int(a["lca"]) == int(r["taxon"]))
                if a["classified"]:
                    same = same and int(a["best"]) == int(b["best"]) and \
                        list(a["taxid"][:a["n_ids"]]) == list(b["taxid"][:b["n_ids"]])
                if not same:
                    bad += 1
                    if bad <= 3:
                        print("  MISMATCH", i, a, b, r)
            ok = ok and bad == 0
            nC = int(res["classified"].sum())
            print(f"{mode} seg={seg}: gpu {best_ms:.2f} ms/batch ({len(reads)/best_ms*1e3:,.0f} reads/s; "
                  f"translate {st.ms_translate:.2f} seg {st.ms_seg:.2f} [{st.n_seg_fragments} frags] "
                  f"search {st.ms_search:.2f} retry {st.ms_retry:.2f}) retries={st.n_overflow_retries} err={st.error_flags} "
                  f"classified={nC}  oracle {nchk/dt:,.0f} reads/s ({cnt['update_si']/nchk:.0f} UpdateSI/read, "
                  f"{cnt['fmindex_current']/nchk:.1f} LF/read)  mismatches={bad}/{nchk}", flush=True)
            clf.close()
    print("PARITY_OK" if ok else "PARITY_FAIL")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
