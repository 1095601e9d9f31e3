"""An index that NEEDS the wide path: >= 2^32 rows (refseq class, bwt.c:105-121 walks 64-bit rows there), built on the
GPU box from a synthetic protein database (tests/tools/gen_db.c, the recipe of SURVEY.md 8d) with kaiju_build_fmi.

   wide_index.py prepare <dir> [nseq]     database, .fmi, nodes.dmp, reads (the expensive part, once)
   wide_index.py parity  <dir>            20 k-read samples vs the oracle: MEM, Greedy, pairs (also tests/test_gpu_wide.py)
   wide_index.py bench   <dir> [out.json] the legs of bench.py (MEM, Greedy, pairs) on this index + reference baseline

No KAIJU_GPU_FORCE_WIDE anywhere: the loader picks the wide layout because the index has 2^32 rows or more."""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from kaiju_amd import api, mkfmi, synth  # noqa: E402


def load_db(W):
    codes = np.fromfile(f"{W}/db.codes", dtype=np.uint8)
    offsets = np.fromfile(f"{W}/db.offsets", dtype=np.int64)
    taxids = np.fromfile(f"{W}/db.taxids", dtype=np.int64)
    return synth.SynthDB(codes=codes, offsets=offsets, taxids=taxids, names=None)


def prepare(W, nseq):
    os.makedirs(W, exist_ok=True)
    t0 = time.time()
    exe = f"{W}/gen_db"
    subprocess.run(["gcc", "-O2", "-o", exe, os.path.join(ROOT, "tests", "tools", "gen_db.c"), "-lm"], check=True)
    subprocess.run([exe, str(nseq), "20260926", f"{W}/db.faa", f"{W}/db.codes", f"{W}/db.offsets", f"{W}/db.taxids"], check=True)
    print(f"[wide] database written ({time.time()-t0:.1f}s)", flush=True)
    t0 = time.time()
    mkfmi.build_fmi(f"{W}/db.faa", f"{W}/db.fmi", threads=0, exponent=3)
    os.remove(f"{W}/db.faa")
    print(f"[wide] .fmi built: {os.path.getsize(f'{W}/db.fmi')/1e9:.2f} GB ({time.time()-t0:.1f}s)", flush=True)
    lines, _ = synth.make_taxonomy()
    synth.write_nodes_dmp(f"{W}/nodes.dmp", lines)
    db = load_db(W)
    t0 = time.time()
    np.save(f"{W}/reads.npy", synth.make_reads(db, 3_000_000, seed=777))
    m1, m2 = synth.make_pairs(db, 1_500_000, seed=778)
    np.save(f"{W}/pairs.npy", np.concatenate([m1, m2], axis=1))
    print(f"[wide] reads generated ({time.time()-t0:.1f}s)", flush=True)


def parity(W, sample=20000):
    import pyoracle as po
    import util
    t0 = time.time()
    index = api.Index(f"{W}/db.fmi")
    print(f"[wide] index in HBM: bwtlen {index.info.bwtlen} ({index.info.bwtlen / 2**32:.3f} x 2^32), {index.info.device_bytes/1e9:.1f} GB, "
          f"warnings {index.info.warnings} ({time.time()-t0:.1f}s)", flush=True)
    assert index.info.bwtlen >= 2 ** 32, "this index does not need the wide path"
    O = po.Oracle()
    t0 = time.time()
    oix, otax = O.load_fmi(f"{W}/db.fmi"), O.load_nodes(f"{W}/nodes.dmp")
    print(f"[wide] oracle loaded the .fmi ({time.time()-t0:.1f}s)", flush=True)
    reads = np.load(f"{W}/reads.npy", mmap_mode="r")[:sample]
    pairs = np.load(f"{W}/pairs.npy", mmap_mode="r")[:sample]
    res = {}
    for name, mode, rd, pe in (("mem", "mem", reads, False), ("greedy", "greedy", reads, False), ("mem_pairs", "mem", pairs, True),
                               ("greedy_pairs", "greedy", pairs[: sample // 4], True)):
        rd = np.ascontiguousarray(rd)
        if pe:
            seqs, off = synth.pack_reads(rd[:, :150], rd[:, 150:])
        else:
            seqs, off = synth.pack_reads(rd)
        clf = api.Classifier(index, api.default_params(mode, seg=1))
        hits = clf.classify(seqs, off, paired=pe)
        st = clf.stats()
        t0 = time.time()
        oh = O.classify(oix, otax, O.params(mode, seg=1, use_evalue=0), seqs, off, paired=pe)
        bad = [i for i in range(len(oh)) if not util.same_hit(oh[i], hits[i])]
        res[name] = {"checked": len(oh), "mismatches": len(bad), "with_hit": int((hits["n_ids"] > 0).sum()),
                     "error_flags": int(st.error_flags), "retries": int(st.n_overflow_retries)}
        print(f"[wide] parity {name}: {res[name]} (oracle {time.time()-t0:.1f}s)", flush=True)
        clf.close()
    return res


def bench(W, out=None):
    import importlib.util
    import torch
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    B = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(B)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    index = api.Index(f"{W}/db.fmi")
    dtax = api.DeviceTaxonomy(api.Taxonomy(f"{W}/nodes.dmp"), 0)
    db_off = np.fromfile(f"{W}/db.offsets", dtype=np.int64)
    nseq = len(db_off) - 1
    reads = np.load(f"{W}/reads.npy")
    pairs = np.load(f"{W}/pairs.npy")
    result = {"index": {"bwtlen": int(index.info.bwtlen), "rows_over_2_32": index.info.bwtlen / 2 ** 32, "nseq": nseq,
                        "fmi_bytes": os.path.getsize(f"{W}/db.fmi"), "hbm_bytes": int(index.info.device_bytes),
                        "footprint": index.footprint.as_dict() if hasattr(index, "footprint") else None}}
    legs = os.environ.get("WIDE_LEGS", "mem,greedy,paired").split(",")
    for name, mode, rd, pe in (("mem", "mem", reads, False), ("greedy", "greedy", reads, False), ("paired", "mem", pairs, True)):
        if name not in legs:
            continue
        leg = B.Leg(name, mode, pe, rd, 150, index, dtax, dev, 0, 1, 1, 1_500_000, 0)
        leg.run(2, 1)
        ref_ops = None
        bl = par = None
        try:
            if os.environ.get("WIDE_NO_REF"):
                raise RuntimeError("reference leg switched off (WIDE_NO_REF)")
            blr, ref = B.run_reference(W, f"{W}/db.fmi", f"{W}/nodes.dmp", rd, 150, pe, mode, 1, 200000)
            bl = blr
            if ref is not None:
                cls, tax, rec = leg.host_records(len(ref[0]))
                best = rec["best"].astype(np.int64)
                badidx = np.nonzero((cls != ref[0]) | (tax != ref[1]) | ((ref[0] != 0) & (best != ref[2])))[0]
                par = {"checked": int(len(ref[0])), "mismatches": int(len(badidx))}
        except Exception as e:  # noqa: BLE001
            print("[wide] reference leg failed:", repr(e), flush=True)
        r = leg.result(1, ref_ops, None, nseq)
        if bl:
            r["cpu_baseline"] = bl
        if par:
            r["parity"] = par
        result[name] = r
        print(f"[wide] {name}: {r['value']/1e6:.1f} M {r['unit']}, kernel {r['roofline']['kernel']} {r['roofline']['avg_launch_ms']:.2f} ms per "
              f"{int(r['roofline']['units_per_launch'])}, frac {r['roofline']['frac']:.3f}, parity {par}", flush=True)
        leg.close()
    if out:
        with open(out, "w") as f:
            json.dump(result, f, indent=1)
    return result


if __name__ == "__main__":
    cmd, W = sys.argv[1], sys.argv[2]
    if cmd == "prepare":
        prepare(W, int(sys.argv[3]) if len(sys.argv) > 3 else 15_500_001)
    elif cmd == "parity":
        r = parity(W)
        sys.exit(1 if any(v["mismatches"] for v in r.values()) else 0)
    elif cmd == "bench":
        bench(W, sys.argv[3] if len(sys.argv) > 3 else None)
