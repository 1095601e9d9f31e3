"""The randomised parity hunt of fuzz_emu.py through the HIP path (GPU box): fuzz_gpu.py [rounds] [seed]"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import util  # noqa: E402
import pyoracle as po  # noqa: E402
from kaiju_amd import api, mkfmi  # noqa: E402
from fuzz_emu import make_db, make_read  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
orc = po.Oracle()
total = 0
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
for rnd in range(first, first + rounds):
    rng = np.random.default_rng(seed * 1000 + rnd)
    with tempfile.TemporaryDirectory() as d:
        faa, fmi, nodes = f"{d}/db.faa", f"{d}/db.fmi", f"{d}/nodes.dmp"
        seqs = make_db(rng, faa, nodes)
        mkfmi.build_fmi(faa, fmi, threads=4, exponent=int(rng.choice([1, 3, 5])))
        idx = api.Index(fmi)
        if idx.info.warnings:
            print(f"round {rnd}: index hits a latent bug of the reference (parity undefined there), skipped", flush=True)
            continue
        ix = orc.load_fmi(fmi); tax = orc.load_nodes(nodes)
        n = int(rng.integers(50, 400))
        r1 = [make_read(rng, seqs) for _ in range(n)]
        paired = rng.random() < 0.4
        r2 = [make_read(rng, seqs) for _ in range(n)] if paired else None
        sq, off = util.pack(r1, r2)
        for mode in ("mem", "greedy"):
            kw = dict(seg=int(rng.integers(0, 2)))
            if mode == "mem":
                kw["min_fragment_length"] = int(rng.choice([7, 9, 11, 11, 15, 20]))
            else:
                kw["mismatches"] = int(rng.choice([0, 1, 3, 3, 5])); kw["min_score"] = int(rng.choice([30, 65, 65, 90]))
                kw["seed_length"] = int(rng.choice([7, 7, 8, 10])); kw["min_fragment_length"] = int(rng.choice([9, 11, 11, 13]))
            p = api.default_params(mode, seg=kw["seg"])
            for k, v in kw.items():
                if k != "seg":
                    setattr(p, k, v)
            oh = orc.classify(ix, tax, orc.params(mode, use_evalue=0, **kw), sq, off, paired=paired)
            clf = api.Classifier(idx, p)
            gh = clf.classify(sq, off, paired=paired)
            st = clf.stats()
            bad = [i for i in range(n) if not util.same_hit(oh[i], gh[i])]
            total += n
            if bad or st.error_flags:
                print("MISMATCH round", rnd, "seed", seed, mode, kw, "paired", paired, "reads", bad[:5], "err", st.error_flags, flush=True)
                i = bad[0]
                print("  oracle", oh[i]["best"], oh[i]["n_ids"], list(oh[i]["taxid"][:4]), "gpu", gh[i]["best"], gh[i]["n_ids"], list(gh[i]["taxid"][:4]), hex(int(gh[i]["flags"])))
                print("  read", r1[i][:150], (r2[i][:60] if paired else b""))
                sys.exit(1)
        del clf, idx
    print(f"round {rnd}: ok ({total})", flush=True)
print("FUZZ_OK", total)
