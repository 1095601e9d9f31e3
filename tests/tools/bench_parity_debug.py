"""debug: reference lines vs device compact records on a few reads of the benchmark workload"""
import os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import importlib.util
spec = importlib.util.spec_from_file_location('bench', os.path.join(ROOT, 'bench.py')); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
from kaiju_amd import api, synth
import torch
W = "/tmp/kaiju_amd_bench"
lines, leaves = synth.make_taxonomy()
db = synth.make_db(nseq=680001, seed=12345, leaves=leaves)
fmi, nodes = f"{W}/db_680001.fmi", f"{W}/nodes.dmp"
reads = synth.make_reads(db, 20000, seed=777)
bl, ref = b.run_reference(W, fmi, nodes, reads, 150, False, "mem", 1, 20000)
index = api.Index(fmi); tax = api.Taxonomy(nodes); dtax = api.DeviceTaxonomy(tax)
clf = api.Classifier(index, api.default_params("mem"))
seqs, off = synth.pack_reads(reads)
hits = clf.classify(seqs, off)
res = clf.finalize(tax, hits, off)
rec = clf.lca(dtax, hits)
res2 = clf.finalize_compact(rec, off)
print("host finalize vs ref mismatches", int(((res["classified"] != ref[0]) | (res["taxon"] != ref[1])).sum()))
print("compact finalize vs ref mismatches", int(((res2["classified"] != ref[0]) | (res2["taxon"] != ref[1])).sum()))
# device path as in bench
dev = torch.device("cuda", 0)
leg = b.Leg("dbg", "mem", False, reads, 150, index, dtax, dev, 0, 1, 1, 5000000, 1)
leg.step(False)
torch.cuda.synchronize()
c, t, r = leg.host_records(20000)
bad = np.nonzero((c != ref[0]) | (t != ref[1]))[0]
print("device leg vs ref mismatches", len(bad))
for i in bad[:5]:
    print(i, "ref", ref[0][i], ref[1][i], "gpu", c[i], t[i], r[i], "host", res[i])
leg.clfs[0].count_ops(True); leg.step(False); torch.cuda.synchronize(); leg.clfs[0].count_ops(False)
c, t, r = leg.host_records(20000)
bad = np.nonzero((c != ref[0]) | (t != ref[1]))[0]
print("count kernel vs ref mismatches", len(bad))
