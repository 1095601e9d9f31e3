import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import util, pyoracle as po
from kaiju_amd import api
g = util.Golden()
orc = po.Oracle(); ix = orc.load_fmi(g.fmi); tax = orc.load_nodes(g.nodes)
idx = api.Index(g.fmi)
reads = util.long_reads()
seqs, off = util.pack(reads)
for mode in ("mem", "greedy"):
    for seg in (1, 0):
        clf = api.Classifier(idx, api.default_params(mode, seg=seg))
        hits = clf.classify(seqs, off)
        st = clf.stats()
        oh = orc.classify(ix, tax, orc.params(mode, seg=seg, use_evalue=0), seqs, off)
        bad = [i for i in range(len(oh)) if not util.same_hit(oh[i], hits[i])]
        print(mode, seg, "bad", len(bad), bad[:6], "retries", st.n_overflow_retries, "err", st.error_flags, "segfrags", st.n_seg_fragments, flush=True)
        for i in bad[:2]:
            print("  len", len(reads[i]), "oracle", oh[i]["best"], oh[i]["n_ids"], list(oh[i]["taxid"][:3]), "gpu", hits[i]["best"], hits[i]["n_ids"], list(hits[i]["taxid"][:3]), hex(int(hits[i]["flags"])))

# fragment lists: device (KAIJU_GPU_DUMP_FRAGS) vs host emulation
if os.environ.get("KAIJU_GPU_DUMP_FRAGS"):
    path = os.environ["KAIJU_GPU_DUMP_FRAGS"]
    if os.path.exists(path): os.remove(path)
    clf = api.Classifier(idx, api.default_params("mem", seg=0))
    hits = clf.classify(seqs, off)
    dev = open(path).read().split("#\n")[1:]
    emu = util.Emu(); h = emu.load(g.fmi)
    _, _, dump = emu.classify(h, util.gp("mem", seg=0), seqs, off, want_frags=True)
    em = dump.split("#\n")[1:]
    nb = 0
    for i, (a, b) in enumerate(zip(dev, em)):
        if a != b:
            nb += 1
            if nb <= 2:
                print("read", i, "len", len(reads[i])); print(" dev:", a.replace("\n", " | ")[:600]); print(" emu:", b.replace("\n", " | ")[:600])
    print("reads with different fragment lists:", nb, "of", len(em))
