import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "oracle")); sys.path.insert(0, R)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, util, pyoracle as po
from kaiju_amd import api
g = util.Golden(); O = po.Oracle(); oix, otax = O.load_fmi(g.fmi), O.load_nodes(g.nodes)
idx = api.Index(g.fmi)
for mode in ("mem", "greedy"):
    for verbose in (False, True):
        clf = api.Classifier(idx, api.default_params(mode, seg=1))
        print(mode, verbose, "start", flush=True)
        if verbose: hits, accs, peps = clf.classify_verbose(g.pseqs, g.poff, paired=True)
        else: hits = clf.classify(g.pseqs, g.poff, paired=True)
        oh = O.classify(oix, otax, O.params(mode, seg=1, use_evalue=0), g.pseqs, g.poff, paired=True)
        bad = [i for i in range(len(oh)) if not util.same_hit(oh[i], hits[i])]
        print(mode, verbose, "bad", bad[:10], clf.stats().error_flags, flush=True)
