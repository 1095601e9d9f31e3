"""kaijux golden lines (reference binary oracle/_ref/kaijux, built by `make -C oracle ref`) on the committed
golden index and reads:  python tests/golden/make_golden_kaijux.py
  refx_<mode>[_pe][_v].tsv   `kaijux -z 1 [-v]` of the unmodified reference"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "kaijux")
for mode in ("mem", "greedy"):
    for pe in (False, True):
        for v in (False, True):
            out = f"{HERE}/refx_{mode}{'_pe' if pe else ''}{'_v' if v else ''}.tsv"
            cmd = [REF, "-f", f"{HERE}/db.fmi", "-a", mode, "-z", "1", "-o", out]
            cmd += ["-i", f"{HERE}/pairs_1.fq", "-j", f"{HERE}/pairs_2.fq"] if pe else ["-i", f"{HERE}/reads.fq"]
            if v:
                cmd.append("-v")
            subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
            print(out, sum(1 for _ in open(out)))
