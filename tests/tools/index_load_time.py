"""index load: .fmi (parse + pack + upload) vs device image (read + upload), GPU box: index_load_time.py <workdir>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kaiju_amd import api
W = sys.argv[1]
t = time.time(); api.write_index_image(f"{W}/db.fmi", f"{W}/db.kjimg"); print(f"write image: {time.time()-t:.2f} s, {os.path.getsize(W+'/db.kjimg')/1e6:.0f} MB")
for rep in range(2):
    t = time.time(); a = api.Index(f"{W}/db.fmi"); t1 = time.time() - t; del a
    t = time.time(); b = api.Index(f"{W}/db.kjimg"); t2 = time.time() - t; del b
    print(f"load .fmi {t1:.2f} s, load image {t2:.2f} s")
