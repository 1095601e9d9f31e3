import sys, ctypes as C, numpy as np, faulthandler
faulthandler.enable()
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import util
from kaiju_amd import api
g = util.Golden()
L = api.lib()
idx = api.Index(g.fmi)
for mode in ("mem", "greedy"):
    clf = api.Classifier(idx, api.default_params(mode, seg=1))
    n = len(g.reads)
    hits = np.zeros(n, dtype=api.HIT_DTYPE)
    VB = np.dtype([("n_acc", "<u4"), ("text_len", "<u4"), ("truncated", "<u4"), ("acc", "<u4", (20,))])
    v = np.zeros(n, dtype=VB)
    stride = 8193
    text = np.zeros(n * stride, dtype=np.uint8)
    L.kaiju_gpu_classify_batch_verbose.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    print("calling", mode, flush=True)
    rc = L.kaiju_gpu_classify_batch_verbose(clf._h, g.seqs.ctypes.data, g.off.ctypes.data, n, 0, hits.ctypes.data, v.ctypes.data, text.ctypes.data, stride)
    print("rc", rc, flush=True)
    L.kaiju_gpu_index_seq_name.restype = C.c_char_p
    L.kaiju_gpu_index_seq_name.argtypes = [C.c_void_p, C.c_uint32]
    for r in range(3):
        t = bytes(text[r*stride: r*stride+v[r]["text_len"]]).decode()
        print(g.names[r], hits[r]["best"], v[r]["n_acc"], [L.kaiju_gpu_index_seq_name(idx._h, int(a)) for a in v[r]["acc"][:v[r]["n_acc"]]], t)
