"""What ending a process costs on the GPU box, by what it holds (external wall time of a child that allocates and _exits):
   exit_cost.py        -> table on stdout"""
import ctypes as C
import os
import subprocess
import sys
import time

if len(sys.argv) > 1:
    gb, host_gb, touch = float(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])
    t0 = time.time()
    hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
    n = C.c_int(0)
    hip.hipGetDeviceCount(C.byref(n))
    t1 = time.time()
    p = C.c_void_p()
    if gb > 0:
        hip.hipMalloc(C.byref(p), C.c_size_t(int(gb * 2**30)))
        if touch:
            hip.hipMemset(p, 0, C.c_size_t(int(gb * 2**30)))
            hip.hipDeviceSynchronize()
    t2 = time.time()
    if host_gb > 0:
        import mmap
        m = mmap.mmap(-1, int(host_gb * 2**30))
        step = 4096
        for o in range(0, len(m), step * 256):
            m[o:o + 1] = b"x"
        import numpy as np
        a = np.frombuffer(m, dtype=np.uint8)
        a[::4096] = 1
    t3 = time.time()
    sys.stderr.write("runtime start %.0f ms, device alloc %.0f ms, host alloc %.0f ms\n" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
    os._exit(0)

for gb, host, touch in [(0, 0, 0), (0, 0, 0), (1, 0, 1), (11, 0, 0), (11, 0, 1), (11, 2, 1), (0, 2, 0)]:
    t = time.time()
    r = subprocess.run([sys.executable, __file__, str(gb), str(host), str(touch)], capture_output=True, text=True)
    dt = time.time() - t
    print("device %4.0f GB (touched %d), host %3.0f GB: %.3f s wall | %s" % (gb, touch, host, dt, r.stderr.strip()), flush=True)
