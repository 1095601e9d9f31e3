"""why does ncclCommInitRank fail?  probe: the library's communicator of one rank, with and without torch in the process"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if "torch" in sys.argv:
    import torch
    torch.cuda.init()
    print("torch", torch.__version__, torch.cuda.is_available(), flush=True)
from kaiju_amd import api  # noqa: E402

print("devices", api.device_count(), flush=True)
try:
    c = api.Comm(f"/tmp/probe_{os.getpid()}.id", 0, 1, 0)
    print("comm ok", flush=True)
    c.close()
except Exception as e:  # noqa: BLE001
    print("comm failed:", e, flush=True)
os.system("grep -i rccl /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid())
