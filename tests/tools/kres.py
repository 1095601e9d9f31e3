"""register / scratch / LDS use of every kernel of capi.hip as the compiler reports it (gfx950 cross-compile, no GPU needed)
   python tests/tools/kres.py [extra -D flags ...] [--filter REGEX]"""
import os
import re
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
args = sys.argv[1:]
flt = None
if "--filter" in args:
    i = args.index("--filter")
    flt = re.compile(args[i + 1])
    del args[i:i + 2]
src = os.path.join(R, "kaiju_amd", "csrc", "capi.hip")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-gpu-rdc", "-w"] + args + \
      ["-c", src, "-o", "/tmp/kres_capi.o", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True).stderr
cur = {}
for line in out.splitlines():
    m = re.search(r"remark: [^ ]* *(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|VGPRs Spill|SGPRs Spill): (\S+)", line)
    if not m:
        if "error" in line:
            print(line)
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
    cur[k] = v
    if k.startswith("LDS"):
        n = re.sub(r"^_Z\d+", "", cur["name"])[:30]
        if flt is None or flt.search(n):
            print(f"{n:30s} VGPR {cur.get('VGPRs'):>4s} AGPR {cur.get('AGPRs', '0'):>3s} SGPR {cur.get('TotalSGPRs', '?'):>4s} (spilt {cur.get('SGPRs Spill', '?')}) "
                  f"scratch {cur.get('ScratchSize [bytes/lane]'):>5s} (VGPR spill {cur.get('VGPRs Spill', '?')}) occ {cur.get('Occupancy [waves/SIMD]'):>2s} LDS {v}")
