"""Build the HIP library in-tree: kaiju_amd/libkaiju_gpu.so (gfx950).

hipcc cross-compiles without a GPU, so this runs in the build container as well as on the
GPU box.  The library is the product: kernels + C-ABI (include/kaiju_gpu.h)."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libkaiju_gpu.so")
SOURCES = ["capi.hip", "fmi_stream.hip", "exact_pass.hip", "host_index.cpp", "host_tables.cpp", "taxonomy.cpp", "rccl_gather.cpp"]
HEADERS = ["kj_core.h", "kj_greedy3.h", "fmi_stream.h", "exact_pass.h", "host_index.h", "host_tables.h", os.path.join("..", "..", "include", "kaiju_gpu.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-gpu-rdc", "-Wno-unused-result"]


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the Kaiju GPU library cannot be built")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


MKFMI_LIB = os.path.join(HERE, "libkaiju_mkfmi.so")
MKFMI_SRC = [os.path.join(CSRC, "mkfmi.cpp"), os.path.join(CSRC, "mkfmi.h")]


def build_mkfmi(force=False, verbose=False):
    """the index builder: test / benchmark infrastructure in a library of its own (host only; csrc/mkfmi.h says why it exists)"""
    if force or not os.path.exists(MKFMI_LIB) or any(os.path.getmtime(d) > os.path.getmtime(MKFMI_LIB) for d in MKFMI_SRC):
        cmd = ["g++", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result", "-o", MKFMI_LIB, MKFMI_SRC[0], "-lpthread"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return MKFMI_LIB


def build(force=False, verbose=False):
    build_mkfmi(force=force, verbose=verbose)
    if force or needs_build():
        cmd = [hipcc_path()] + FLAGS + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES] + ["-lpthread", "-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    build_cli(force=force, verbose=verbose)
    return LIB


CLI_SRC = os.path.join(CSRC, "host", "kaiju_main.cpp")
CLI = os.path.join(HERE, "bin", "kaiju")


def build_cli(force=False, verbose=False):
    """the drop-in `kaiju` command (host C++ over the C-ABI)"""
    multi = os.path.join(os.path.dirname(CLI), "kaiju-multi")
    if (not force and os.path.exists(CLI) and os.path.exists(multi) and os.path.exists(os.path.join(os.path.dirname(CLI), "kaijux")) and
            os.path.exists(os.path.join(os.path.dirname(CLI), "kaijup")) and
            os.path.getmtime(CLI) >= max(os.path.getmtime(CLI_SRC), os.path.getmtime(LIB))):
        return CLI
    os.makedirs(os.path.dirname(CLI), exist_ok=True)
    cmd = ["g++", "-O2", "-std=c++17", "-o", CLI, CLI_SRC, "-L" + HERE, "-lkaiju_gpu", "-lz", "-lpthread",
           "-Wl,-rpath,$ORIGIN/.."]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    # kaiju-multi is the same program (it looks at its name): comma separated file lists, one index load
    import shutil as _sh
    _sh.copy2(CLI, multi)
    _sh.copy2(CLI, os.path.join(os.path.dirname(CLI), "kaijux"))      # kaijux: database sequences instead of taxa
    _sh.copy2(CLI, os.path.join(os.path.dirname(CLI), "kaijup"))      # kaijup: kaijux for protein reads
    return CLI


if __name__ == "__main__":
    build(force=True, verbose=True)
