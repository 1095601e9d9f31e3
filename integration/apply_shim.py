#!/usr/bin/env python3
"""Build the reference's `kaiju` with its ConsumerThread loop replaced by the C-ABI shim.

    python3 integration/apply_shim.py [--ref /root/reference] [--out oracle/_ref/kaiju_gpu_shim]

The reference's sources are copied to a scratch directory OUTSIDE this repository, three small edits are
made there (listed in EDITS below, each anchored on a line that must occur exactly once), our
integration/ConsumerThread_gpu.cpp is compiled with them, and the result is linked against
kaiju_amd/libkaiju_gpu.so.  The scratch directory is removed afterwards; only the binary stays
(oracle/_ref/ is git-ignored and travels to the GPU box with gpurun like the other prebuilt files)."""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (file, anchor that must occur exactly once, replacement)
EDITS = [
    # Config gets the handle of the index in HBM
    ("Config.hpp", "\t\tAlphabetStruct * astruct;\n",
     "\t\tAlphabetStruct * astruct;\n\t\tstruct kaiju_gpu_index * gpu_index = nullptr;\n"),
    # kaiju.cpp: after readFMI() + config->init() the index goes to the GPU
    ("kaiju.cpp", "\tconfig->init();\n", "\tconfig->init();\n\tkaiju_gpu_attach(config);\n"),
    ("kaiju.cpp", "#include \"ConsumerThread.hpp\"\n", "#include \"ConsumerThread.hpp\"\nvoid kaiju_gpu_attach(Config *);\n"),
    # the CPU loop stays in the binary under another name; doWork() is ConsumerThread_gpu.cpp's
    ("ConsumerThread.cpp", "void ConsumerThread::doWork() {\n", "void ConsumerThread::doWork_cpu() {\n"),
    ("ConsumerThread.hpp", "\tvoid doWork();\n", "\tvoid doWork();\n\tvoid doWork_cpu();\n"),
]

BLAST_C = """algo/blast/core/pattern.c algo/blast/core/blast_posit.c
algo/blast/composition_adjustment/matrix_frequency_data.c algo/blast/core/blast_dynarray.c
algo/blast/core/matrix_freq_ratios.c algo/blast/core/blast_encoding.c algo/blast/core/blast_stat.c
algo/blast/core/blast_filter.c algo/blast/core/blast_util.c algo/blast/core/blast_message.c
algo/blast/core/ncbi_erf.c algo/blast/core/blast_options.c algo/blast/core/ncbi_math.c
algo/blast/core/blast_program.c algo/blast/core/ncbi_std.c algo/blast/core/blast_psi_priv.c
util/tables/raw_scoremat.c algo/blast/core/blast_query_info.c algo/blast/core/blast_seg.c""".split()


def run(cmd, **kw):
    subprocess.run(cmd, check=True, **kw)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "oracle", "_ref", "kaiju_gpu_shim"))
    args = ap.parse_args()
    lib = os.path.join(ROOT, "kaiju_amd", "libkaiju_gpu.so")
    if not os.path.exists(lib):
        raise SystemExit("build kaiju_amd/libkaiju_gpu.so first (python -m kaiju_amd.build)")
    out = os.path.abspath(args.out)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    scratch = tempfile.mkdtemp(prefix="kaiju_shim_")
    try:
        src = os.path.join(scratch, "src")
        shutil.copytree(os.path.join(args.ref, "src"), src)
        for fn, anchor, repl in EDITS:
            p = os.path.join(src, fn)
            s = open(p).read()
            if s.count(anchor) != 1:
                raise SystemExit(f"{fn}: anchor {anchor!r} occurs {s.count(anchor)} times, expected 1")
            open(p, "w").write(s.replace(anchor, repl))
        shutil.copy(os.path.join(ROOT, "integration", "ConsumerThread_gpu.cpp"), src)
        inc = ["-I" + os.path.join(src, "include"), "-I" + os.path.join(src, "include", "ncbi-blast+"), "-I" + src,
               "-I" + os.path.join(ROOT, "include")]
        objs = []
        jobs = []
        for c in BLAST_C:
            o = os.path.join(scratch, "o_" + c.replace("/", "_") + ".o")
            jobs.append(["gcc", "-O3", "-DNDEBUG", "-w"] + inc + ["-c", "-o", o, os.path.join(src, "include", "ncbi-blast+", c)])
            objs.append(o)
        for c in ("bwt.c", "compactfmi.c", "sequence.c", "suffixArray.c"):
            o = os.path.join(scratch, "o_bwt_" + c + ".o")
            jobs.append(["gcc", "-O3", "-DNDEBUG", "-w", "-c", "-o", o, os.path.join(src, "bwt", c)])
            objs.append(o)
        for c in ("kaiju.cpp", "ReadItem.cpp", "Config.cpp", "ConsumerThread.cpp", "util.cpp", "ConsumerThread_gpu.cpp"):
            o = os.path.join(scratch, "o_" + c + ".o")
            jobs.append(["g++", "-O3", "-pthread", "-std=c++11", "-DNDEBUG", "-w"] + inc + ["-c", "-o", o, os.path.join(src, c)])
            objs.append(o)
        procs = [subprocess.Popen(j) for j in jobs]
        for p, j in zip(procs, jobs):
            if p.wait() != 0:
                raise SystemExit("failed: " + " ".join(j))
        # $ORIGIN/../../kaiju_amd: oracle/_ref/kaiju_gpu_shim finds the library of this checkout wherever it is copied
        run(["g++", "-o", out] + objs + ["-L" + os.path.dirname(lib), "-lkaiju_gpu", "-lpthread", "-lz",
                                         "-Wl,-rpath,$ORIGIN/../../kaiju_amd"])
        print("built", out)
    finally:
        shutil.rmtree(scratch, ignore_errors=True)


if __name__ == "__main__":
    sys.exit(main())
