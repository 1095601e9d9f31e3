#!/usr/bin/env python3
"""kaiju -v (MEM and Greedy), host emulation: columns 6 / 7 from the second-generation lanes (VERBOSE instantiations of mem_lane2 / greedy_lane2 +
mem_verbose_read) against the first-generation lanes' (pinned on the reference's lines by test_verbose_columns), on the
random databases / reads / parameters of fuzz_emu.py.  Accessions compare as sets (the host sorts them), peptides byte for
byte.  usage: fuzz_verbose.py [rounds] [seed] [first]; KAIJU_GPU_FORCE_WIDE=16 in the environment: the wide lanes;
FUZZ_VARIANT=kaijux: the list order of maxMatches(.., 1)"""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import util  # noqa: E402
import fuzz_emu as fz  # noqa: E402
from kaiju_amd import mkfmi  # noqa: E402


def run(emu, h, gp, sq, off, paired, n, cap, v1):
    E = emu.lib
    E.emu_set_verbose.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    nacc = np.zeros(n, dtype=np.uint32); acc = np.zeros(n * 20, dtype=np.uint32)
    tlen = np.zeros(n, dtype=np.uint32); text = np.zeros(n * cap, dtype=np.uint8)
    if v1:
        os.environ["KAIJU_EMU_VERBOSE_V1"] = "1"
    else:
        os.environ.pop("KAIJU_EMU_VERBOSE_V1", None)
    E.emu_set_verbose(nacc.ctypes.data, acc.ctypes.data, tlen.ctypes.data, text.ctypes.data, cap)
    try:
        gh, _ = emu.classify(h, gp, sq, off, paired=paired, allow_capacity=True)
    finally:
        E.emu_set_verbose(None, None, None, None, 0)
        os.environ.pop("KAIJU_EMU_VERBOSE_V1", None)
    return gh, nacc, acc, tlen, text


def main(rounds=None, seed=None, first=None):
    variant = os.environ.get("FUZZ_VARIANT", "kaiju")
    xmode = variant == "kaijux"
    rounds = rounds if rounds is not None else (int(sys.argv[1]) if len(sys.argv) > 1 else 20)
    seed = seed if seed is not None else (int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    first = first if first is not None else (int(sys.argv[3]) if len(sys.argv) > 3 else 0)
    emu = util.Emu()
    total = 0
    cap = 4096
    for rnd in range(first, first + rounds):
        rng = np.random.default_rng(seed * 1000 + rnd)
        with tempfile.TemporaryDirectory() as d:
            faa, fmi, nodes = f"{d}/db.faa", f"{d}/db.fmi", f"{d}/nodes.dmp"
            seqs = fz.make_db(rng, faa, nodes)
            mkfmi.build_fmi(faa, fmi, threads=2, exponent=int(rng.choice([1, 3, 5])))
            if xmode:
                emu.lib.emu_index_load_x.restype = C.c_void_p
                emu.lib.emu_index_load_x.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
                err = C.create_string_buffer(256)
                h = emu.lib.emu_index_load_x(fmi.encode(), err, 256)
                assert h, err.value
            else:
                h = emu.load(fmi)
            if emu.lib.emu_index_warnings(h):
                emu.lib.emu_index_free(h)
                continue
            n = int(rng.integers(50, 400))
            r1 = [fz.make_read(rng, seqs) for _ in range(n)]
            paired = rng.random() < 0.4
            r2 = [fz.make_read(rng, seqs) for _ in range(n)] if paired else None
            sq, off = util.pack(r1, r2)
            for seg, mode in ((0, "mem"), (1, "mem"), (int(rng.integers(0, 2)), "greedy"), (int(rng.integers(0, 2)), "greedy")):
                if mode == "mem":
                    m = int(rng.choice([7, 9, 11, 11, 15, 20]))
                    gp = util.gp("mem", m=m, seg=seg)
                else:
                    m = int(rng.choice([9, 11, 11, 13]))
                    gp = util.gp("greedy", m=m, mismatches=int(rng.choice([0, 1, 3, 3, 5])), min_score=int(rng.choice([30, 65, 65, 90])),
                                 seed_length=int(rng.choice([7, 7, 8, 10])), seg=seg)
                a = run(emu, h, gp, sq, off, paired, n, cap, v1=True)
                b = run(emu, h, gp, sq, off, paired, n, cap, v1=False)
                if a[0] is None or b[0] is None:
                    continue
                for r in range(n):
                    if int(a[0][r]["flags"]) & 0x80000000:
                        continue
                    total += 1
                    ok = util.same_hit(a[0][r], b[0][r])
                    sa = set(a[2][r * 20: r * 20 + int(a[1][r])].tolist()); sb = set(b[2][r * 20: r * 20 + int(b[1][r])].tolist())
                    ta = bytes(a[4][r * cap: r * cap + min(cap, int(a[3][r]))]); tb = bytes(b[4][r * cap: r * cap + min(cap, int(b[3][r]))])
                    if not ok or sa != sb or ta != tb:
                        print("MISMATCH round", rnd, "seed", seed, mode, "seg", seg, "m", m, "paired", paired, "read", r, flush=True)
                        print("  v1", a[0][r]["best"], sorted(sa), ta)
                        print("  v2", b[0][r]["best"], sorted(sb), tb)
                        print("  read", r1[r][:150])
                        return 1
            emu.lib.emu_index_free(h)
        print(f"round {rnd}: ok ({total} verbose reads compared so far)", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
