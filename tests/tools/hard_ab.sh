#!/bin/bash
# the hostile legs of bench.py (and the headline, for the i.i.d. side) with this library and with KAIJU_AB_LIB
O=${1:-gpurun_out/hard_ab}; mkdir -p $O
ARGS="--legs hard --steps 3 --leg-steps 3 --no-cpu-baseline"
timeout 900 python bench.py $ARGS > $O/bench_new.json 2> $O/bench_new.err; echo "new rc=$?"
OLD=${KAIJU_AB_LIB:-kaiju_amd/variants/libkaiju_gpu_base.so}
[ -f $OLD ] && { KAIJU_GPU_LIB=$OLD timeout 900 python bench.py $ARGS > $O/bench_old.json 2> $O/bench_old.err; echo "old rc=$?"; }
python - <<PY
import json
for tag in ("new", "old"):
    try: d = json.loads(open("$O/bench_%s.json" % tag).read().strip().split("\n")[-1])
    except Exception as e: print(tag, "no line", e); continue
    st = lambda r: {k: round(v, 2) for k, v in r["roofline"]["stage_ms_per_step_exclusive"].items()}
    print(tag, "headline %.1f M reads/s" % (d["value"] / 1e6), st(d), "| hard %.1f" % (d["hard"]["value"] / 1e6), st(d["hard"]), "| hard_greedy %.1f" % (d["hard_greedy"]["value"] / 1e6), st(d["hard_greedy"]))
PY
