#!/bin/bash
# The wide lanes on an index that needs them, this library against another one (KAIJU_AB_LIB, default kaiju_amd/variants/libkaiju_gpu_r03.so):
# a 2.3 M-protein database with every protein seven times (4.5 G rows, every interval seven rows and more - what k_mem_locate_wide
# pays for), pairs in MEM mode + a Greedy leg, parity against the reference binary.   usage (lease.sh): sh:tests/tools/wide_ab.sh
O=${1:-gpurun_out/wide_ab}; mkdir -p $O
W=/dev/shm/kaiju_wide_ab; mkdir -p $W
ARGS="--work $W --nseq 2300001 --copies 7 --paired --reads 3000000 --steps 4 --warmup 1 --legs greedy --leg-steps 2 --no-ref-ops --cpu-sample 100000 --cpu-sample-legs 50000"
( time KAIJU_GPU_LOAD_TIMES=1 timeout 1500 python bench.py $ARGS ) > $O/bench_new.json 2> $O/bench_new.err; echo "new rc=$?"; grep "leg \|built\|index in HBM" $O/bench_new.err
OLD=${KAIJU_AB_LIB:-kaiju_amd/variants/libkaiju_gpu_r03.so}
if [ -f $OLD ]; then
  ( time KAIJU_GPU_LIB=$OLD timeout 900 python bench.py $ARGS --no-cpu-baseline ) > $O/bench_old.json 2> $O/bench_old.err; echo "old rc=$?"; grep "leg " $O/bench_old.err
fi
python - <<PY
import json
for tag in ("new", "old"):
    try:
        d = json.loads(open("$O/bench_%s.json" % tag).read().strip().split("\n")[-1])
    except Exception as e:
        print(tag, "no line", e); continue
    r = d["roofline"]
    print(tag, "%.1f M pairs/s" % (d["value"] / 1e6), "step %.2f ms" % d["ms_per_step"], "stages", {k: round(v, 2) for k, v in r["stage_ms_per_step_exclusive"].items()},
          "greedy %.2f M reads/s" % (d["greedy"]["value"] / 1e6), "parity", d.get("mismatches"), d.get("parity_checked_reads"))
PY
rm -rf $W
