#!/bin/bash
# Text verification on an index with 64-bit rows (DevIndex::sa_tpos5): round 3's really sorted 4.35 G-row index (15.5 M proteins, no
# replication) with the text position of every row (what the loader picks there), of every second row (what it picks at
# refseq_ref's 28 G rows) and without the text arrays; first the GPU tests of the feature.   usage (lease.sh): sh:tests/tools/wide_text_ab.sh
O=${1:-gpurun_out/wide_text}; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "text_positions_of_an_index_with_64_bit_rows or wide_index_path" ) > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
W=/dev/shm/kjw; mkdir -p $W
ARGS="--work $W --nseq 15500001 --image --reads 3000000 --steps 3 --warmup 1 --leg-steps 2 --no-ref-ops"
( time KAIJU_GPU_LOAD_TIMES=1 timeout 1500 python bench.py $ARGS --legs greedy --cpu-sample 200000 --cpu-sample-legs 100000 ) > $O/bench_tv0.json 2> $O/bench_tv0.err; echo "tv0 rc=$?"
grep "leg \|built\|index in HBM\|text \|HBM:" $O/bench_tv0.err
( time KAIJU_GPU_LOAD_TIMES=1 KAIJU_GPU_TV_SHIFT=1 timeout 900 python bench.py $ARGS --legs none --no-cpu-baseline ) > $O/bench_tv1.json 2> $O/bench_tv1.err; echo "tv1 rc=$?"; grep "text \|HBM:" $O/bench_tv1.err
( time KAIJU_GPU_LOAD_TIMES=1 KAIJU_GPU_NO_TEXT=1 timeout 900 python bench.py $ARGS --legs none --no-cpu-baseline ) > $O/bench_notext.json 2> $O/bench_notext.err; echo "notext rc=$?"
python - <<PY
import json
for tag in ("tv0", "tv1", "notext"):
    try:
        d = json.loads(open("$O/bench_%s.json" % tag).read().strip().split("\n")[-1])
    except Exception as e:
        print(tag, "no line", e); continue
    r = d["roofline"]
    print(tag, "%.1f M reads/s" % (d["value"] / 1e6), "step %.2f ms" % d["ms_per_step"], "stages", {k: round(v, 2) for k, v in r["stage_ms_per_step_exclusive"].items()},
          "kernel %.3f ms" % r.get("avg_launch_ms", 0), "frac %.3f" % r["frac"],
          ("greedy %.2f M reads/s" % (d["greedy"]["value"] / 1e6)) if "greedy" in d else "", "parity", d.get("mismatches"), d.get("parity_checked_reads"),
          "index bytes", d["config"].get("index_hbm_bytes"))
PY
rm -rf $W
