#!/bin/bash
# Second A/B of the text verification of the wide MEM lane on the sorted 4.35 G-row index, all on ONE box: the library of the commit
# before it (variants/head) against the current one (variants/cur) without text arrays and with the text position of every row,
# index loaded from the image; and variants/head once more with the index loaded from the .fmi.   usage (lease.sh): sh:tests/tools/wide_text_ab2.sh
O=${1:-gpurun_out/wide_text2}; mkdir -p $O
R=$(cd "$(dirname "$0")/../.." && pwd); V=$R/kaiju_amd/variants
W=/dev/shm/kjw; mkdir -p $W
ARGS="--work $W --nseq 15500001 --reads 3000000 --steps 3 --warmup 1 --legs none --no-cpu-baseline --no-ref-ops"
run() { tag=$1; extra=$2; shift 2; ( time env KAIJU_GPU_LOAD_TIMES=1 "$@" timeout 900 python bench.py $ARGS $extra ) > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "$tag rc=$?"; }
run head_fmi "" KAIJU_GPU_LIB=$V/libkaiju_gpu_head.so
grep "built" $O/bench_head_fmi.err
run head_image --image KAIJU_GPU_LIB=$V/libkaiju_gpu_head.so
run cur_notext_image --image KAIJU_GPU_LIB=$V/libkaiju_gpu_cur.so KAIJU_GPU_NO_TEXT=1
run cur_tv0_image --image KAIJU_GPU_LIB=$V/libkaiju_gpu_cur.so
python - <<PY
import json
for tag in ("head_fmi", "head_image", "cur_notext_image", "cur_tv0_image"):
    try:
        d = json.loads(open("$O/bench_%s.json" % tag).read().strip().split("\n")[-1])
    except Exception as e:
        print(tag, "no line", e); continue
    r = d["roofline"]
    print(tag, "%.1f M reads/s" % (d["value"] / 1e6), "step %.2f ms" % d["ms_per_step"], "stages", {k: round(v, 2) for k, v in r["stage_ms_per_step_exclusive"].items()},
          "kernel %.3f ms" % r.get("avg_launch_ms", 0), "HBM %.2f GB" % (d["config"]["index_hbm_bytes"]["total"] / 1e9))
PY
rm -rf $W
