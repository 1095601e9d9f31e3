#!/bin/bash
# round 6, lease 4: the fused post-search pass of MEM (k_mem_post1 / _post2) - tests, then A/B of the headline leg
O=$1
( timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fused or device_resident or lca_compact or golden_single or golden_paired or fullsize" ) > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log
for v in "A=1" "KAIJU_GPU_FUSED_POST=0 KAIJU_BENCH_TWO_CALLS=1" "A=2" "KAIJU_GPU_FUSED_POST=0 KAIJU_BENCH_TWO_CALLS=1"; do
  tag=$(echo "$v" | tr -c 'A-Za-z0-9\n' '_' | cut -c1-24)_$RANDOM
  echo "== $v"
  env $v timeout 600 python bench.py --legs "paired,hard" --steps 6 --warmup 2 --no-cpu-baseline --no-ref-ops > $O/bench_$tag.json 2> $O/bench_$tag.err
  echo "rc=$?"; grep "leg \|mismatch" $O/bench_$tag.err | cut -c1-200
  python - <<P
import json
d=json.load(open("$O/bench_$tag.json")); print(d["value"]/1e6, d["ms_per_step"], d["stage_ms"], d["parity"], {k:v["value"]/1e6 for k,v in d["legs"].items()})
P
done
