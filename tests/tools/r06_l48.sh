#!/bin/bash
# lease 48: the headline with two (default) and three contexts per step, alternating, final library
O=$GRAFT_REPO_ROOT/gpurun_out/r06_l48; mkdir -p $O
for k in 1 2; do for c in 2 3; do
  timeout 200 python bench.py --no-cpu-baseline --legs "" --steps 6 --contexts $c > $O/b_c${c}_$k.json 2> $O/b_c${c}_$k.err
  python - $O/b_c${c}_$k.json $c <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("contexts", sys.argv[2], round(d["value"]/1e6,1), "M reads/s,", d["ms_per_step"], "ms per step", d.get("stage_ms"))
PY
done; done | tee $O/summary.txt
