#!/bin/bash
# round 6, lease 7: tests of the compact path without the record memset; locate variants at the wide leg's real size; PMC legs
O=$1
V=kaiju_amd/variants
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py -m gpu -q -x -k "fused or device_resident or lca_compact or cli" ) > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for v in cur ilp2 loc16 cur; do
  KAIJU_GPU_LIB=$PWD/$V/libkaiju_gpu_$v.so timeout 900 python bench.py --reads 2000000 --contexts 1 --steps 1 --warmup 0 --leg-steps 4 --no-cpu-baseline --no-ref-ops --legs wide > $O/wide_${v}_$RANDOM.json 2> $O/wide_${v}_$RANDOM.err
  python - <<P
import json
try:
    d=json.load(open("gpurun_out/bench_detail_n1.json"))
    for nm in ("wide","wide_greedy"):
        r=d[nm]; print("$v", nm, round(r["value"]/1e6,1), "M reads/s", {k:round(x,2) for k,x in r["roofline"]["stage_ms_per_step_exclusive"].items()})
except Exception as e: print("$v failed", e)
P
done
bash tests/tools/pmc_legs.sh $O/pmc_legs hard wide long protein > $O/pmc_legs.log 2>&1; tail -8 $O/pmc_legs.log
cp profiles/traffic.json $O/traffic.json
python tests/tools/pmc_legs_collect.py $O/pmc_legs $O/traffic.json profiles/r06_pmc_legs
find $O/pmc_legs -name "*.csv" -size +20M -delete
find $O/pmc_legs -name "*kernel_trace*" -delete
