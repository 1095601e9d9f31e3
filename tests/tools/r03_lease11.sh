#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_l11; mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py -m gpu -x -q ) > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log
python tests/tools/prof_prepare.py /tmp/kjw 680001 4000000 > $O/prepare.log 2>&1
KAIJU_GPU_LIB=$PWD/kaiju_amd/libkaiju_gpu.so python tests/tools/prof_run.py /tmp/kjw greedy 1 3 4000000 > $O/greedy_new.txt 2>&1; grep -E "search|checksum" $O/greedy_new.txt | tail -2
timeout 1200 python bench.py --nseq 3600001 --reads 3000000 --steps 2 --warmup 1 --mode greedy --legs "" --no-cpu-baseline > $O/bench_1g_greedy.json 2> $O/bench_1g_greedy.err
grep -E "leg|database" $O/bench_1g_greedy.err | tail -3
python - <<'PY'
import json
p=json.load(open('gpurun_out/r03_l11/bench_1g_greedy.json'))
print(p['value']/1e6, p['ms_per_step'], p['config']['overflow_retries_per_step'], p['roofline']['stage_ms_per_step_exclusive'])
PY
