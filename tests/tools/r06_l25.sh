#!/bin/bash
# lease 25: the SEG pass with teams of lanes per fragment (k_seg_teams<T>) against one wavefront per fragment (k_seg)
O=$GRAFT_REPO_ROOT/gpurun_out/r06_l25; mkdir -p $O
run() {   # run <tag> [env...]
  local tag=$1; shift
  env "$@" timeout 900 python bench.py --mode greedy --legs "" --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_$tag.json 2> $O/bench_$tag.err
  python - $O/bench_$tag.err $tag <<'PY'
import json,re,sys
t=open(sys.argv[1]).read()
m=re.search(r'\[bench\] detail: (\{.*\})',t)
if not m: print(sys.argv[2],'no detail line'); sys.exit(0)
d=json.loads(m.group(1))
print(sys.argv[2], round(d['value']/1e6,2),'M reads/s', {k:round(v,2) for k,v in d['stage_ms_per_step_exclusive'].items()}, 'parity', d.get('parity_checked_reads'), d.get('mismatches'))
PY
}
run t64 KAIJU_GPU_SEG_TEAM=64
run t32 KAIJU_GPU_SEG_TEAM=32
run t16 KAIJU_GPU_SEG_TEAM=16
run t8 KAIJU_GPU_SEG_TEAM=8
run t16w4 KAIJU_GPU_SEG_TEAM=16 KAIJU_GPU_LIB=$GRAFT_REPO_ROOT/kaiju_amd/variants/libkaiju_gpu_segw4.so
run t8w4 KAIJU_GPU_SEG_TEAM=8 KAIJU_GPU_LIB=$GRAFT_REPO_ROOT/kaiju_amd/variants/libkaiju_gpu_segw4.so
run s1noscan KAIJU_GPU_LIB=$GRAFT_REPO_ROOT/kaiju_amd/variants/libkaiju_gpu_s1noscan.so
run s1noappend KAIJU_GPU_LIB=$GRAFT_REPO_ROOT/kaiju_amd/variants/libkaiju_gpu_s1noappend.so
for t in 8 32; do
  ( KAIJU_GPU_SEG_TEAM=$t timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "seg or golden or greedy" ) > $O/gpu_tests_team$t.log 2>&1; echo "[l25] team $t tests rc=$?"; tail -2 $O/gpu_tests_team$t.log
done
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/gpu_tests.log 2>&1; echo "[l25] suite rc=$?"; tail -3 $O/gpu_tests.log
