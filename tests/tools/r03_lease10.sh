#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_l10; mkdir -p $O
export TMPDIR=/tmp
python tests/tools/prof_prepare.py /tmp/kjw 680001 4000000 > $O/prepare.log 2>&1
run() { name=$1; lib=$2; mode=$3; shift 3
  env "$@" KAIJU_GPU_LIB=$PWD/$lib python tests/tools/prof_run.py /tmp/kjw $mode 1 3 4000000 > $O/$name.txt 2>&1
  echo "== $name"; grep -E "search|checksum" $O/$name.txt | tail -2; }
run mem_loop kaiju_amd/variants/libkaiju_gpu_loop.so mem X=1
run mem_singlepass kaiju_amd/libkaiju_gpu.so mem X=1
run greedy_new kaiju_amd/libkaiju_gpu.so greedy KAIJU_GPU_OVF_STATS=1
SWEEP="3:32,3:0,1:32,1:0,7:32,3:16,1:16,0:0" python tests/tools/greedy_sweep.py /tmp/kjw 4000000 > $O/gate_sweep.txt 2>&1; cat $O/gate_sweep.txt
KAIJU_GPU_LIB=$PWD/kaiju_amd/variants/libkaiju_gpu_prof.so python tests/tools/prof_run.py /tmp/kjw greedy 1 2 4000000 > $O/greedy_prof.txt 2>&1
grep "kj prof" $O/greedy_prof.txt | tail -27
KAIJU_GPU_OVF_STATS=1 timeout 1200 python bench.py --nseq 3600001 --reads 3000000 --steps 2 --warmup 1 --mode greedy --legs "" --no-cpu-baseline > $O/bench_1g_greedy.json 2> $O/bench_1g_greedy.err
grep -E "retry pass|leg|database" $O/bench_1g_greedy.err | sort | uniq -c | tail -8
