#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_l7; mkdir -p $O
export TMPDIR=/tmp
python tests/tools/prof_prepare.py /tmp/kjw 680001 4000000 > $O/prepare.log 2>&1
run() { name=$1; lib=$2; mode=$3; shift 3
  env "$@" KAIJU_GPU_LIB=$PWD/$lib python tests/tools/prof_run.py /tmp/kjw $mode 1 3 4000000 > $O/$name.txt 2>&1
  echo "== $name"; grep -E "search|checksum" $O/$name.txt | tail -2; }
run mem_new kaiju_amd/libkaiju_gpu.so mem X=1
run greedy_new kaiju_amd/libkaiju_gpu.so greedy X=1
run mem_prof kaiju_amd/variants/libkaiju_gpu_prof.so mem X=1
run greedy_prof kaiju_amd/variants/libkaiju_gpu_prof.so greedy X=1
grep "kj prof" $O/mem_prof.txt | tail -22; grep "kj prof" $O/greedy_prof.txt | tail -30
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -o s -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --legs "" --steps 5 > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/bench_under_rocprof.err
cd $GRAFT_REPO_ROOT; find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; head -25 $O/kernel_stats.csv | cut -c1-150
python bench.py --no-cpu-baseline --legs "" --steps 5 --contexts 2 --chunk 5000000 > $O/bench_ctx2.json 2> $O/bench_ctx2.err; tail -1 $O/bench_ctx2.err
