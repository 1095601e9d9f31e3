#!/bin/bash
# round 5, lease 2: where the hostile leg's time goes - kernel summary of its post-search bucket, section profile and op counts
# of k_mem on it, the same leg with 10 M reads per step (how much of its cost is the tail of a 2 M-read launch)
O=$1
bash tests/tools/hard_stats.sh $O/hard_stats > $O/hard_stats.txt 2>&1; tail -26 $O/hard_stats.txt
python tests/tools/hard_prepare.py /tmp/kjh 200001 2000000 > $O/hard_prepare.log 2>&1
for v in cur prof; do
  PROF_RUN_COUNTS=$([ $v = cur ] && echo 1) KAIJU_GPU_LIB=kaiju_amd/variants/libkaiju_gpu_$v.so python tests/tools/prof_run.py /tmp/kjh mem 1 3 > $O/hard_mem_$v.txt 2>&1
  echo "== mem $v"; grep -E "search|checksum|ops per read|kj prof" $O/hard_mem_$v.txt | tail -40
done
timeout 600 python bench.py --legs hard --hard-reads 10000000 --no-cpu-baseline --steps 1 --warmup 1 > $O/bench_hard10m.json 2> $O/bench_hard10m.err; grep "leg " $O/bench_hard10m.err
