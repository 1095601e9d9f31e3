#!/bin/bash
# round 5, lease 7: A/B of the narrow MEM lane with its rarely touched state in LDS (90 VGPRs: five wavefronts per SIMD) against
# the lane with that state in registers (107 VGPRs: four), on the i.i.d. and the hostile workload, alternating runs
O=$1
python tests/tools/hard_prepare.py /tmp/kjh 200001 2000000 > $O/hard_prepare.log 2>&1
python tests/tools/prof_prepare.py /tmp/kjw 680001 4000000 > $O/iid_prepare.log 2>&1
for rep in 1 2; do
  for v in cur lds; do
    for w in kjw kjh; do
      KAIJU_GPU_LIB=kaiju_amd/variants/libkaiju_gpu_$v.so python tests/tools/prof_run.py /tmp/$w mem 1 3 > $O/${w}_${v}_$rep.txt 2>&1
      echo "== $w $v $rep"; grep -E "search|checksum" $O/${w}_${v}_$rep.txt | tail -3
    done
  done
done
for v in cur lds; do
  KAIJU_GPU_LIB=kaiju_amd/variants/libkaiju_gpu_$v.so timeout 600 python bench.py --legs paired --no-cpu-baseline --steps 3 > $O/bench_$v.json 2> $O/bench_$v.err; grep "leg " $O/bench_$v.err
done
