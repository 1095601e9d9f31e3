#!/bin/bash
# round 5, lease 3: the MEM lanes without their in-lane locate (every read's matches go to k_mem_locate*): the GPU suite, A/B of
# the narrow lane at four and five wavefronts per SIMD on the i.i.d. and on the hostile workload, the default bench line
O=$1
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/gpu_tests.log 2>&1; echo "[l3] suite rc=$?"; tail -4 $O/gpu_tests.log
python tests/tools/hard_prepare.py /tmp/kjh 200001 2000000 > $O/hard_prepare.log 2>&1
python tests/tools/prof_prepare.py /tmp/kjw 680001 4000000 > $O/iid_prepare.log 2>&1
for rep in 1 2; do
  for v in cur w5; do
    for w in kjw kjh; do
      PROF_RUN_COUNTS=$([ $rep = 1 ] && echo 1) KAIJU_GPU_LIB=kaiju_amd/variants/libkaiju_gpu_$v.so python tests/tools/prof_run.py /tmp/$w mem 1 3 > $O/${w}_${v}_$rep.txt 2>&1
      echo "== $w $v $rep"; grep -E "search|checksum|ops per read" $O/${w}_${v}_$rep.txt | tail -3
    done
  done
done
KAIJU_GPU_LIB=kaiju_amd/variants/libkaiju_gpu_prof.so python tests/tools/prof_run.py /tmp/kjh mem 1 2 > $O/kjh_prof.txt 2>&1; grep "kj prof" $O/kjh_prof.txt | tail -18
( time timeout 1500 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err ); echo "[l3] bench rc=$?"; grep "leg \|wide index" $O/bench_n1.err
