#!/bin/bash
# round 5, lease 1: the streamed .fmi load on the device, the full GPU suite, the default bench line (with the wide leg)
O=$1
( time timeout 600 python -m pytest tests/test_gpu_stream_load.py -x -q ) > $O/stream_tests.log 2>&1; echo "[l1] stream tests rc=$?"; tail -15 $O/stream_tests.log
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/gpu_tests.log 2>&1; echo "[l1] suite rc=$?"; tail -6 $O/gpu_tests.log
( time KAIJU_GPU_LOAD_TIMES=1 timeout 1500 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err ); echo "[l1] bench rc=$?"; grep -v "kaiju_gpu pack\]" $O/bench_n1.err | tail -60
