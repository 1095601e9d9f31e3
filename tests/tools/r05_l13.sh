#!/bin/bash
O=$1
( NCCL_DEBUG=WARN timeout 900 python -m pytest tests/test_gpu_dist1.py -x -q ) > $O/dist1_tests.log 2>&1; echo "[l13] dist1 rc=$?"; grep -n "passed\|failed\|Cuda failure\|KaijuGpuError:\|Error" $O/dist1_tests.log | tail -8
