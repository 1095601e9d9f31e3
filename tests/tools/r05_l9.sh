#!/bin/bash
# round 5, lease 9: the product-side RCCL gather on the device (communicator of one rank), then the GPU suite as the driver runs it
O=$1
( time timeout 900 python -m pytest tests/test_gpu_dist1.py -x -q ) > $O/dist1_tests.log 2>&1; echo "[l9] dist1 rc=$?"; tail -15 $O/dist1_tests.log
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/gpu_tests.log 2>&1; echo "[l9] suite rc=$?"; tail -4 $O/gpu_tests.log
