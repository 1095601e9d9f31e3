#!/bin/bash
O=$1
( NCCL_DEBUG=INFO timeout 600 python -m pytest tests/test_gpu_dist1.py -x -q -k "librarys" ) > $O/dist1_debug.log 2>&1; echo "[l11] rc=$?"; grep -n "passed\|failed\|WARN.*hip\|WARN.*cuda\|WARN.*Cuda\|error" $O/dist1_debug.log | grep -v "iommu\|Could not read node" | tail -20
( timeout 600 python -m pytest tests/test_gpu_dist1.py -x -q ) > $O/dist1_tests.log 2>&1; echo "[l11] all dist1 rc=$?"; tail -3 $O/dist1_tests.log
