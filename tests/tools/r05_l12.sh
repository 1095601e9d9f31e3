#!/bin/bash
O=$1
( NCCL_DEBUG=WARN timeout 600 python -m pytest tests/test_gpu_dist1.py -x -q ) > $O/dist1_tests.log 2>&1; echo "[l12] dist1 rc=$?"; grep -n "passed\|failed\|Cuda failure\|KaijuGpuError:" $O/dist1_tests.log | tail -8
timeout 120 python tests/tools/rccl_probe.py > $O/probe_plain.log 2>&1; grep "comm ok\|comm failed" $O/probe_plain.log
timeout 120 python tests/tools/rccl_probe.py torch > $O/probe_torch.log 2>&1; grep "comm ok\|comm failed" $O/probe_torch.log
