#!/bin/bash
O=$1
NCCL_DEBUG=INFO timeout 120 python tests/tools/rccl_probe.py > $O/probe_plain.log 2>&1; echo "== plain"; grep -v "^$" $O/probe_plain.log | tail -25
NCCL_DEBUG=INFO timeout 120 python tests/tools/rccl_probe.py torch > $O/probe_torch.log 2>&1; echo "== torch"; grep -v "^$" $O/probe_torch.log | tail -25
