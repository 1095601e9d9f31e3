#!/bin/bash
# round 5, last lease: the GPU suite and smoke() exactly as the driver runs them, with the final library
O=$1
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/gpu_tests.log 2>&1; echo "[l14] suite rc=$?"; tail -4 $O/gpu_tests.log
timeout 600 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo "[l14] smoke rc=$?"; tail -3 $O/smoke.log
