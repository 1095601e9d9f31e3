#!/bin/bash
# round 6, second closing lease (the library with the score-bounded chain pruning): GPU suite, randomised hunt, PMC passes of
# headline / greedy / paired and of the hard legs
O=$1
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/gpu_tests.log 2>&1; echo "[final] suite rc=$?"; tail -3 $O/gpu_tests.log
( time timeout 900 python tests/tools/fuzz_gpu.py 80 91 ) > $O/fuzz_gpu_narrow.log 2>&1; echo "[final] fuzz narrow rc=$?"; tail -2 $O/fuzz_gpu_narrow.log
( time KAIJU_GPU_NO_TEXT=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x ) > $O/no_text_tests.log 2>&1; echo "no text rc=$?"; tail -2 $O/no_text_tests.log
bash tests/tools/pmc_legs.sh $O/pmc_legs hard > $O/pmc_legs.log 2>&1; tail -3 $O/pmc_legs.log
cp profiles/traffic.json $O/traffic_legs.json
python tests/tools/pmc_legs_collect.py $O/pmc_legs $O/traffic_legs.json profiles/r06_pmc_legs
find $O/pmc_legs -name "*kernel_trace*" -delete
