#!/bin/bash
# A/B of the libraries under kaiju_amd/variants/ on the 4 M-read profiling workload (tests/tools/mem_variants.sh run): Greedy and MEM.
# usage (lease.sh): sh:tests/tools/ab_variants.sh   - VARIANTS / AB_MODES / AB_N from the environment
O=${1:-gpurun_out/ab}
export VARIANTS=${VARIANTS:-"r03 notext cur prof"}
for mode in ${AB_MODES:-greedy}; do
  PROF_RUN_COUNTS=1 KAIJU_GPU_PROF=1 bash tests/tools/mem_variants.sh run $O/$mode $mode ${AB_N:-4000000}
done
