#!/bin/bash
# round 6, lease 21: Greedy stage 1 with the key sums read four letters at a time - A/B (records must agree)
O=$1
VARIANTS="before cur before cur" bash tests/tools/mem_variants.sh run $O/ab greedy 4000000
VARIANTS="before cur" bash tests/tools/mem_variants.sh run $O/abmem mem 4000000
