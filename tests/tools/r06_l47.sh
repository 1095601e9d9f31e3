#!/bin/bash
# closing lease of the final library, part two: -v on the hostile database after the first-generation MEM lane writes the peptides
# of the fragments behind ids_from_SI's limit as well (both generations now agree with the reference there)
O=$GRAFT_REPO_ROOT/gpurun_out/r06_close8; mkdir -p $O
PREP_HARD=1 python tests/tools/prof_prepare.py /tmp/kjh 200001 400000 > $O/prepare_hard.log 2>&1
( timeout 900 python tests/tools/cli_verbose_check.py /tmp/kjh 400000 ) > $O/verbose_check_hard.txt 2>&1; echo "[close8] hard rc=$?"; grep "^-a\|OTHER\|gpu:\|ref:" $O/verbose_check_hard.txt | cut -c1-330
( KAIJU_GPU_FORCE_WIDE=20 timeout 900 python tests/tools/cli_verbose_check.py /tmp/kjh 400000 ) > $O/verbose_check_hard_wide.txt 2>&1; echo "[close8] hard, forced wide rc=$?"; grep "^-a\|OTHER\|gpu:\|ref:" $O/verbose_check_hard_wide.txt | cut -c1-330
