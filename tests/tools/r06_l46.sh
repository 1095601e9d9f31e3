#!/bin/bash
# lease 46: -v against the reference binary on the HOSTILE database (protein families: matches of hundreds of rows in several
# fragments, the 20-accession and 21-id limits, reads in the retry pass; reads with Ns), 400 000 reads, narrow and forced wide
O=$GRAFT_REPO_ROOT/gpurun_out/r06_l46; mkdir -p $O
PREP_HARD=1 python tests/tools/prof_prepare.py /tmp/kjh 200001 400000 > $O/prepare.log 2>&1; tail -n 1 $O/prepare.log
( timeout 900 python tests/tools/cli_verbose_check.py /tmp/kjh 400000 ) > $O/verbose_check_hard.txt 2>&1; echo "[l46] hard rc=$?"; grep "^-a\|OTHER\|gpu:\|ref:" $O/verbose_check_hard.txt | cut -c1-330
( KAIJU_GPU_FORCE_WIDE=20 VB_MODES=mem,greedy timeout 900 python tests/tools/cli_verbose_check.py /tmp/kjh 400000 ) > $O/verbose_check_hard_wide.txt 2>&1; echo "[l46] hard, forced wide rc=$?"; grep "^-a\|OTHER\|gpu:\|ref:" $O/verbose_check_hard_wide.txt | cut -c1-330
