#!/bin/bash
# lease 43: where the wall time of `kaiju -a greedy -v` goes (2 M reads): stage marks of the command line program, first / second call of the library
O=$GRAFT_REPO_ROOT/gpurun_out/r06_l43; mkdir -p $O
W=/tmp/kjw
python tests/tools/prof_prepare.py $W 680001 2000000 > /dev/null 2>&1
python - <<'PY'
import numpy as np
W="/tmp/kjw"; n=2000000
reads=np.load(f"{W}/reads.npy")[:n]; n,L=reads.shape
with open(f"{W}/v_{n}.fq","wb") as f:
    f.write(b"".join(b"@r%d\n" % i + reads[i].tobytes() + b"\n+\n" + b"I"*L + b"\n" for i in range(n)))
PY
for v in "" "-v"; do for a in mem greedy; do
  echo "== -a $a $v"; ( cd /tmp && KAIJU_GPU_STAGE_TIMES=1 timeout 120 $GRAFT_REPO_ROOT/kaiju_amd/bin/kaiju -t $W/nodes.dmp -f $W/db.fmi -i $W/v_2000000.fq -o $W/vg.tsv -a $a $v 2>&1 | grep -v "^1[0-9]:\|Reading\|Parameters\|run mode\|minimum\|input file\|output file" )
done; done > $O/cli_stage_times.txt 2>&1
tail -n 60 $O/cli_stage_times.txt | cut -c1-200
timeout 300 python tests/tools/g1_probe.py $W 250000 2>&1 | tail -n 2 | tee $O/g1_probe.txt
