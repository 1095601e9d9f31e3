#!/bin/bash
# lease 44: host-side marks of the first calls of `kaiju -a greedy -v` (KAIJU_GPU_CALL_TIMES=1)
O=$GRAFT_REPO_ROOT/gpurun_out/r06_l44; mkdir -p $O
W=/tmp/kjw
python tests/tools/prof_prepare.py $W 680001 2000000 > /dev/null 2>&1
python - <<'PY'
import numpy as np
W="/tmp/kjw"; n=2000000
reads=np.load(f"{W}/reads.npy")[:n]; n,L=reads.shape
with open(f"{W}/v_{n}.fq","wb") as f:
    f.write(b"".join(b"@r%d\n" % i + reads[i].tobytes() + b"\n+\n" + b"I"*L + b"\n" for i in range(n)))
PY
( cd /tmp && KAIJU_GPU_CALL_TIMES=1 KAIJU_GPU_STAGE_TIMES=1 timeout 120 $GRAFT_REPO_ROOT/kaiju_amd/bin/kaiju -t $W/nodes.dmp -f $W/db.fmi -i $W/v_2000000.fq -o $W/vg.tsv -a greedy -v 2>&1 | grep "call \|gpu call\|contexts created" | head -n 40 ) > $O/cli_call_times.txt 2>&1
cut -c1-160 $O/cli_call_times.txt
