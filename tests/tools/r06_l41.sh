#!/bin/bash
# lease 41: Greedy -v - scratch of the first-generation main pass and blocks of the retry pass (250 000 reads per call, as the CLI's batches)
O=$GRAFT_REPO_ROOT/gpurun_out/r06_l41; mkdir -p $O
W=/tmp/kjw
python tests/tools/prof_prepare.py $W 680001 500000 > /dev/null 2>&1
for cfg in "" "KAIJU_GPU_G1_POOL=512" "KAIJU_GPU_G1_POOL=1024" "KAIJU_GPU_G1_MATCH=256" "KAIJU_GPU_G1_POOL=512 KAIJU_GPU_G1_MATCH=256" "KAIJU_GPU_RETRY_BLOCKS=16" "KAIJU_GPU_G1_POOL=512 KAIJU_GPU_G1_MATCH=256 KAIJU_GPU_RETRY_BLOCKS=16"; do
  env $cfg timeout 300 python tests/tools/g1_probe.py $W 250000 2>&1 | tail -n 1
done | tee $O/g1_probe.txt
