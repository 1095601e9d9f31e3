#!/bin/bash
# lease 31: contexts in flight x chunk size for the MEM legs (round 2 / 3 measured + 1.6 % and nothing: the step has more kernels
# next to k_mem now); k_mem_post1 compiled for more wavefronts (variant postw6)
O=$GRAFT_REPO_ROOT/gpurun_out/r06_l31; mkdir -p $O
show() { python - $1 "$2" <<'PY'
import json,re,sys
t=open(sys.argv[1]).read()
m=re.search(r'\[bench\] detail: (\{.*\})',t)
if not m: print(sys.argv[2],'no detail line'); sys.exit(0)
d=json.loads(m.group(1))
print(sys.argv[2].ljust(28), round(d['value']/1e6,2), 'M/s  ms/step', round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['roofline']['stage_ms_per_step_exclusive'].items()}, 'ctx', d['config'].get('contexts_in_flight'), 'chunk', d['config'].get('chunk'))
PY
}
run() { local tag=$1; shift; timeout 600 python bench.py --legs "" --steps 6 --no-cpu-baseline "$@" > $O/b_$tag.json 2> $O/b_$tag.err; show $O/b_$tag.err "$tag"; }
run mem_c1_10M
run mem_c2_5M --contexts 2 --chunk 5000000
run mem_c2_3.4M --contexts 2 --chunk 3400000
run mem_c2_2.5M --contexts 2 --chunk 2500000
run mem_c3_3.4M --contexts 3 --chunk 3400000
run mem_c4_2.5M --contexts 4 --chunk 2500000
run mem_c2_5M_again --contexts 2 --chunk 5000000
run pairs_c1_5M --paired --reads 5000000
run pairs_c2_2.5M --paired --reads 5000000 --contexts 2 --chunk 2500000
run mem2M_c1 --reads 2000000
run mem2M_c2_1M --reads 2000000 --contexts 2 --chunk 1000000
KAIJU_GPU_LIB=$GRAFT_REPO_ROOT/kaiju_amd/variants/libkaiju_gpu_postw6.so run mem_c1_postw6
KAIJU_GPU_LIB=$GRAFT_REPO_ROOT/kaiju_amd/variants/libkaiju_gpu_postw6.so run mem_c2_5M_postw6 --contexts 2 --chunk 5000000
