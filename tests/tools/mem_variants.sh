#!/bin/bash
# A/B of compile-time variants of the search lanes.  One library per variant under kaiju_amd/variants/ (git-ignored,
# travels to the GPU box with the snapshot):
#   mem_variants.sh build                      here (hipcc cross-compiles; ~35 s per variant)
#   mem_variants.sh build_rev <rev> [tag]      here: the library of another commit (git archive of its sources into a temporary
#                                              directory) as variants/libkaiju_gpu_<tag>.so - the "before" of an A/B (default tag: head)
#   mem_variants.sh run <outdir> [mode] [n]    on the GPU box: the prepared workload with each library, search times and a
#                                              checksum of the records (all variants must agree)
# VARIANTS="cur prof" (default: the current source, the section profiler); add entries to DEF for new experiments.  Round 3
# measured gate1 / gate3 / roll / locp / gdefer / g_occ3 this way (profiles/r03_variants/): the winners are the default code now
R=$(cd "$(dirname "$0")/../.." && pwd)
V=$R/kaiju_amd/variants
SRC="$R/kaiju_amd/csrc/capi.hip $R/kaiju_amd/csrc/fmi_stream.hip $R/kaiju_amd/csrc/exact_pass.hip $R/kaiju_amd/csrc/host_index.cpp $R/kaiju_amd/csrc/host_tables.cpp $R/kaiju_amd/csrc/taxonomy.cpp $R/kaiju_amd/csrc/mkfmi.cpp $R/kaiju_amd/csrc/rccl_gather.cpp"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-gpu-rdc -Wno-unused-result -w"
declare -A DEF=( [cur]="" [prof]="-DKJ_PROF" [stats]="-DKJ_STATS" [ovf]="-DKJ_OVF_STATS" [norule]="-DKJ_NO_SPAN_RULE -DKJ_NO_PROBE" [noprobe]="-DKJ_NO_PROBE" [nospaneq]="-DKJ_NO_SPAN_EQ" [w5]="-DKJ_MEM_WAVES=5" [w6]="-DKJ_MEM_WAVES=6"
  [noprune]="-DKJ_NO_CHAIN_PRUNE" [prunehint]="-DKJ_CHAIN_PRUNE_HINT_ONLY" [g3]="-DKJ_GREEDY3" [g3prof]="-DKJ_GREEDY3 -DKJ_PROF" [loc16]="-DKJ_LOC_TEAM=16" [loc32]="-DKJ_LOC_TEAM=32" [ilp2]="-DKJ_LOC_ILP=2" [ilp2_16]="-DKJ_LOC_ILP=2 -DKJ_LOC_TEAM=16" [ilp2_4]="-DKJ_LOC_ILP=2 -DKJ_LOC_TEAM=4" [segw4]="-DKJ_SEG_WAVES=4" [nowidecp]="-DKJ_NO_WIDE_CHAIN_PRUNE" [postw6]="-DKJ_POST_WAVES=6" [postw5]="-DKJ_POST_WAVES=5" [postnotrig]="-DKJ_POST_NOTRIG" [postnoloc]="-DKJ_POST_NOLOCATE" [postnone]="-DKJ_POST_NOTRIG -DKJ_POST_NOLOCATE" [s1noscan]="-DKJ_S1_NOSCAN" )
LIST=${VARIANTS:-cur prof}
if [ "$1" = build ]; then
  mkdir -p $V
  for v in $LIST; do
    /opt/rocm/bin/hipcc $FLAGS ${DEF[$v]} -o $V/libkaiju_gpu_$v.so $SRC -lpthread -ldl && echo "built $v" || echo "$v: build failed"
  done
  exit 0
fi
if [ "$1" = build_rev ]; then
  REV=${2:?revision}; TAG=${3:-head}
  T=$(mktemp -d) && mkdir -p $V
  ( cd $R && git archive $REV kaiju_amd/csrc include ) | tar -x -C $T || { echo "no such revision: $REV"; exit 1; }
  ( cd $T && /opt/rocm/bin/hipcc $FLAGS -o $V/libkaiju_gpu_$TAG.so $(echo $SRC | sed "s#$R/##g") -lpthread ) && echo "built $TAG from $REV" || echo "$TAG: build failed"
  rm -rf $T
  exit 0
fi
OUT=$2; MODE=${3:-mem}; N=${4:-4000000}
mkdir -p $OUT
[ -f /tmp/kjw/reads.npy ] || python $R/tests/tools/prof_prepare.py /tmp/kjw 680001 $N > /dev/null 2>&1
for v in $LIST; do
  lib=$V/libkaiju_gpu_$v.so
  [ -f $lib ] || { echo "$v: not built (mem_variants.sh build)"; continue; }
  # (the section profiler's build has no counting instantiation worth running: its counting pass is skipped)
  case $v in *prof*) CNT= ;; *) CNT=$PROF_RUN_COUNTS ;; esac
  PROF_RUN_COUNTS=$CNT KAIJU_GPU_LIB=$lib python $R/tests/tools/prof_run.py /tmp/kjw $MODE 1 3 $N > $OUT/$v.txt 2>&1
  echo "== $v"; grep -E "search|checksum" $OUT/$v.txt | tail -3
done
