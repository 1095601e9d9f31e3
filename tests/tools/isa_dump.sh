#!/bin/bash
# Device assembly of kaiju_amd/csrc/capi.hip and exact_pass.hip (gfx950, the flags of kaiju_amd/build.py), one file per kernel, into $1.
# Compare two source states with `diff -rq`: a change that must not touch a hot kernel leaves its file identical.
set -e
out=${1:?output directory}
mkdir -p "$out"
rm -f "$out"/*.s
for src in capi exact_pass; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-gpu-rdc -Wno-unused-result \
    --cuda-device-only -S -o "$out/all_$src.asm" kaiju_amd/csrc/$src.hip
  awk -v out="$out" '
    /^[ \t]*\.type[ \t]+[A-Za-z0-9_]+,@function/ { name=$2; sub(/,.*/,"",name); f=out "/" name ".s"; next }
    /^[ \t]*\.size[ \t]/ { f="" ; next }
    f && !/^[ \t]*;/ && !/^\.L(func|tmp)/ { gsub(/BB[0-9]+_/, "BB_"); sub(/[ \t]*;.*$/, ""); print > f }' "$out/all_$src.asm"
done
# registers and scratch per kernel
cat "$out"/all_*.asm | awk '/^[ \t]*\.amdhsa_kernel /{k=$2} /amdhsa_next_free_vgpr|amdhsa_private_segment_fixed_size/{print k, $1, $2}' > "$out/resources.txt"
ls "$out"/*.s | wc -l
