#!/bin/bash
# Device assembly of kaiju_amd/csrc/capi.hip (gfx950, the flags of kaiju_amd/build.py), one file per kernel, into $1.
# Compare two source states with `diff -rq`: a change that must not touch a hot kernel leaves its file identical.
set -e
out=${1:?output directory}
mkdir -p "$out"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-gpu-rdc -Wno-unused-result \
  --cuda-device-only -S -o "$out/all.s" kaiju_amd/csrc/capi.hip
awk -v out="$out" '
  /^[ \t]*\.type[ \t]+[A-Za-z0-9_]+,@function/ { name=$2; sub(/,.*/,"",name); f=out "/" name ".s"; next }
  /^[ \t]*\.size[ \t]/ { f="" ; next }
  f && !/^[ \t]*;/ && !/^\.L(func|tmp)/ { gsub(/BB[0-9]+_/, "BB_"); sub(/[ \t]*;.*$/, ""); print > f }' "$out/all.s"
ls "$out" | wc -l
