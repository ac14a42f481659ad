#!/bin/bash
# Attention microbench over several builds of the library in ONE box session.  usage: scripts/ab_attn.sh name1 name2 ...
# (foundationpose_amd/lib/libfp_<name>.so); the library in place is restored at the end.
cd "$(dirname "$0")/.." || exit 1
L=foundationpose_amd/lib
cp $L/libfoundationpose_amd.so $L/libfp_keep.so || exit 1
for r in 1 2; do
  for n in "$@"; do
    cp $L/libfp_$n.so $L/libfoundationpose_amd.so && echo -n "$n: " && REPS=200 python scripts/bench_attn.py || exit 1
  done
done
cp $L/libfp_keep.so $L/libfoundationpose_amd.so
