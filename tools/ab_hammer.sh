#!/bin/bash
# A/B: N processes of tools/repro_seq.py per library; prints the last line of each.  usage: ab_hammer.sh N lib1 [lib2 ...]
N=$1; shift
export REPRO_SHOW=0 REPRO_SHORT=${REPRO_SHORT-1} REPRO_POISON=0x7fc00000,0,0x3f800000,0x7fc00000,0x41200000
for k in $(seq 1 $N); do
  for lib in "$@"; do
    if [ "$lib" = "base" ]; then unset FASTGRNN_HIP_LIB; else export FASTGRNN_HIP_LIB=$PWD/$lib; fi
    echo "$lib #$k: $(timeout -k 10 120 python tools/repro_seq.py 30 2>&1 | tail -1)"
  done
done
