#!/bin/bash
# Run ON THE GPU BOX: tools/kbench.py with several libraries, interleaved, R rounds.  usage: kb_ab.sh R "kbench args" lib...
R=$1; ARGS=$2; shift 2
for k in $(seq 1 $R); do
  for lib in "$@"; do
    if [ "$lib" = "base" ]; then unset FASTGRNN_HIP_LIB; else export FASTGRNN_HIP_LIB=$PWD/$lib; fi
    echo "$lib #$k: $(python tools/kbench.py $ARGS 2>&1 | tr '\n' ' ')"
  done
done
