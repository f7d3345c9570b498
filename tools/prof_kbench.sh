#!/bin/bash
# Run ON THE GPU BOX (via gpurun): per-kernel time of tools/kbench.py for one shape.
#   bash tools/prof_kbench.sh <tag> <kbench args...>   -> gpurun_out/kprof_<tag>/
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/kprof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 $ROOT/tools/kbench.py "$@" > "$OUT/stats.log" 2>&1
f=$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv" && head -12 "$f"
