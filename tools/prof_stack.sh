#!/bin/bash
# Run ON THE GPU BOX (via gpurun): per-kernel time of tools/stack_bench.py.
#   bash tools/prof_stack.sh <tag> [stack_bench args...]   -> gpurun_out/sprof_<tag>/kernel_stats.csv
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/sprof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 $ROOT/tools/stack_bench.py "$@" > "$OUT/stats.log" 2>&1
f=$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv" && head -24 "$f" | cut -d, -f1-6 | cut -c1-150
tail -2 "$OUT/stats.log"
