#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel-trace stats + separate PMC passes for the bench.
#   bash tools/profile.sh <tag>     -> gpurun_out/prof_<tag>/{stats,fetch,write,sq}/...
# rocprofv3 is given `python3 <script>` directly (no env/bash hop), counters are collected
# in their own passes (FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2).
set -u
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 $ARGS > "$OUT/stats.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 $ARGS > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 $ARGS > "$OUT/write.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq" -- python3 $ARGS > "$OUT/sq.log" 2>&1
# the summary is made HERE (the raw counter files of a run with every configuration exceed what gpurun copies back)
python3 $ROOT/tools/summarize_pmc.py "$OUT" > "$OUT/pmc_summary.md" 2> "$OUT/pmc_summary.err"
f=$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv"
find "$OUT" -name "*counter_collection.csv" -delete
find "$OUT" -name "*kernel_trace.csv" -size +4M -delete
du -sh "$OUT"
head -40 "$OUT/pmc_summary.md"
