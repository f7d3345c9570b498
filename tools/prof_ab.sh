#!/bin/bash
# Run ON THE GPU BOX: same-run A/B of two libraries under rocprofv3 kernel stats (tools/stack_bench.py).
#   bash tools/prof_ab.sh <tag> <libA|base> <libB> [stack_bench args]
TAG=$1; LA=$2; LB=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for v in A B; do
  lib=$LA; [ $v = B ] && lib=$LB
  if [ "$lib" = "base" ]; then unset FASTGRNN_HIP_LIB; else export FASTGRNN_HIP_LIB=$ROOT/$lib; fi
  OUT=$ROOT/gpurun_out/ab_${TAG}_$v; mkdir -p $OUT
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/tools/stack_bench.py "$@" > $OUT/stats.log 2>&1)
  f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats.csv
  echo "== $v ($lib): $(grep 'ms/step' $OUT/stats.log)"
  python3 - $OUT/kernel_stats.csv <<'PY'
import csv,re,sys
for r in csv.DictReader(open(sys.argv[1])):
    m=re.search(r"(\w+(?:<[^>]*>)?)\(", r["Name"])
    if "fastgrnn" in r["Name"] and float(r["AverageNs"])>20000:
        print("  %-58s %4s %8.1f us"%(m.group(1)[:58], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
