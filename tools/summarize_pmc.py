#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output of tools/profile.sh into per-kernel averages.
    python tools/summarize_pmc.py gpurun_out/prof_<tag>  > profiles/<name>.md
"""
import csv
import re
import glob
import os
import sys
from collections import defaultdict


def short(name):
    """kernel name with its template arguments ("bwd_scan_split_w8<0, true, false, false, false, false>"); None for
    kernels that are not ours (torch fills, copies)"""
    m = re.search(r"(?:fastgrnn::)(?:\(anonymous namespace\)::)?(\w+(?:<[^>]*>)?)\(", name) or \
        re.search(r"(?:fastgrnn::)(?:\(anonymous namespace\)::)?(\w+(?:<[^>]*>)?)", name)
    return m.group(1) if m else None


def main(root):
    print("# rocprofv3 summary of `bench.py --steps 10 --warmup 3` (%s)\n" % os.path.basename(root.rstrip("/")))
    for f in glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True):
        print("## kernel-trace --stats\n\n| kernel | calls | avg us | total % |\n|---|---|---|---|")
        for r in csv.DictReader(open(f)):
            k = short(r["Name"])
            if k:
                print("| %s | %s | %.1f | %s |" % (k, r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
        print()
    for pas in ("fetch", "write", "sq"):
        for f in glob.glob(os.path.join(root, pas, "**", "*counter_collection.csv"), recursive=True):
            acc = defaultdict(lambda: defaultdict(list))
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if k:
                    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            print("## --pmc pass `%s` (per-dispatch average)\n\n| kernel | counter | avg | n |\n|---|---|---|---|" % pas)
            for k in sorted(acc):
                for c in sorted(acc[k]):
                    v = acc[k][c]
                    print("| %s | %s | %.6g | %d |" % (k, c, sum(v) / len(v), len(v)))
            print()


if __name__ == "__main__":
    main(sys.argv[1])
