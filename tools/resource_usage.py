#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS usage of one HIP source, from hipcc's own remarks
(-Rpass-analysis=kernel-resource-usage; cross-compiles, no GPU needed).

    python tools/resource_usage.py kws_amd/csrc/kernels_split.hip [substring]

Prints one line per kernel instantiation: VGPR, AGPR, scratch bytes (non-zero = spills: not allowed in a kernel
written to the operand rule), LDS bytes, waves per SIMD.
"""
import re
import subprocess
import sys

HIPCC = "/opt/rocm/bin/hipcc"


def demangle(names):
    for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
        try:
            out = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True, check=True).stdout
            return out.splitlines()
        except (FileNotFoundError, subprocess.CalledProcessError):
            continue
    return names


def usage(src):
    r = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-c", src,
                        "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    rows, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        if cur is None:
            continue
        for key, pat in (("vgpr", r"\bVGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                         ("sgpr", r"\bSGPRs: (\d+)")):
            m = re.search(pat, line)
            if m:
                cur[key] = int(m.group(1))
    names = demangle([x["name"] for x in rows])
    for x, n in zip(rows, names):
        n = n.replace("fastgrnn::(anonymous namespace)::", "")
        x["pretty"] = re.sub(r"\(.*", "", n)
    return rows


if __name__ == "__main__":
    rows = usage(sys.argv[1])
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    for x in rows:
        if sub in x["pretty"]:
            print("%-86s V%-4d A%-4d scratch %-5d lds %-6d occ %d" % (
                x["pretty"][:86], x.get("vgpr", -1), x.get("agpr", -1), x.get("scratch", -1), x.get("lds", -1),
                x.get("occ", -1)))
