#!/usr/bin/env python3
"""Register / spill / LDS table of every kernel of a HIP source (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kernel_resources.py gapro_amd/csrc/svgp_fit.hip [-DGAPRO_NT=256 ...]
"""
import re
import subprocess
import sys


def main():
    src, extra = sys.argv[1], sys.argv[2:]
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "-Rpass-analysis=kernel-resource-usage",
           "-Wno-pass-failed", "-o", "/dev/null", src] + extra
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: [^:]*:\d+:\d+: (.*?) \[-Rpass", line) or re.search(r"remark: (.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    for r in rows:
        name = re.sub(r"\(anonymous namespace\)::", "", subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip())
        name = re.sub(r"^void ", "", name).split("(")[0]
        print("%-60s VGPR %4s AGPR %4s spill %4s scratch %5s LDS %6s occ %s" % (
            name[-60:], r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("VGPRs Spill", "?"),
            r.get("ScratchSize [bytes/lane]", "?"), r.get("LDS Size [bytes/block]", "?"), r.get("Occupancy [waves/SIMD]", "?")))


if __name__ == "__main__":
    main()
