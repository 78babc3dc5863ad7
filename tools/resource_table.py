"""Compiler resource usage of every kernel of csrc/gsr_hip.hip (VGPRs, AGPRs, scratch, LDS, occupancy) as a markdown table.
No GPU needed: hipcc cross-compiles for gfx950 with -Rpass-analysis=kernel-resource-usage.
usage: python tools/resource_table.py [extra hipcc flags...] > profiles/rNN_kernel_resources.md"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pf3plat_amd import _lib  # noqa: E402


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return [re.sub(r"\(.*", "", s).replace("void ", "") for s in out.strip().splitlines()]


def main():
    flags = [f for f in _lib.HIPCC_FLAGS if f not in ("-shared",)] + sys.argv[1:]
    cmd = [_lib.find_hipcc(), *flags, "-shared", "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null", _lib.SRC]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark:\s+(?:Function Name|Name): (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+(?:\[[^\]]*\])?): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    names = demangle([r["name"] for r in rows])
    print("| kernel | VGPRs | AGPRs | SGPRs | scratch B/lane | LDS B (static) | waves/SIMD |")
    print("|---|---|---|---|---|---|---|")
    for r, n in zip(rows, names):
        print(f"| `{n}` | {r.get('VGPRs', '')} | {r.get('AGPRs', '')} | {r.get('TotalSGPRs', '')} | {r.get('ScratchSize [bytes/lane]', '')} | "
              f"{r.get('LDS Size [bytes/block]', '')} | {r.get('Occupancy [waves/SIMD]', '')} |")


if __name__ == "__main__":
    main()
