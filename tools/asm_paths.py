#!/usr/bin/env python3
"""Where a kernel's instructions sit, by the marker comments the kernel body leaves ("; first pass, diagonal weights", "; diagonal weights",
"; general weights", "; @phase ..."): instruction classes per layout region between two markers.
usage: asm_paths.py <file.s> <kernel-name-substring> [marker-regex]"""
import re
import sys

sys.path.insert(0, __import__("os").path.dirname(__file__))
from asm_phases import classify, kernel_body  # noqa: E402


def main():
    path, key = sys.argv[1], sys.argv[2]
    marker = re.compile(sys.argv[3] if len(sys.argv) > 3 else r"^\s*; (first pass, diagonal weights|diagonal weights|general weights)")
    name, body = kernel_body(path, key)
    regions, cur, counts = [], "entry", {}
    for ln in body:
        t = ln.strip()
        m = marker.match(ln)
        if m:
            regions.append((cur, counts))
            cur, counts = m.group(1), {}
            continue
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        for c in classify(t):
            counts[c] = counts.get(c, 0) + 1
    regions.append((cur, counts))
    print("#", name)
    cols = ("valu", "pk", "accvgpr", "mov", "salu", "lds", "vmem", "scratch", "wait")
    print(f"{'region (layout order)':40s}" + "".join(f"{c:>9s}" for c in cols))
    for nm, c in regions:
        print(f"{nm:40s}" + "".join(f"{c.get(k, 0):9d}" for k in cols))


if __name__ == "__main__":
    main()
