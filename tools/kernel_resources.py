#!/usr/bin/env python3
"""Registers / scratch / spills of every kernel in a gfx950 assembly file (the .s that hipcc -save-temps leaves), from the
.amdgpu_metadata block.  usage: kernel_resources.py build/nmpc_block_kernel-hip-amdgcn-amd-amdhsa-gfx950.s [substring]"""
import re
import sys


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    recs, cur = [], {}
    for line in open(path):
        m = re.match(r"\s+(?:- )?\.(name|agpr_count|vgpr_count|sgpr_count|private_segment_fixed_size|vgpr_spill_count|group_segment_fixed_size):\s+(\S+)", line)
        if not m:
            continue
        k, v = m.groups()
        if k == "agpr_count" and "agpr_count" in cur:
            recs.append(cur)
            cur = {}
        cur[k] = v
    if cur:
        recs.append(cur)
    print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'scratch':>8} {'spills':>7}  kernel")
    for r in recs:
        name = r.get("name", "?")
        if name.endswith(".kd") or want not in name:
            continue
        m = re.match(r"_ZN4nmpc(\d+)(\w+?)I((?:L[ib]\d+E)+)E", name)
        if m:  # nmpc::kernel<template arguments>
            kn = m.group(2)[: int(m.group(1))]
            dem = kn + "<" + ", ".join(re.findall(r"L[ib](\d+)E", m.group(3))) + ">"
        else:
            dem = name
        print(f"{r.get('vgpr_count','?'):>5} {r.get('agpr_count','?'):>5} {r.get('sgpr_count','?'):>5} {r.get('private_segment_fixed_size','?'):>8} {r.get('vgpr_spill_count','?'):>7}  {dem}")


if __name__ == "__main__":
    main()
