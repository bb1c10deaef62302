#!/usr/bin/env python3
"""Per-role accounting of a two-phase grid from its trace (ALORE_NMPC_TP_TRACE=<file>: 4 words per workgroup -- 100 MHz counter at
start / inputs landed / end, role | batch << 8 | turns << 40).   usage: tp_trace.py <file.N> [--timeline]"""
import sys

import numpy as np


def main():
    raw = np.fromfile(sys.argv[1], dtype=np.int64)
    hdr, w = raw[:8], raw[8:].reshape(-1, 4)
    nwg, count, grid, count2, tail, lag = (int(v) for v in hdr[1:7])
    print(f"# {nwg} workgroups: {count} batches of {grid} blocks, {count2} in two phases, tail {tail} per batch, lag {lag} units")
    started = w[:, 0] != 0
    role = (w[:, 3] & 0xFF).astype(int)
    turns = (w[:, 3] >> 40).astype(int)
    ran = started & (w[:, 2] != 0)
    t0 = w[started, 0].min()
    tend = w[ran, 2].max()
    print(f"grid: first start -> last end {(tend - t0) / 100.0:.1f} us; workgroups that left at once (no trace): {int((~started).sum())}, started but no end (empty tails): {int((started & ~ran).sum())}")
    names = {0: "first pass", 1: "one pass", 2: "tail"}
    for r in (0, 1, 2):
        m = ran & (role == r)
        if not m.any():
            continue
        life = (w[m, 2] - w[m, 0]) / 100.0
        load = (w[m, 1] - w[m, 0]) / 100.0
        st = (w[m, 0] - t0) / 100.0
        en = (w[m, 2] - t0) / 100.0
        print(f"{names[r]:10s}: {int(m.sum()):5d} workgroups, lifetime mean {life.mean():6.2f} us (p10 {np.percentile(life, 10):.2f}, p50 {np.percentile(life, 50):.2f}, p90 {np.percentile(life, 90):.2f}, max {life.max():.2f}); "
              f"start -> inputs landed mean {load.mean():.2f} us; starts {st.min():.1f} .. {st.max():.1f} us, ends {en.min():.1f} .. {en.max():.1f} us; turns > 1: {int((turns[m] > 1).sum())}")
    # slot occupancy over time: workgroups in flight per 5 us
    if "--timeline" in sys.argv:
        edges = np.arange(0, (tend - t0) / 100.0 + 5, 5.0)
        print("t_us    in flight: first pass / one pass / tail   (mean over the bin)")
        for a, b in zip(edges[:-1], edges[1:]):
            row = []
            for r in (0, 1, 2):
                m = ran & (role == r)
                s = (w[m, 0] - t0) / 100.0
                e = (w[m, 2] - t0) / 100.0
                ov = np.clip(np.minimum(e, b) - np.maximum(s, a), 0, None).sum() / (b - a)
                row.append(ov)
            print(f"{a:6.0f}  {row[0]:7.0f} {row[1]:7.0f} {row[2]:7.0f}   total {sum(row):7.0f}")


if __name__ == "__main__":
    main()
