#!/usr/bin/env python3
"""Per-SIMD timeline of one grid of nmpc::rti_block_kernel from the record of its instrumented twin.

    ALORE_NMPC_TRACE=/tmp/t.bin python bench.py --no-cpu-baseline --no-extras --no-converged --no-graph --steps 20 --warmup 5
    python tools/trace_timeline.py /tmp/t.bin

The file holds the LAST traced grid (header of 8 words, then 8 words per workgroup: 100 MHz real-time counter at wave start,
after the staggered wait, inputs landed, last store issued, stores acknowledged; HW_ID | XCC_ID << 32; sweeps of the slowest
problem of the wavefront; diagonal-weight path).  Printed: where the grid's duration goes -- spread of the first starts, wave
lifetimes by number of sweeps, the gap between one wavefront's end and the next one's start on the same SIMD, and the tail
(SIMD-time idle after a SIMD's last wavefront) -- all in microseconds.
"""
import sys

import numpy as np


def main():
    path = sys.argv[1]
    raw = np.fromfile(path, dtype=np.int64)
    hdr, w = raw[:8], raw[8:].reshape(-1, 8)
    n, count, per_batch, stag_blocks, stag_x1024, pg = (int(v) for v in hdr[1:7])
    t = w[:, :5].astype(np.float64) * 0.01  # us
    t0 = t[:, 0].min()
    t -= t0
    hw = w[:, 5] & 0xFFFFFFFF
    xcc = (w[:, 5] >> 32) & 0xF
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
    slot = ((((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd).astype(np.int64)
    sweeps = w[:, 6]
    life = t[:, 4] - t[:, 0]
    dur = t[:, 4].max()
    print(f"grid: {n} workgroups = {count} batches x {per_batch}; stagger {stag_blocks} blocks x {stag_x1024 / 1024 * 10:.2f} ns; prediction steps {pg}")
    print(f"duration first start -> last acknowledged store: {dur:8.2f} us   ({dur / count:.3f} us per batch)")
    slots = np.unique(slot)
    print(f"SIMDs used: {slots.size}; XCDs {np.unique(xcc).size}; workgroups per SIMD min/mean/max {np.bincount(np.searchsorted(slots, slot)).min()} / {n / slots.size:.2f} / {np.bincount(np.searchsorted(slots, slot)).max()}")

    def stats(name, v):
        v = np.asarray(v, dtype=np.float64)
        if v.size == 0:
            return
        print(f"  {name:44s} n={v.size:6d} mean {v.mean():7.2f}  p10 {np.percentile(v, 10):7.2f}  p50 {np.percentile(v, 50):7.2f}  p90 {np.percentile(v, 90):7.2f}  max {v.max():7.2f}")

    print("per wavefront (us):")
    stats("lifetime (start -> stores acknowledged)", life)
    stats("  staggered wait", t[:, 1] - t[:, 0])
    stats("  loads (issue -> landed)", t[:, 2] - t[:, 1])
    stats("  compute (landed -> last store issued)", t[:, 3] - t[:, 2])
    stats("  store acknowledgement", t[:, 4] - t[:, 3])
    for s in np.unique(sweeps):
        m = sweeps == s
        stats(f"  compute, slowest problem took {int(s)} sweep(s) [{100.0 * m.mean():.1f} %]", (t[:, 3] - t[:, 2])[m])
    first = t[:, 0] < np.percentile(t[:, 0], 100.0 * min(1.0, slots.size / n))
    stats("loads of the first residency", (t[:, 2] - t[:, 1])[first])
    stats("loads of the later residencies", (t[:, 2] - t[:, 1])[~first])
    # per SIMD: order by start, gaps between acknowledged end and the next start, idle tail
    order = np.lexsort((t[:, 0], slot))
    so, st, en = slot[order], t[order, 0], t[order, 4]
    same = so[1:] == so[:-1]
    gaps = (st[1:] - en[:-1])[same]
    stats("gap: end of a wavefront -> start of the next on its SIMD", gaps)
    last_end = np.array([en[so == s].max() for s in slots])
    first_start = np.array([st[so == s].min() for s in slots])
    busy = np.array([(en[so == s] - st[so == s]).sum() for s in slots])
    print(f"SIMD-time: total {slots.size * dur:9.0f} us = wavefront lifetimes {busy.sum() / (slots.size * dur) * 100:5.1f} % (of which staggered wait "
          f"{(t[:, 1] - t[:, 0]).sum() / (slots.size * dur) * 100:4.1f} %, waiting for loads {(t[:, 2] - t[:, 1]).sum() / (slots.size * dur) * 100:4.1f} %, "
          f"store acks {(t[:, 4] - t[:, 3]).sum() / (slots.size * dur) * 100:4.1f} %)")
    print(f"           before the first wavefront {first_start.sum() / (slots.size * dur) * 100:5.1f} %, between wavefronts {gaps.sum() / (slots.size * dur) * 100:5.1f} %, "
          f"after the last one (tail) {(dur - last_end).sum() / (slots.size * dur) * 100:5.1f} %")
    stats("first start per SIMD", first_start)
    stats("tail per SIMD (grid end - its last wavefront's end)", dur - last_end)
    # the grid is dealt to the XCDs round-robin (workgroup i -> XCD i mod 8): per XCD, when it ran dry
    print("per XCD: workgroups, end of its last wavefront (us), mean lifetime (us), mean compute (us)")
    for xc in np.unique(xcc):
        m = xcc == xc
        print(f"  XCD {int(xc)}: {int(m.sum()):6d}   {t[m, 4].max():8.2f}   {life[m].mean():6.2f}   {(t[m, 3] - t[m, 2]).mean():6.2f}")
    se_key = (xcc * 8 + se)
    ends = np.array([t[se_key == k, 4].max() for k in np.unique(se_key)])
    cnt = np.array([(se_key == k).sum() for k in np.unique(se_key)])
    print(f"per shader engine ({ends.size}): workgroups min/max {cnt.min()} / {cnt.max()}; last end min / mean / max {ends.min():.2f} / {ends.mean():.2f} / {ends.max():.2f} us")
    # resident wavefronts over time
    edges = np.linspace(0.0, dur, 25)
    mid = 0.5 * (edges[1:] + edges[:-1])
    res = [(np.minimum(t[:, 4], b) - np.maximum(t[:, 0], a)).clip(min=0).sum() / (b - a) for a, b in zip(edges[:-1], edges[1:])]
    load = [(np.minimum(t[:, 2], b) - np.maximum(t[:, 1], a)).clip(min=0).sum() / (b - a) for a, b in zip(edges[:-1], edges[1:])]
    print("time (us)      resident wavefronts   of which waiting for their loads")
    for m_, r_, l_ in zip(mid, res, load):
        print(f"  {m_:8.1f}        {r_:8.0f}              {l_:8.0f}")


if __name__ == "__main__":
    main()
