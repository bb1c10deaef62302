#!/usr/bin/env python3
"""Batches of B problems of horizon N kept in flight by one alore_nmpc_rti_many call (what bench.py's reference_horizon_n50 leg times):
us per batch in flight and one launch at a time.   usage: horizon_in_flight.py [N] [slots] [lanes (e.g. 0x108, 0 = automatic)] [B] [warm_start_steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alore_legged_manipulator_amd.nmpc import BatchedNmpc  # noqa: E402
from alore_legged_manipulator_amd.scenarios import make_batch  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 40
lanes = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0
B = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
pg = int(sys.argv[5]) if len(sys.argv) > 5 else -1
eng = BatchedNmpc(B, N, slots=slots + 2, lanes_per_problem=lanes, warm_start_steps=pg)
batch = make_batch(B, N)
ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
res = []
for rep in range(4):
    eng.load(batch, slot=None)
    eng.rti_range(0, 2)
    torch.cuda.synchronize()
    ev_a.record()
    eng.rti_range(2, slots)
    ev_b.record()
    torch.cuda.synchronize()
    res.append(ev_a.elapsed_time(ev_b) / slots * 1e3)
info = eng.launch_info()
bad = int((eng.ts["status"][2:] != 0).sum())
eng.set_launch_overlap(1)
eng.load(batch, slot=None)
eng.rti(1, slot=0)
torch.cuda.synchronize()
ev_a.record()
for i in range(2, 12):
    eng.rti(1, slot=i)
ev_b.record()
torch.cuda.synchronize()
one = ev_a.elapsed_time(ev_b) / 10 * 1e3
bytes_per = 4 * (51 * N + 28) * B
print(f"pg {pg}: N = {N}, B = {B}, {slots} batches in flight, lanes {info['lanes_per_problem']:#x} (grid {info['grid']}, LDS {info['lds_bytes_per_block']} B): "
      f"{' '.join(f'{r:.2f}' for r in res)} us per batch = {bytes_per / (min(res) * 1e-6) / 8e12:.3f} of HBM peak; unsolved {bad}; "
      f"one launch at a time {one:.2f} us (lanes {eng.launch_info()['lanes_per_problem']:#x}); n_iter mean {float(eng.ts['n_iter'].float().mean()):.3f}")
