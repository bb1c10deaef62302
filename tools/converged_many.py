#!/usr/bin/env python3
"""Converged solves (15 real-time iterations per launch) of K independent batches: one launch per batch in order against ONE grid
for all of them (alore_nmpc_rti_many), microseconds per batch of 4096."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from alore_legged_manipulator_amd.nmpc import BatchedNmpc
from alore_legged_manipulator_amd.scenarios import make_batch

def main():
    B, N, K = 4096, 20, int(sys.argv[1]) if len(sys.argv) > 1 else 20
    n_sqp = int(sys.argv[2]) if len(sys.argv) > 2 else 15
    dev = torch.device("cuda:0")
    eng = BatchedNmpc(B, N, device=0, slots=K)
    batch = make_batch(B, N)
    eng.load(batch, slot=None)
    keep = {k: eng.ts[k].clone() for k in ("x", "u", "dual")}
    def restore():
        for k, v in keep.items(): eng.ts[k].copy_(v)
    for mode in ("in order", "one grid"):
        ts = []
        for rep in range(4):
            restore(); torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            if mode == "in order":
                for s in range(K): eng.rti(n_sqp, slot=s)
            else:
                eng.rti_range(0, K, n_sqp)
            torch.cuda.synchronize(dev)
            ts.append((time.perf_counter() - t0) / K * 1e6)
        st = eng.fetch(names=("status",), slot=K - 1)["status"]
        print(f"n_sqp={n_sqp} K={K} {mode:9s}: {min(ts[1:]):8.2f} us per batch of {B} ({B / min(ts[1:]) * 1e6:.3g} solves/s), unsolved {int((st != 0).sum())}", flush=True)
        print("   ", eng.launch_info())

if __name__ == "__main__":
    main()
