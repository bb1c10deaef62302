"""Worker of tests/test_shard_gloo.py: one rank of a world_size-2 gloo job on CPU.

Exercises the multi-GPU data path of bench.py with the CPU stand-ins that exist
on a machine without GPUs: each rank owns a contiguous block of the global batch
(generated locally from the seeded stream), solves it (here with the oracle --
test infrastructure), and the results are collected with the same asynchronous
bucketed all-gather the benchmark uses.  Rank 0 checks the gathered result
against a single-process solve of the whole batch."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from alore_legged_manipulator_amd.scenarios import make_batch, problem  # noqa: E402
from alore_legged_manipulator_amd.shard import ResultGatherer, partition  # noqa: E402
from oracle.drivers import Oracle  # noqa: E402


def solve(batch, N):
    B = batch["x"].shape[0]
    orc = Oracle(N)
    x = np.zeros((B, N + 1, 3), np.float32)
    u = np.zeros((B, N, 2), np.float32)
    st = np.zeros((B,), np.int32)
    for b in range(B):
        orc.reset(); orc.initialize_solver(); orc.load(problem(batch, b))
        orc.preparation_step()
        st[b] = orc.feedback_step()
        x[b] = orc.v["x"].reshape(N + 1, 3)
        u[b] = orc.v["u"].reshape(N, 2)
    return x, u, st


def main():
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    N, total, steps = 20, 22, 3  # 22 problems over 2 ranks: 11 + 11; steps = buckets of different sizes
    off, cnt = partition(total, world, rank)
    assert cnt == 11 and off == rank * 11
    g = ResultGatherer(dist, world)
    xs, us, sts = [], [], []
    for s in range(steps):
        batch = make_batch(cnt, N, seed=1000 + s, offset=off)
        x, u, st = solve(batch, N)
        xs.append(torch.from_numpy(x)); us.append(torch.from_numpy(u)); sts.append(torch.from_numpy(st))
    # bucket 1: steps 0..1 in one collective per tensor, bucket 2: step 2
    g.submit({"x": torch.stack(xs[:2]), "u": torch.stack(us[:2]), "status": torch.stack(sts[:2])})
    first = {k: v.clone() for k, v in g.wait().items()}
    g.submit({"x": torch.stack(xs[2:]), "u": torch.stack(us[2:]), "status": torch.stack(sts[2:])})
    second = g.wait()
    ok = True
    if rank == 0:
        for s in range(steps):
            full = make_batch(total, N, seed=1000 + s)
            x, u, st = solve(full, N)
            src = first if s < 2 else second
            i = s if s < 2 else 0
            gx = torch.cat([src["x"][r][i] for r in range(world)]).numpy()
            gu = torch.cat([src["u"][r][i] for r in range(world)]).numpy()
            gs = torch.cat([src["status"][r][i] for r in range(world)]).numpy()
            ok = ok and np.array_equal(gx, x) and np.array_equal(gu, u) and np.array_equal(gs, st)
        print("GATHER_OK" if ok else "GATHER_MISMATCH", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
