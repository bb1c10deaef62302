"""Worker of tests/test_shard_gloo.py: one rank of a world_size-2 (or -3: uneven partition) gloo job on CPU.

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
from alore_legged_manipulator_amd.shard import ResultGatherer, concat_gathered, pad_rows, partition  # noqa: E402
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


class OracleEngine:
    """Stand-in for BatchedNmpc with the same surface shard.timed_pass uses (load, rti, per-slot result tensors), the
    oracle solving the problems: test infrastructure, so that the multi-rank bookkeeping of bench.py runs on CPU."""

    def __init__(self, B, N, slots):
        self.B, self.N = B, N
        self.ts = {"x": torch.zeros(slots, B, N + 1, 3), "u": torch.zeros(slots, B, N, 2),
                   "status": torch.full((slots, B), -1, dtype=torch.int32), "kkt": torch.zeros(slots, B)}
        self.launches = []
        self.batch = None

    def load(self, batch, slot=None):
        assert slot is None
        self.batch = batch
        for k in ("x", "u"):
            self.ts[k][:] = torch.from_numpy(batch[k]).reshape(self.ts[k].shape[1:])
        self.ts["status"].fill_(-1)

    def rti(self, n, slot=0):
        self.launches.append(slot)
        orc = Oracle(self.N)
        for b in range(self.B):
            pr = problem(self.batch, b)
            pr["x"] = self.ts["x"][slot, b].numpy().reshape(-1).copy(); pr["u"] = self.ts["u"][slot, b].numpy().reshape(-1).copy()
            orc.reset(); orc.initialize_solver(); orc.load(pr)
            for it in range(n):   # n real-time iterations, as alore_nmpc_rti(n_sqp = n)
                orc.preparation_step()
                self.ts["status"][slot, b] = orc.feedback_step()
            self.ts["x"][slot, b] = torch.from_numpy(orc.v["x"].reshape(self.N + 1, 3))
            self.ts["u"][slot, b] = torch.from_numpy(orc.v["u"].reshape(self.N, 2))
            self.ts["kkt"][slot, b] = orc.get_kkt()


def bench_pass_bookkeeping(rank, world):
    """bench.py's timed passes (shard.timed_pass) under gloo: bucket boundaries of the every-batch gather, the
    last-batch-only gather, a failing secondary pass that must not cost the primary figure."""
    from alore_legged_manipulator_amd.shard import HostHooks, timed_pass
    N, B, W, K, ge = 20, 3, 1, 5, 2
    batch = make_batch(B, N, seed=77, offset=rank * B)
    eng = OracleEngine(B, N, W + K)
    g = ResultGatherer(dist, world)
    hooks = HostHooks(dist, world)
    submits = []
    orig = g.submit
    g.submit = lambda t: (submits.append(tuple(t["x"].shape)), orig(t))[1]
    checks = {}
    el, dms, graph = timed_pass(eng, batch, "full", K, W, ge, g, hooks, world)
    checks["times"] = el > 0 and dms > 0 and not graph
    checks["launch order"] = eng.launches == list(range(W + K))           # W warm-up + K timed steps, each slot once
    checks["buckets"] = submits == [(1, B, N + 1, 3), (2, B, N + 1, 3), (2, B, N + 1, 3), (1, B, N + 1, 3)]   # warm-up step, then 2, 2 and the remainder
    last = g.wait()
    checks["gather shape"] = tuple(last["x"].shape) == (world, 1, B, N + 1, 3)
    checks["own slab"] = torch.equal(last["x"][rank][0], eng.ts["x"][W + K - 1]) and bool((last["status"] == 0).all())
    peer = (rank + 1) % world
    other = make_batch(B, N, seed=77, offset=peer * B)                    # what the next rank must have sent
    e2 = OracleEngine(B, N, 1); e2.load(other); e2.rti(1, 0)
    checks["peer slab"] = torch.equal(last["u"][peer][0], e2.ts["u"][0])
    # last-batch-only gather
    submits.clear(); eng.launches.clear()
    el2, dms2, _ = timed_pass(eng, batch, "last", K, W, ge, g, hooks, world)
    checks["last: one submit"] = submits == [(B, N + 1, 3)] and eng.launches == list(range(W + K))
    checks["last: slab"] = torch.equal(g.wait()["x"][rank], eng.ts["x"][W + K - 1])
    # the north star's unit: every step converged (here 2 iterations) and gathered to every rank, the previous gather still
    # readable while the next one lands (two output buffers in turn)
    submits.clear(); eng.launches.clear()
    g2 = ResultGatherer(dist, world, depth=2)
    seen = []
    orig2 = g2.submit
    g2.submit = lambda t: (seen.append({k: v.clone() for k, v in t.items()}), orig2(t))[1]
    el3, dms3, graph3 = timed_pass(eng, batch, "converged", K, W, ge, g2, hooks, world, conv_iters=2)
    lastc = g2.wait()
    checks["converged: one gather per step"] = len(seen) == W + K and eng.launches == list(range(W + K)) and not graph3
    checks["converged: own slab"] = torch.equal(lastc["x"][rank], eng.ts["x"][W + K - 1]) and torch.equal(lastc["kkt"][rank], eng.ts["kkt"][W + K - 1])
    e3 = OracleEngine(B, N, 1); e3.load(other); e3.rti(2, 0)
    checks["converged: peer slab"] = torch.equal(lastc["x"][peer], e3.ts["x"][0]) and torch.equal(lastc["u"][peer], e3.ts["u"][0])
    checks["converged: every problem of every rank"] = tuple(lastc["status"].shape) == (world, B) and int(lastc["status"].sum()) == 0
    e1 = OracleEngine(B, N, 1); e1.load(batch); e1.rti(1, 0)
    checks["converged: two iterations differ from one"] = not torch.equal(e1.ts["u"][0], eng.ts["u"][W + K - 1])
    # ... and with the steps of a bucket in one grid / one gather per bucket (the engine here has no rti_range: one launch per step)
    submits.clear(); eng.launches.clear()
    g3 = ResultGatherer(dist, world, depth=2)
    seen3 = []
    orig3 = g3.submit
    g3.submit = lambda t: (seen3.append({k: tuple(v.shape) for k, v in t.items()}), orig3(t))[1]
    el4, dms4, graph4 = timed_pass(eng, batch, "converged_in_flight", K, W, ge, g3, hooks, world, conv_iters=2)
    lastf = g3.wait()
    n_buckets = -(-W // ge) + -(-K // ge)
    checks["in flight: one gather per bucket"] = len(seen3) == n_buckets and eng.launches == list(range(W + K)) and not graph4
    checks["in flight: own slab"] = torch.equal(lastf["x"][rank][-1], eng.ts["x"][W + K - 1]) and torch.equal(lastf["kkt"][rank][-1], eng.ts["kkt"][W + K - 1])
    checks["in flight: peer slab"] = torch.equal(lastf["x"][peer][-1], e3.ts["x"][0]) and torch.equal(lastf["u"][peer][-1], e3.ts["u"][0])
    checks["in flight: the same trajectories as one step at a time"] = torch.equal(lastf["x"][:, -1], lastc["x"]) and torch.equal(lastf["status"][:, -1], lastc["status"])
    # secondary pass that fails on every rank before any collective: caught, the primary figure stands
    class Boom(OracleEngine):
        def load(self, batch, slot=None):
            raise RuntimeError("secondary pass failed")
    alt = None
    try:
        timed_pass(Boom(B, N, W + K), batch, "last", K, W, ge, g, hooks, world)
    except Exception as e:  # bench.py reports it in the line
        alt = {"error": f"{type(e).__name__}: {e}"}
    checks["error path"] = alt is not None and "secondary pass failed" in alt["error"]
    ok = all(checks.values())
    if not ok:
        print(f"rank {rank}: failed checks {[k for k, v in checks.items() if not v]}", flush=True)
    dist.barrier()   # the ranks are still in step after the failure
    if rank == 0:
        print("BENCH_PASS_OK" if ok else "BENCH_PASS_MISMATCH", flush=True)
    return ok


def main():
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    N, total, steps = 20, 22, 3  # 22 problems: 11 + 11 over 2 ranks, 8 + 7 + 7 over 3; steps = buckets of different sizes
    off, cnt = partition(total, world, rank)
    counts = [partition(total, world, r)[1] for r in range(world)]
    assert counts == ([11, 11] if world == 2 else [8, 7, 7] if world == 3 else counts) and off == sum(counts[:rank]) and cnt == counts[rank]
    rows = max(counts)  # ranks with one problem fewer pad their slab: the collective wants equal shapes
    g = ResultGatherer(dist, world)
    xs, us, sts = [], [], []
    for s in range(steps):
        batch = make_batch(cnt, N, seed=1000 + s, offset=off)
        x, u, st = solve(batch, N)
        xs.append(pad_rows(torch.from_numpy(x), rows)); us.append(pad_rows(torch.from_numpy(u), rows)); sts.append(pad_rows(torch.from_numpy(st), rows))
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
            gx = concat_gathered(src["x"][:, i], counts).numpy()   # [world, rows, ...] -> the global batch in problem order
            gu = concat_gathered(src["u"][:, i], counts).numpy()
            gs = concat_gathered(src["status"][:, i], counts).numpy()
            ok = ok and gx.shape[0] == total and np.array_equal(gx, x) and np.array_equal(gu, u) and np.array_equal(gs, st)
        print("GATHER_OK" if ok else "GATHER_MISMATCH", flush=True)
    ok = bench_pass_bookkeeping(rank, world) and ok
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
