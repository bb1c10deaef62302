"""Sharding of a batch of independent NMPC problems over the GPUs of one node.

Problems are independent: rank r of G owns the
contiguous block [offset, offset+count) of the global problem index, generates
or receives its own inputs, and solves with no communication.  The only
collective is the collection of the results: one all-gather of per-rank result
slabs (x, u, status, kkt), issued asynchronously and in buckets of several
solve steps so that it overlaps with the following solves (xGMI is
point-to-point: a few large collectives beat many small ones).

Backend-agnostic: with backend "nccl" (RCCL on ROCm) the tensors are GPU
tensors; the CPU tests run the same code with "gloo".
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple


def partition(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block partition; the first (total % world) ranks get one extra."""
    assert 0 <= rank < world and total >= 0
    base, rem = divmod(total, world)
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


def pad_rows(t, rows: int):
    """`t` with its leading dimension padded with zeros to `rows`: ranks of an uneven partition (partition(): the first
    total % world ranks own one problem more) bring equal shapes to the all-gather"""
    import torch
    if t.shape[0] == rows:
        return t
    pad = torch.zeros((rows - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    return torch.cat([t, pad], dim=0)


def concat_gathered(out, counts: List[int], dim: int = 0):
    """the padded slabs of an all-gather ([world, rows_max, ...]) cut back to the ranks' real counts and joined in rank order:
    the global batch in its global problem order"""
    import torch
    return torch.cat([out[r].narrow(dim, 0, c) for r, c in enumerate(counts)], dim=dim)


class ResultGatherer:
    """All-gather of result tensors with equal per-rank shapes.

    ``submit(tensors)`` starts one asynchronous all-gather per tensor and
    returns immediately; ``wait()`` blocks until every outstanding collective has
    finished and returns the gathered tensors of the LAST submit (shape
    [world, *local_shape])."""

    def __init__(self, dist, world: int, depth: int = 1):
        self.dist = dist
        self.world = world
        self.depth = max(1, depth)   # output buffers per tensor, used in turn: with 2 the previous gather stays readable while the next lands
        self._turn = 0
        self._pending: List = []
        self._last: Optional[Dict[str, object]] = None
        self._out_cache: Dict[Tuple, object] = {}

    def _out_like(self, name: str, t):
        import torch
        key = (name, tuple(t.shape), t.dtype, str(t.device), self._turn % self.depth)
        out = self._out_cache.get(key)
        if out is None:
            out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
            self._out_cache[key] = out
        return out

    def submit(self, tensors: Dict[str, object]) -> None:
        gathered = {}
        for name, t in tensors.items():
            t = t.contiguous()
            out = self._out_like(name, t)
            if self.world == 1:
                out[0].copy_(t)
            else:
                w = self.dist.all_gather_into_tensor(out.view(-1), t.view(-1), async_op=True)
                self._pending.append(w)
            gathered[name] = out
        self._turn += 1
        self._last = gathered

    def wait(self) -> Optional[Dict[str, object]]:
        for w in self._pending:
            w.wait()
        self._pending.clear()
        return self._last


class HostHooks:
    """Device hooks of `timed_pass` for a machine without GPUs (the gloo tests): nothing to synchronise, no graphs,
    the device clock is the host clock."""

    def __init__(self, dist=None, world: int = 1):
        self.dist, self.world = dist, world

    def sync(self) -> None:
        pass

    def barrier(self) -> None:
        if self.world > 1:
            self.dist.barrier()
        self.sync()

    def capture(self, fn):
        """record fn() once and return a zero-argument replay callable, or None to launch eagerly"""
        return None

    def device_timer(self):
        import time

        class T:
            def start(self_inner):
                self_inner.t0 = time.perf_counter()

            def stop(self_inner):
                self_inner.t1 = time.perf_counter()

            def elapsed_ms(self_inner):
                return (self_inner.t1 - self_inner.t0) * 1e3
        return T()

    def max_over_ranks(self, values):
        if self.world == 1:
            return list(values)
        import torch
        t = torch.tensor(list(values), dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return [float(v) for v in t]


def timed_pass(eng, batch, mode: str, K: int, warmup: int, gather_every: int, gatherer, hooks, world: int, conv_iters: int = 15):
    """W untimed + K timed steps of `eng` (anything with load(batch, slot=None), rti(1, slot=i) and per-slot result
    tensors eng.ts[name][slot]) with the result exchange `mode`:
      full -- the trajectories (x, u, status, kkt) of EVERY timed batch are all-gathered to every rank, in buckets of
              `gather_every` steps issued asynchronously and all completed inside the timed region (eager launches:
              collectives sit between the solves);
      last -- only the last batch's trajectories are exchanged (the K launches may replay as one graph);
      none -- results stay sharded;
      converged -- the north star's unit: every step solves its batch to convergence (`conv_iters` real-time iterations inside
              one launch, the cold start of MpcWrapper::solve + its update() calls) and the converged trajectories (x, u,
              status, kkt) are all-gathered to every rank, one collective per tensor per step, issued asynchronously right
              behind the solve so that it runs under the next step's solve, all completed inside the timed region.
      converged_in_flight -- the same unit with the steps of a bucket (`gather_every` independent batches) solved by ONE grid
              (alore_nmpc_rti_many with `conv_iters` iterations: the packed lane mapping fills the chip, as the headline's pass does
              for single iterations) and ONE all-gather per tensor and bucket behind it, running under the next bucket's grid.
    Returns (elapsed seconds, device milliseconds, graph used): both times are the MAXIMUM over the ranks, measured
    between two barriers.  bench.py calls this with the GPU hooks; tests/dist_worker.py with HostHooks under gloo."""
    import time
    do_gather = world > 1 and mode == "full"
    gather_last = world > 1 and mode == "last"
    converged = mode == "converged"
    converged_many = mode == "converged_in_flight"
    ge = max(1, gather_every)

    def run_steps(first, count):
        if converged_many:
            lo = first
            while lo < first + count:
                n = min(ge, first + count - lo)
                if hasattr(eng, "rti_range"):
                    eng.rti_range(lo, n, conv_iters)
                else:
                    for i in range(lo, lo + n):
                        eng.rti(conv_iters, slot=i)
                if gatherer is not None:
                    gatherer.submit({k: eng.ts[k][lo:lo + n] for k in ("x", "u", "status", "kkt")})
                lo += n
            if gatherer is not None:
                gatherer.wait()
            return
        if converged:
            for i in range(first, first + count):
                eng.rti(conv_iters, slot=i)
                if gatherer is not None:
                    gatherer.submit({k: eng.ts[k][i] for k in ("x", "u", "status", "kkt")})
            if gatherer is not None:
                gatherer.wait()
            return
        if not do_gather and count > 0 and hasattr(eng, "rti_range"):
            eng.rti_range(first, count)  # no collective between the solves: the launches go out from one library call
            return
        for i in range(first, first + count):
            eng.rti(1, slot=i)
            if do_gather and ((i - first + 1) % ge == 0 or i == first + count - 1):
                lo = first + ((i - first) // ge) * ge
                gatherer.submit({k: eng.ts[k][lo:i + 1] for k in ("x", "u", "status", "kkt")})
        if do_gather:
            gatherer.wait()

    eng.load(batch, slot=None)  # every pass starts from the same cold-start iterates in every slot
    hooks.sync()
    run_steps(0, warmup)
    hooks.barrier()
    replay = None
    if not do_gather and not converged and not converged_many and K > 0:  # never with collectives between the solves
        if hasattr(eng, "prepare_range"):
            # the independence check of the K timed slots before the timed region, for eager launches as a stream capture
            # has it (alore_nmpc_rti_many_prepare: a host that steps the same slots every tick pays for it once)
            eng.prepare_range(warmup, K)
        replay = hooks.capture(lambda: run_steps(warmup, K))
    timer = hooks.device_timer()
    hooks.barrier()
    t0 = time.perf_counter()
    timer.start()
    t_a = time.perf_counter()
    if replay is not None:
        replay()
    else:
        run_steps(warmup, K)
    t_b = time.perf_counter()
    timer.stop()
    t_c = time.perf_counter()
    if gather_last:  # the converged trajectories of the last batch on every rank (x, u, status, kkt)
        last = warmup + K - 1
        gatherer.submit({k: eng.ts[k][last] for k in ("x", "u", "status", "kkt")})
        gatherer.wait()
    hooks.barrier()
    t1 = time.perf_counter()
    # where the host clock of the region goes (this rank): first event, the launch call(s), second event, waiting for the device
    hooks.last_host_breakdown_us = {"start_event": (t_a - t0) * 1e6, "launch_calls": (t_b - t_a) * 1e6, "stop_event": (t_c - t_b) * 1e6,
                                    "wait": (t1 - t_c) * 1e6}
    el, dms = hooks.max_over_ranks([t1 - t0, timer.elapsed_ms()])
    return el, dms, replay is not None
