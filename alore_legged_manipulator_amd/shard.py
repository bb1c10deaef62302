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


class ResultGatherer:
    """All-gather of result tensors with equal per-rank shapes.

    ``submit(tensors)`` starts one asynchronous all-gather per tensor and
    returns immediately; ``wait()`` blocks until every outstanding collective has
    finished and returns the gathered tensors of the LAST submit (shape
    [world, *local_shape])."""

    def __init__(self, dist, world: int):
        self.dist = dist
        self.world = world
        self._pending: List = []
        self._last: Optional[Dict[str, object]] = None
        self._out_cache: Dict[Tuple, object] = {}

    def _out_like(self, name: str, t):
        import torch
        key = (name, tuple(t.shape), t.dtype, str(t.device))
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
        self._last = gathered

    def wait(self) -> Optional[Dict[str, object]]:
        for w in self._pending:
            w.wait()
        self._pending.clear()
        return self._last
