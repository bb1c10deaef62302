"""Multi-rank path on CPU: world_size 2, gloo, one process per rank launched the
way the driver launches bench.py (torch.distributed.run, rendezvous on 127.0.0.1)."""
import os
import subprocess
import sys

from alore_legged_manipulator_amd.shard import partition

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_is_a_contiguous_cover():
    for total in (0, 1, 7, 4096, 262144, 1000003):
        for world in (1, 2, 3, 8):
            blocks = [partition(total, world, r) for r in range(world)]
            assert blocks[0][0] == 0
            for (o0, c0), (o1, _) in zip(blocks, blocks[1:]):
                assert o0 + c0 == o1
            assert blocks[-1][0] + blocks[-1][1] == total
            assert max(c for _, c in blocks) - min(c for _, c in blocks) <= 1


def test_two_rank_gather_matches_single_process():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "tests", "dist_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "GATHER_OK" in r.stdout
    assert "BENCH_PASS_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]   # bench.py's timed passes under gloo


def test_three_ranks_uneven_partition():
    """world_size 3, 22 problems -> 8 / 7 / 7: the ranks with one problem fewer pad their slabs for the all-gather, rank 0 cuts the
    gathered slabs back to the real counts and finds the single-process solve of the whole batch; bench.py's timed passes
    (every-batch buckets, last batch, converged solve + gather) run with three ranks too."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3",
           "--master-addr", "127.0.0.1", "--master-port", "29519", os.path.join(ROOT, "tests", "dist_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "GATHER_OK" in r.stdout
    assert "BENCH_PASS_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
