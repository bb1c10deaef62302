"""One traced K-batch grid of the stage-block kernel (after an untraced-size warm-up grid) -> per-SIMD timeline.

    ALORE_NMPC_TRACE=/tmp/tr python tools/trace_grid.py [K] [reps]     (ALORE_NMPC_STAGGER_NS=<ns> to move the stagger)
"""
import glob
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from alore_legged_manipulator_amd.nmpc import BatchedNmpc  # noqa: E402
from alore_legged_manipulator_amd.scenarios import make_batch  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
prefix = os.environ["ALORE_NMPC_TRACE"]
B, N, W = 4096, 20, 5
eng = BatchedNmpc(B, N, device=0, slots=K + W)
eng.set_launch_overlap(16)
batch = make_batch(B, N)
for r in range(reps):
    eng.load(batch, slot=None)
    torch.cuda.synchronize()
    eng.rti_range(0, W)
    torch.cuda.synchronize()
    eng.rti_range(W, K)
    torch.cuda.synchronize()
files = sorted((f for f in glob.glob(prefix + ".*") if f.rsplit(".", 1)[1].isdigit()), key=lambda f: int(f.rsplit(".", 1)[1]))
for f in files[1::2]:  # the K-batch grids
    print(f"==== {os.path.basename(f)}")
    sys.stdout.flush()
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "trace_timeline.py"), f])
for f in files:
    os.remove(f)
