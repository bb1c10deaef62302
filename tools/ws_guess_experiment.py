"""Driver of tools/ws_guess_experiment.cpp: sweeps after the first one, per problem and per wavefront of 16 problems
(the (4, 5) mapping: a wavefront's restart costs the highest restart stage over its 16 problems), for the bench batch."""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from alore_legged_manipulator_amd.scenarios import make_batch, make_wide_batch  # noqa: E402

so = "/tmp/libws_guess_experiment.so"
subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", os.path.join(ROOT, "tools/ws_guess_experiment.cpp"), "-o", so])
lib = ctypes.CDLL(so)
fp = ctypes.POINTER(ctypes.c_float)
ip = ctypes.POINTER(ctypes.c_int)


def run(batch, N, mode, est, passes, G=16, S=5):
    B = batch["x"].shape[0]
    sweeps = np.zeros(B, int)
    rest = np.full((B, 32), -1, int)
    nact = np.zeros(B, int)
    r = (ctypes.c_int * 32)()
    na = ctypes.c_int()
    P = lambda a: a.ctypes.data_as(fp)
    for b in range(B):
        a = {k: np.ascontiguousarray(v[b]).reshape(-1) for k, v in batch.items()}
        n = lib.ws_experiment(N, ctypes.c_float(0.01), P(a["x"]), P(a["u"]), P(a["od"]), P(a["y"]), P(a["yN"]), P(a["W"]), P(a["WN"]),
                              P(a["x0"]), P(a["lbValues"]), P(a["ubValues"]), mode, est, passes, r, ctypes.byref(na))
        sweeps[b] = n
        rest[b, :min(n - 1, 32)] = list(r)[:min(n - 1, 32)]
        nact[b] = na.value
    # wavefront model: stage steps after the first sweep = sum over rounds of (max over the 16 problems of the restart lane + 1) * S
    nw = B // G
    extra = np.zeros(nw)
    rounds = np.zeros(nw, int)
    for w in range(nw):
        rr = rest[w * G:(w + 1) * G]
        for i in range(32):
            col = rr[:, i]
            if (col < 0).all():
                break
            extra[w] += (col.max() // S + 1) * S
            rounds[w] += 1
    return sweeps, nact, extra, rounds


if __name__ == "__main__":
    B, N = 4096, 20
    which = sys.argv[1] if len(sys.argv) > 1 else "bench"
    batch = make_batch(B, N) if which == "bench" else make_wide_batch(B, N, 7)
    for mode, est, passes in ((0, 0, 0), (3, 0, 0), (0, 106, 9), (3, 106, 9), (3, 100, 1), (3, 100, 2), (3, 100, 3), (3,100,4), (3, 100, 5), (0, 100, 3)):
        sw, na, extra, rounds = run(batch, N, mode, est, passes)
        print(f"mode {mode} est {est} passes {passes}: problems with active bounds {np.mean(na > 0):.3f}; sweeps hist {np.bincount(sw)[:10]}; "
              f"wavefronts needing >1 round {np.mean(rounds > 0):.3f}, mean extra stage steps per wavefront {extra.mean():.2f} (first sweep = {N}), "
              f"max rounds {rounds.max()}")
