#!/usr/bin/env python3
"""Whole-body class, B = 4096, N = 20: a few real-time iterations for rocprofv3 (kernel trace / PMC passes).
    rocprofv3 --kernel-trace --stats -d out -o wb -- python3 tools/wb_profile.py
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d out -o wbpmc -- python3 tools/wb_profile.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from alore_legged_manipulator_amd.whole_body import BatchedWholeBody  # noqa: E402
from wb_cases import make_problems_fast, weights  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N, iters = 20, 5
eng = BatchedWholeBody(B, N, 0.01)
x0, xref, uref, xi, ui = make_problems_fast(B, N, seed=3)
eng.set_weights(*weights())
eng.set_problem(x0, xref, uref)
for it in range(iters):
    eng.set_iterate(xi, ui)
    eng.rti(1)
    lin, ric = eng.last_times()
    print(f"iter {it}: linearise {lin:.3f} ms  riccati {ric:.3f} ms  -> {B / ((lin + ric) * 1e-3):.0f} solves/s")
dx, du = eng.last_step()
print("max |dx|", float(np.max(np.abs(dx))), "finite", bool(np.isfinite(dx).all()))
