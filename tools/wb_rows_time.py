"""Whole-body class, B = 4096, N = 20: milliseconds of the linearisation (+ row elimination) and of the sweep with the hard contact
rows on and off, and of one real-time iteration in the exact working-set mode (with its number of sweeps)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from alore_legged_manipulator_amd.whole_body import BatchedWholeBody  # noqa: E402
from wb_cases import make_problems_fast, weights  # noqa: E402

B, N = 4096, 20
e = BatchedWholeBody(B, N, 0.01)
x0, xref, uref, xi, ui = make_problems_fast(B, N, seed=3)
e.set_weights(*weights()); e.set_problem(x0, xref, uref)
for rows in (False, True, False, True):
    e.set_contact_rows(rows)
    e.set_iterate(xi, ui); e.rti(1); torch.cuda.synchronize()
    e.set_iterate(xi, ui); e.rti(1); torch.cuda.synchronize()
    lin, ric = e.last_times()
    print(f"contact rows {rows}: linearise (+ rows) {lin:.3f} ms, sweep (+ forces) {ric:.3f} ms, status ok {bool((e.status() == 0).all())}")
e.set_contact_rows(False)
if len(sys.argv) > 1:
    e.set_contact_constraints(True, 0.7)
    e.set_constraint_mode(True, int(sys.argv[1]))
    e.set_iterate(xi, ui)
    t0 = time.perf_counter(); e.rti(1); torch.cuda.synchronize(); t1 = time.perf_counter()
    sweeps, open_, ws, gu = e.working_set_info(B)
    print(f"exact mode: {sweeps} sweeps, {(t1 - t0) * 1e3:.1f} ms for one real-time iteration of {B} problems, {int(open_.sum())} problems still open, "
          f"held inputs per problem mean {float((ws[:, :, :30] != 0).sum() / B):.1f}")
