"""Single-instance ACADO-compatible ABI (libalore_acado_compat.so): the acado_* symbols the reference's
MpcWrapper links against, served by the GPU engine with B = 1.  The caller side (two global structs) is
tests/harness/acado_caller.c.  Driven exactly like mpc_wrapper.cpp drives the generated solver,
including solve()'s reset-between-preparation-and-feedback, and compared with the oracle (and with the
compiled reference where oracle/_ref is present) doing the same calls."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from alore_legged_manipulator_amd.scenarios import make_batch, problem
from oracle.drivers import Oracle, RefAcado, ref_available

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "alore_legged_manipulator_amd")
CALLER_SO = os.path.join(ROOT, "tests", "harness", "libacado_caller.so")
FP = C.POINTER(C.c_float)


def build_caller():
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", CALLER_SO,
                           os.path.join(ROOT, "tests", "harness", "acado_caller.c"),
                           "-L" + PKG, "-lalore_acado_compat", "-Wl,-rpath," + PKG])


def test_struct_sizes_match_the_reference_layout():
    """sizeof(ACADOvariables / ACADOworkspace) as measured on the reference (8488 / 86296 B)."""
    build_caller()
    import torch  # noqa: F401  one HIP runtime per process
    L = C.CDLL(CALLER_SO, mode=C.RTLD_GLOBAL)
    assert L.caller_sizeof_variables() == 8488
    assert L.caller_sizeof_workspace() == 86296
    if ref_available():
        R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libacado_ref.so"))
        assert R.ref_sizeof_variables() == 8488 and R.ref_sizeof_workspace() == 86296


class Compat:
    N = 50

    def __init__(self):
        build_caller()
        import torch  # noqa: F401
        self.L = C.CDLL(CALLER_SO, mode=C.RTLD_GLOBAL)
        self.S = C.CDLL(os.path.join(PKG, "libalore_acado_compat.so"), mode=C.RTLD_GLOBAL)
        sizes = {"x": 153, "u": 100, "od": 153, "y": 250, "yN": 3, "W": 1250, "WN": 9, "x0": 3, "lbValues": 100,
                 "ubValues": 100, "d": 150, "evGx": 450, "evGu": 300, "dx": 100, "dual": 100, "lb": 100, "ub": 100, "H": 10000, "g": 100}
        names = {"d": "ws_d", "evGx": "ws_evGx", "evGu": "ws_evGu", "dx": "ws_x", "dual": "ws_y", "lb": "ws_lb", "ub": "ws_ub", "H": "ws_H", "g": "ws_g"}
        self.v = {}
        for k, n in sizes.items():
            fn = getattr(self.L, "caller_" + names.get(k, k))
            fn.restype = FP
            self.v[k] = np.ctypeslib.as_array(fn(), shape=(n,))
        self.S.acado_getKKT.restype = C.c_float
        self.S.acado_getObjective.restype = C.c_float

    def reset(self): self.L.caller_reset()
    def initialize_solver(self): return self.S.acado_initializeSolver()
    def initialize_nodes_by_forward_simulation(self): self.S.acado_initializeNodesByForwardSimulation()
    def preparation_step(self): return self.S.acado_preparationStep()
    def feedback_step(self): return self.S.acado_feedbackStep()
    def get_kkt(self): return float(self.S.acado_getKKT())
    def get_nwsr(self): return int(self.S.acado_getNWSR())
    def solve(self): return int(self.S.acado_solve())


def wrapper_sequence(s, p, ticks):
    """What MpcWrapper does: ctor (mpc_wrapper.cpp:33-93), then solve() on the first tick
    (:267-275: reset x/u AFTER the preparation, then feedback + preparation) and update() afterwards."""
    N = s.N
    s.reset()
    s.initialize_solver()
    s.v["W"][:] = p["W"]; s.v["WN"][:] = p["WN"]
    s.v["od"][:] = np.tile(np.float32([0.0, -0.2, 0.2]), N + 1)      # default ICR
    s.initialize_nodes_by_forward_simulation()
    s.preparation_step()
    out = []
    for t in range(ticks):
        s.v["od"][:] = p["od"]                                        # setICRParameters
        s.v["y"][:] = p["y"]; s.v["yN"][:] = p["yN"]                 # setTrajectory
        if t == 0:                                                     # solve(): reset the iterate
            s.v["x"][:] = np.tile(p["x0"], N + 1)
            s.v["u"][:] = 0
        s.v["x0"][:] = p["x0"]
        st = s.feedback_step()
        s.preparation_step()
        out.append((st, s.v["x"].copy(), s.v["u"].copy(), s.v["dual"].copy()))
    return out


def relerr(a, b):
    return float(np.max(np.abs(a - b)) / max(1.0, np.max(np.abs(b))))


@pytest.mark.gpu
def test_wrapper_call_sequence_matches_reference_semantics():
    N = 50
    batch = make_batch(6, N, seed=77, fast_tail=0.4)
    comp = Compat()
    ref = RefAcado() if ref_available() else None
    orc = Oracle(N)
    for b in range(6):
        p = problem(batch, b)
        got = wrapper_sequence(comp, p, 4)
        for name, s in (("oracle", orc), ("reference", ref)):
            if s is None:
                continue
            exp = wrapper_sequence(s, p, 4)
            for t, (g, e) in enumerate(zip(got, exp)):
                assert g[0] == e[0] == 0, (name, b, t)
                # tick 0 uses the stale linearisation (prepared before the reset): must agree as well
                assert relerr(g[1], e[1]) < 1e-4, (name, b, t, "x")
                assert relerr(g[2], e[2]) < 1e-4, (name, b, t, "u")
    # the preparation-side workspace members are kept up to date
    orc.preparation_step()
    comp.preparation_step()
    assert np.max(np.abs(comp.v["d"] - orc.v["d"])) < 5e-6
    assert np.max(np.abs(comp.v["evGu"] - orc.v["evGu"])) < 5e-6


@pytest.mark.gpu
def test_acado_solve_and_the_condensed_workspace_members():
    """acadoWorkspace.H / g after acado_preparationStep + acado_feedbackStep are the condensed QP the reference leaves there
    (CG/acado_solver.c:327-891; compared with the oracle, bit-exact with the reference at N = 50, and with the compiled reference
    itself where it is present), and acado_solve() (CG/acado_qpoases_interface.cpp:39-60) solves the QP that is in the
    workspace: called again after the feedback step it returns the step the feedback step took; with a changed gradient it
    returns the minimiser of the changed QP (checked in float64)."""
    from scipy.optimize import lsq_linear
    N = 50
    batch = make_batch(3, N, seed=91, fast_tail=0.5)
    comp = Compat()
    orc = Oracle(N)
    ref = RefAcado() if ref_available() else None
    for b in range(3):
        p = problem(batch, b)
        for s in (comp, orc, ref):
            if s is None:
                continue
            s.reset(); s.initialize_solver()
            for k in ("x", "u", "od", "y", "yN", "W", "WN", "x0"):
                s.v[k][:] = p[k]
            assert s.preparation_step() == 0 and s.feedback_step() == 0
        H, g = orc.v["H"].reshape(100, 100), orc.v["g"]
        assert np.max(np.abs(comp.v["H"].reshape(100, 100) - H)) < 1e-5 * np.max(np.abs(H))
        assert np.max(np.abs(comp.v["g"] - g)) < 1e-5 * max(1.0, np.max(np.abs(g)))
        if ref is not None:
            assert np.max(np.abs(comp.v["H"] - ref.v["H"])) < 1e-5 * np.max(np.abs(ref.v["H"]))
            assert np.max(np.abs(comp.v["g"] - ref.v["g"])) < 1e-5 * max(1.0, np.max(np.abs(ref.v["g"])))
        step = comp.v["dx"].copy()
        assert comp.solve() == 0 and comp.get_nwsr() >= 1
        assert np.max(np.abs(comp.v["dx"] - step)) < 1e-4 * max(1.0, np.max(np.abs(step)))
        # a QP the caller changed: the minimiser of what is in the workspace
        comp.v["g"][:] = comp.v["g"] * 0.5 + 0.3
        Hc = comp.v["H"].reshape(100, 100).astype(np.float64); Hc = 0.5 * (Hc + Hc.T)
        Lc = np.linalg.cholesky(Hc)
        tr = lsq_linear(Lc.T, -np.linalg.solve(Lc, comp.v["g"].astype(np.float64)), bounds=(comp.v["lb"].astype(np.float64), comp.v["ub"].astype(np.float64)),
                        method="bvls", tol=1e-15, max_iter=2000).x
        assert comp.solve() == 0
        assert np.max(np.abs(comp.v["dx"] - tr)) < 1e-4 * max(1.0, np.max(np.abs(tr)))


@pytest.mark.gpu
def test_dense_workspace_refresh_can_be_switched_off():
    """alore_acado_dense_workspace(0): the preparation / feedback steps no longer touch acadoWorkspace.H / g (a caller like
    MpcWrapper never reads them: one launch and 40 KB over the bus less per step); the tick itself is unchanged; switched on
    again they are refreshed."""
    N = 50
    batch = make_batch(1, N, seed=17, fast_tail=0.5)
    p = problem(batch, 0)
    comp = Compat()
    comp.S.alore_acado_dense_workspace.argtypes = [C.c_int]
    comp.S.alore_acado_dense_workspace.restype = None

    def tick():
        comp.reset(); comp.initialize_solver()
        for k in ("x", "u", "od", "y", "yN", "W", "WN", "x0"):
            comp.v[k][:] = p[k]
        assert comp.preparation_step() == 0 and comp.feedback_step() == 0
        return comp.v["u"].copy(), comp.v["H"].copy(), comp.v["g"].copy()

    u_on, H_on, g_on = tick()
    assert np.max(np.abs(H_on)) > 0.0
    comp.S.alore_acado_dense_workspace(0)
    try:
        comp.v["H"][:] = 7.0; comp.v["g"][:] = -3.0
        # caller_reset() clears the structs: fill the markers after it, inside the sequence
        comp.reset(); comp.initialize_solver()
        for k in ("x", "u", "od", "y", "yN", "W", "WN", "x0"):
            comp.v[k][:] = p[k]
        comp.v["H"][:] = 7.0; comp.v["g"][:] = -3.0
        assert comp.preparation_step() == 0 and comp.feedback_step() == 0
        assert np.all(comp.v["H"] == 7.0) and np.all(comp.v["g"] == -3.0)
        assert np.array_equal(comp.v["u"], u_on)
    finally:
        comp.S.alore_acado_dense_workspace(1)
    u_again, H_again, g_again = tick()
    assert np.array_equal(H_again, H_on) and np.array_equal(g_again, g_on) and np.array_equal(u_again, u_on)
