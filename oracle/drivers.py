"""ctypes drivers for the two CPU checkers.  TEST INFRASTRUCTURE ONLY.

* ``Oracle``   -- our horizon-generic C restatement (oracle/nmpc_oracle.c).
* ``RefAcado`` -- the reference's own ACADO-generated solver + vendored qpOASES,
  compiled from /root/reference by ``make -C oracle ref`` into
  ``oracle/_ref/libacado_ref.so`` (N = 50 only, process-global state).

Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg may
import this module; the product package never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle_nmpc.so")
REF_SO = os.path.join(HERE, "_ref", "libacado_ref.so")

VAR_FIELDS = ("x", "u", "od", "y", "yN", "W", "WN", "x0", "lbValues", "ubValues")
WS_FIELDS = ("d", "Dy", "DyN", "evGx", "evGu", "Q1", "Q2", "R1", "R2", "QN1", "QN2", "sbar",
             "Dx0", "E", "QDy", "H", "g", "lb", "ub", "dx", "dual")


def build(ref: bool = True) -> None:
    """(Re)build the checker libraries with oracle/Makefile."""
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    if ref:
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


def field_sizes(N: int) -> dict:
    n = 2 * N
    return {
        "x": 3 * (N + 1), "u": 2 * N, "od": 3 * (N + 1), "y": 5 * N, "yN": 3, "W": 25 * N,
        "WN": 9, "x0": 3, "lbValues": n, "ubValues": n,
        "d": 3 * N, "Dy": 5 * N, "DyN": 3, "evGx": 9 * N, "evGu": 6 * N, "Q1": 9 * N,
        "Q2": 15 * N, "R1": 4 * N, "R2": 10 * N, "QN1": 9, "QN2": 9, "sbar": 3 * (N + 1),
        "Dx0": 3, "E": N * (N + 1) // 2 * 6, "QDy": 3 * (N + 1), "H": n * n, "g": n,
        "lb": n, "ub": n, "dx": n, "dual": n, "rk_kkk": 6,
    }


class _SolverBase:
    """Common numpy-view plumbing: ``self.v[name]`` is a live float32 view."""

    N: int
    v: dict

    def load(self, prob: dict) -> None:
        """Copy a problem dict (keys of VAR_FIELDS plus optional 'dual') in."""
        for k in VAR_FIELDS:
            if k in prob:
                self.v[k][:] = np.asarray(prob[k], dtype=np.float32).ravel()
        if "dual" in prob:
            self.v["dual"][:] = np.asarray(prob["dual"], dtype=np.float32).ravel()

    def snapshot(self, names=None) -> dict:
        names = names or (VAR_FIELDS + WS_FIELDS)
        return {k: self.v[k].copy() for k in names}

    # one reference tick as the node runs it (mpc_wrapper.cpp:279-373 preceded
    # by the preparation of the previous tick): prepare on (x,u), then feedback
    def rti(self):
        st_prep = self.preparation_step()
        st_fb = self.feedback_step()
        return st_prep, st_fb


class Oracle(_SolverBase):
    def __init__(self, N: int, dt: float = 0.01):
        if not os.path.exists(ORACLE_SO):
            build(ref=False)
        L = C.CDLL(ORACLE_SO)
        self.L = L
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int, C.c_double]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_ptr.restype = C.POINTER(C.c_float)
        L.orc_ptr.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]
        for f in ("orc_model_simulation", "orc_solve_qp", "orc_initialize_solver",
                  "orc_preparation_step", "orc_feedback_step", "orc_get_nwsr"):
            getattr(L, f).restype = C.c_int
            getattr(L, f).argtypes = [C.c_void_p]
        for f in ("orc_evaluate_objective", "orc_condense_prep", "orc_condense_fdb", "orc_expand",
                  "orc_initialize_nodes_by_forward_simulation"):
            getattr(L, f).restype = None
            getattr(L, f).argtypes = [C.c_void_p]
        for f in ("orc_get_kkt", "orc_get_objective"):
            getattr(L, f).restype = C.c_float
            getattr(L, f).argtypes = [C.c_void_p]
        L.orc_integrate.restype = C.c_int
        L.orc_integrate.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int]
        L.orc_shift_states.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_shift_controls.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.orc_time_rti.restype = C.c_double
        L.orc_time_rti.argtypes = [C.c_void_p, C.c_int]
        L.orc_qpb_solve.restype = C.c_int
        L.orc_qpb_solve.argtypes = [C.c_int] + [C.POINTER(C.c_float)] * 5 + [C.POINTER(C.c_int)] + \
            [C.POINTER(C.c_float)] * 2
        L.orc_rhs.argtypes = [C.POINTER(C.c_float)] * 2
        L.orc_diffs.argtypes = [C.POINTER(C.c_float)] * 2
        self.N = N
        self.h = L.orc_create(N, dt)
        if not self.h:
            raise MemoryError("orc_create failed")
        self.v = {}
        for name, cnt in field_sizes(N).items():
            ln = C.c_int(0)
            p = L.orc_ptr(self.h, name.encode(), C.byref(ln))
            assert p and ln.value == cnt, (name, ln.value, cnt)
            self.v[name] = np.ctypeslib.as_array(p, shape=(cnt,))

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.L.orc_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def reset(self):
        for a in self.v.values():
            a[:] = 0

    def initialize_solver(self): return self.L.orc_initialize_solver(self.h)
    def initialize_nodes_by_forward_simulation(self): self.L.orc_initialize_nodes_by_forward_simulation(self.h)
    def model_simulation(self): return self.L.orc_model_simulation(self.h)
    def evaluate_objective(self): self.L.orc_evaluate_objective(self.h)
    def condense_prep(self): self.L.orc_condense_prep(self.h)
    def condense_fdb(self): self.L.orc_condense_fdb(self.h)
    def solve_qp(self): return self.L.orc_solve_qp(self.h)
    def expand(self): self.L.orc_expand(self.h)
    def preparation_step(self): return self.L.orc_preparation_step(self.h)
    def feedback_step(self): return self.L.orc_feedback_step(self.h)
    def get_kkt(self): return float(self.L.orc_get_kkt(self.h))
    def get_objective(self): return float(self.L.orc_get_objective(self.h))
    def get_nwsr(self): return int(self.L.orc_get_nwsr(self.h))
    def time_rti(self, iters): return float(self.L.orc_time_rti(self.h, iters))

    def integrate(self, eta: np.ndarray, reset: int = 1) -> int:
        assert eta.dtype == np.float32 and eta.size == 23
        return self.L.orc_integrate(self.h, eta.ctypes.data_as(C.POINTER(C.c_float)), reset)

    def shift_states(self, strategy, xEnd=None, uEnd=None):
        fp = C.POINTER(C.c_float)
        xe = np.asarray(xEnd, np.float32) if xEnd is not None else None
        ue = np.asarray(uEnd, np.float32) if uEnd is not None else None
        self.L.orc_shift_states(self.h, strategy,
                                xe.ctypes.data_as(fp) if xe is not None else None,
                                ue.ctypes.data_as(fp) if ue is not None else None)

    def shift_controls(self, uEnd=None):
        fp = C.POINTER(C.c_float)
        ue = np.asarray(uEnd, np.float32) if uEnd is not None else None
        self.L.orc_shift_controls(self.h, ue.ctypes.data_as(fp) if ue is not None else None)

    def qpb_solve(self, H, g, lb, ub, yOpt=None, nwsr_max=300):
        fp = C.POINTER(C.c_float)
        H = np.ascontiguousarray(H, np.float32)
        n = H.shape[0]
        g, lb, ub = (np.ascontiguousarray(a, np.float32) for a in (g, lb, ub))
        y0 = np.ascontiguousarray(yOpt, np.float32) if yOpt is not None else None
        x = np.zeros(n, np.float32)
        y = np.zeros(n, np.float32)
        nw = C.c_int(nwsr_max)
        rv = self.L.orc_qpb_solve(n, H.ctypes.data_as(fp), g.ctypes.data_as(fp), lb.ctypes.data_as(fp),
                                  ub.ctypes.data_as(fp), y0.ctypes.data_as(fp) if y0 is not None else None,
                                  C.byref(nw), x.ctypes.data_as(fp), y.ctypes.data_as(fp))
        return rv, x, y, nw.value


_REF_NAMES = {  # our field name -> accessor in oracle/ref_glue.c
    **{k: "ref_" + k for k in VAR_FIELDS},
    **{k: "ref_ws_" + k for k in WS_FIELDS if k not in ("dx", "dual")},
    "dx": "ref_ws_x", "dual": "ref_ws_y", "rk_kkk": "ref_rk_kkk",
}


def ref_available() -> bool:
    return os.path.exists(REF_SO)


class RefAcado(_SolverBase):
    """The compiled reference (N = 50).  Process-global: one instance at a time."""

    def __init__(self):
        if not ref_available():
            raise FileNotFoundError(REF_SO + " (run `make -C oracle ref` where /root/reference exists)")
        L = C.CDLL(REF_SO)
        self.L = L
        self.N = L.ref_N()
        self.v = {}
        sizes = field_sizes(self.N)
        for name, acc in _REF_NAMES.items():
            fn = getattr(L, acc)
            fn.restype = C.POINTER(C.c_float)
            self.v[name] = np.ctypeslib.as_array(fn(), shape=(sizes[name],))
        for f in ("acado_getKKT", "acado_getObjective"):
            getattr(L, f).restype = C.c_float
        L.acado_integrate.argtypes = [C.POINTER(C.c_float), C.c_int]
        L.acado_shiftStates.argtypes = [C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.acado_shiftControls.argtypes = [C.POINTER(C.c_float)]
        L.ref_time_rti.restype = C.c_double
        L.ref_time_rti.argtypes = [C.c_int]
        L.acado_rhs.argtypes = [C.POINTER(C.c_float)] * 2
        L.acado_diffs.argtypes = [C.POINTER(C.c_float)] * 2

    def reset(self): self.L.ref_reset()
    def initialize_solver(self): return self.L.acado_initializeSolver()
    def initialize_nodes_by_forward_simulation(self): self.L.acado_initializeNodesByForwardSimulation()
    def model_simulation(self): return self.L.acado_modelSimulation()
    def evaluate_objective(self): self.L.acado_evaluateObjective()
    def condense_prep(self): self.L.acado_condensePrep()
    def condense_fdb(self): self.L.acado_condenseFdb()
    def solve_qp(self): return self.L.acado_solve()
    def expand(self): self.L.acado_expand()
    def preparation_step(self): return self.L.acado_preparationStep()
    def feedback_step(self): return self.L.acado_feedbackStep()
    def get_kkt(self): return float(self.L.acado_getKKT())
    def get_objective(self): return float(self.L.acado_getObjective())
    def get_nwsr(self): return int(self.L.acado_getNWSR())
    def time_rti(self, iters): return float(self.L.ref_time_rti(iters))

    def integrate(self, eta: np.ndarray, reset: int = 1) -> int:
        assert eta.dtype == np.float32 and eta.size == 23
        return self.L.acado_integrate(eta.ctypes.data_as(C.POINTER(C.c_float)), reset)

    def shift_states(self, strategy, xEnd=None, uEnd=None):
        fp = C.POINTER(C.c_float)
        xe = np.asarray(xEnd, np.float32) if xEnd is not None else None
        ue = np.asarray(uEnd, np.float32) if uEnd is not None else None
        self.L.acado_shiftStates(strategy, xe.ctypes.data_as(fp) if xe is not None else None,
                                 ue.ctypes.data_as(fp) if ue is not None else None)

    def shift_controls(self, uEnd=None):
        fp = C.POINTER(C.c_float)
        ue = np.asarray(uEnd, np.float32) if uEnd is not None else None
        self.L.acado_shiftControls(ue.ctypes.data_as(fp) if ue is not None else None)


INTERMEDIATES_DEFAULT = ("d", "evGx", "evGu", "g", "lb", "ub", "sbar", "QDy")
