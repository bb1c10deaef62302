"""Whole-body NMPC class on the GPU (include/alore_wb.h): B2 + Z1, floating base + 18 joints.
ctypes binding of the C ABI; there is no CPU path (creation fails without a GPU)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

NQ, NV, NX, NU, NJ = 24, 24, 48, 30, 18
DP = C.POINTER(C.c_double)


class WbConfig(C.Structure):
    _fields_ = [("horizon", C.c_int), ("dt", C.c_double), ("device", C.c_int), ("max_problems", C.c_int)]


class WbError(RuntimeError):
    pass


def _bind(L):
    if getattr(L, "_wb_bound", False):
        return
    H = C.c_void_p
    sig = {
        "alore_wb_default_config": (None, [C.POINTER(WbConfig)]),
        "alore_wb_create": (C.c_int, [C.POINTER(WbConfig), C.POINTER(H)]),
        "alore_wb_destroy": (C.c_int, [H]),
        "alore_wb_last_error": (C.c_char_p, [H]),
        "alore_wb_model_info": (C.c_int, [DP, DP, DP, DP]),
        "alore_wb_kernel_info": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "alore_wb_rnea": (C.c_int, [H, C.c_int, DP, DP, DP, DP, C.c_int, DP]),
        "alore_wb_forward_dynamics": (C.c_int, [H, C.c_int, DP, DP, DP, DP, DP]),
        "alore_wb_aba": (C.c_int, [H, C.c_int, DP, DP, DP, DP]),
        "alore_wb_set_weights": (C.c_int, [H, DP, DP, DP]),
        "alore_wb_set_torque_limits": (C.c_int, [H, C.c_int]),
        "alore_wb_set_contact_constraints": (C.c_int, [H, C.c_int, C.c_double]),
        "alore_wb_set_contact_schedule": (C.c_int, [H, C.c_int, C.c_void_p]),
        "alore_wb_set_constraint_mode": (C.c_int, [H, C.c_int, C.c_int]),
        "alore_wb_working_set_info": (C.c_int, [H, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, DP]),
        "alore_wb_set_contact_penalty": (C.c_int, [H, C.c_double]),
        "alore_wb_set_refinement": (C.c_int, [H, C.c_int]),
        "alore_wb_set_contact_rows": (C.c_int, [H, C.c_int]),
        "alore_wb_set_problem": (C.c_int, [H, C.c_int, DP, DP, DP]),
        "alore_wb_set_x0": (C.c_int, [H, C.c_int, DP]),
        "alore_wb_shift_iterate": (C.c_int, [H, C.c_int, C.c_void_p]),
        "alore_wb_get_first_input": (C.c_int, [H, C.c_int, DP]),
        "alore_wb_set_iterate": (C.c_int, [H, C.c_int, DP, DP]),
        "alore_wb_get_iterate": (C.c_int, [H, C.c_int, DP, DP]),
        "alore_wb_linearize": (C.c_int, [H, C.c_int, DP, DP, DP]),
        "alore_wb_rti": (C.c_int, [H, C.c_int, C.c_int, C.c_void_p]),
        "alore_wb_last_step": (C.c_int, [H, C.c_int, DP, DP]),
        "alore_wb_status": (C.c_int, [H, C.c_int, C.POINTER(C.c_int)]),
        "alore_wb_last_times": (C.c_int, [H, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    L._wb_bound = True


def _dp(a):
    return None if a is None else a.ctypes.data_as(DP)


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def model_info() -> dict:
    L = _lib.load(); _bind(L)
    m, lo, hi, ef = np.zeros(19), np.zeros(18), np.zeros(18), np.zeros(18)
    L.alore_wb_model_info(_dp(m), _dp(lo), _dp(hi), _dp(ef))
    return {"masses": m, "lower": lo, "upper": hi, "effort": ef}


class BatchedWholeBody:
    def __init__(self, max_problems: int, horizon: int = 20, dt: float = 0.01, device: int = 0):
        self.L = _lib.load()
        _bind(self.L)
        self.cfg = WbConfig(horizon, dt, device, max_problems)
        self.h = C.c_void_p()
        rc = self.L.alore_wb_create(C.byref(self.cfg), C.byref(self.h))
        if rc != 0:
            raise WbError(f"alore_wb_create failed ({rc}): no GPU or bad configuration; there is no CPU path")
        self.B, self.N, self.dt = max_problems, horizon, dt

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.L.alore_wb_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise WbError(f"alore_wb error {rc}: {self.L.alore_wb_last_error(self.h).decode()}")

    # ---- dynamics -------------------------------------------------------------------------------------------------
    def rnea(self, q, v, a, f=None, gravity=True):
        q, v, a = _f64(q, (-1, NQ)), _f64(v, (-1, NV)), _f64(a, (-1, NV))
        f = None if f is None else _f64(f, (-1, 12))
        tau = np.zeros((q.shape[0], NV))
        self._check(self.L.alore_wb_rnea(self.h, q.shape[0], _dp(q), _dp(v), _dp(a), _dp(f), 1 if gravity else 0, _dp(tau)))
        return tau

    def forward_dynamics(self, q, v, u):
        q, v, u = _f64(q, (-1, NQ)), _f64(v, (-1, NV)), _f64(u, (-1, NU))
        n = q.shape[0]
        M, a = np.zeros((n, NV, NV)), np.zeros((n, NV))
        self._check(self.L.alore_wb_forward_dynamics(self.h, n, _dp(q), _dp(v), _dp(u), _dp(M), _dp(a)))
        return M, a

    def aba(self, q, v, u):
        """forward dynamics by the articulated-body algorithm (no mass matrix)"""
        q, v, u = _f64(q, (-1, NQ)), _f64(v, (-1, NV)), _f64(u, (-1, NU))
        a = np.zeros((q.shape[0], NV))
        self._check(self.L.alore_wb_aba(self.h, q.shape[0], _dp(q), _dp(v), _dp(u), _dp(a)))
        return a

    # ---- OCP ------------------------------------------------------------------------------------------------------
    def set_weights(self, Q, R, QN):
        Q, R, QN = _f64(Q, (NX,)), _f64(R, (NU,)), _f64(QN, (NX,))
        self._check(self.L.alore_wb_set_weights(self.h, _dp(Q), _dp(R), _dp(QN)))

    def set_torque_limits(self, enable: bool):
        self._check(self.L.alore_wb_set_torque_limits(self.h, 1 if enable else 0))

    def set_contact_constraints(self, enable: bool, mu: float = 0.7):
        """friction pyramid + unilateral normal force of the stance feet, zero force of the swing feet, inside the sweep"""
        self._check(self.L.alore_wb_set_contact_constraints(self.h, 1 if enable else 0, float(mu)))

    def set_constraint_mode(self, exact: bool, max_sweeps: int = 8):
        """inequality constraints by one projection per stage inside the sweep (False, the default) or EXACTLY, by a working-set
        iteration around the unconstrained sweep (alore_wb_set_constraint_mode)"""
        self._check(self.L.alore_wb_set_constraint_mode(self.h, 1 if exact else 0, int(max_sweeps)))

    def working_set_info(self, B: int):
        """(sweeps of the last real-time iteration, changes per problem in its last sweep [B], codes [B][N][32], input gradients [B][N][30])"""
        N = self.N
        sweeps = C.c_int(0)
        changed = np.zeros(B, np.int32)
        ws = np.zeros((B, N, 32), np.uint8)
        gu = np.zeros((B, N, NU), np.float64)
        self._check(self.L.alore_wb_working_set_info(self.h, B, C.byref(sweeps), changed.ctypes.data_as(C.POINTER(C.c_int)), ws.ctypes.data_as(C.c_void_p), _dp(gu)))
        return int(sweeps.value), changed, ws, gu

    def set_contact_rows(self, enable: bool = True):
        """hard contact rows J_c(q_k) v_{k+1} = 0 of the stance feet, solved for the foot forces (alore_wb_set_contact_rows)"""
        self._check(self.L.alore_wb_set_contact_rows(self.h, 1 if enable else 0))

    def set_refinement(self, steps=1):
        """steps of iterative refinement of the LQ solution with float64 residuals (alore_wb_set_refinement; True = 1)"""
        self._check(self.L.alore_wb_set_refinement(self.h, int(steps)))

    def set_contact_penalty(self, rho: float):
        """1/2 rho |J_c(q_k) v_k|^2 over the stance feet in the stage cost (alore_wb_set_contact_penalty; 0 = off)"""
        self._check(self.L.alore_wb_set_contact_penalty(self.h, float(rho)))

    def set_contact_schedule(self, stance):
        """stance [B, N, 4] (truthy = foot in contact at that stage), or None for every foot at every stage"""
        if stance is None:
            self._check(self.L.alore_wb_set_contact_schedule(self.h, self.B, None))
            return
        s = np.ascontiguousarray(np.asarray(stance) != 0, dtype=np.uint8)
        assert s.shape[1:] == (self.N, 4) and s.shape[0] <= self.B
        self._check(self.L.alore_wb_set_contact_schedule(self.h, s.shape[0], s.ctypes.data_as(C.c_void_p)))

    def set_problem(self, x0, xref, uref):
        x0, xref, uref = _f64(x0, (-1, NX)), _f64(xref, (-1, self.N + 1, NX)), _f64(uref, (-1, self.N, NU))
        self._n = x0.shape[0]
        self._check(self.L.alore_wb_set_problem(self.h, self._n, _dp(x0), _dp(xref), _dp(uref)))

    def set_x0(self, x0):
        x0 = _f64(x0, (-1, NX))
        self._check(self.L.alore_wb_set_x0(self.h, x0.shape[0], _dp(x0)))

    def shift_iterate(self):
        self._check(self.L.alore_wb_shift_iterate(self.h, self._n, None))

    def first_input(self):
        u0 = np.zeros((self._n, NU))
        self._check(self.L.alore_wb_get_first_input(self.h, self._n, _dp(u0)))
        return u0

    def set_iterate(self, x, u):
        x, u = _f64(x, (-1, self.N + 1, NX)), _f64(u, (-1, self.N, NU))
        self._n = x.shape[0]
        self._check(self.L.alore_wb_set_iterate(self.h, self._n, _dp(x), _dp(u)))

    def get_iterate(self):
        x, u = np.zeros((self._n, self.N + 1, NX)), np.zeros((self._n, self.N, NU))
        self._check(self.L.alore_wb_get_iterate(self.h, self._n, _dp(x), _dp(u)))
        return x, u

    def linearize(self):
        n = self._n
        A, Bm, nx = np.zeros((n, self.N, NX, NX)), np.zeros((n, self.N, NX, NU)), np.zeros((n, self.N, NX))
        self._check(self.L.alore_wb_linearize(self.h, n, _dp(A), _dp(Bm), _dp(nx)))
        return A, Bm, nx

    def rti(self, n_iter: int = 1, stream=None):
        self._check(self.L.alore_wb_rti(self.h, self._n, int(n_iter), stream))

    def last_step(self):
        dx, du = np.zeros((self._n, self.N + 1, NX)), np.zeros((self._n, self.N, NU))
        self._check(self.L.alore_wb_last_step(self.h, self._n, _dp(dx), _dp(du)))
        return dx, du

    def status(self):
        st = np.zeros(self._n, np.int32)
        self._check(self.L.alore_wb_status(self.h, self._n, st.ctypes.data_as(C.POINTER(C.c_int))))
        return st

    def last_times(self):
        a, b = C.c_float(), C.c_float()
        self._check(self.L.alore_wb_last_times(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value
