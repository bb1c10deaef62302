"""Host-side mirror of the reference's `mpc` node numerics for a batch of robots (include/alore_ltv_mpc.h):
``BatchedLtvMpc.get_cmd`` = MpcController::getCmd (mpc_controller/src/mpc.cpp:569-614) for B robots on the GPU."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

DP = C.POINTER(C.c_double)
IP = C.POINTER(C.c_int)


class LtvConfig(C.Structure):
    _fields_ = [("dt", C.c_double), ("predict_steps", C.c_int), ("delay_num", C.c_int), ("matrix_q", C.c_double * 4),
                ("matrix_r", C.c_double * 2), ("matrix_rd", C.c_double * 2), ("max_vel", C.c_double), ("min_vel", C.c_double),
                ("max_omega", C.c_double), ("max_acc", C.c_double), ("max_domega", C.c_double), ("max_sweeps", C.c_int)]


def _bind(L):
    if getattr(L, "_ltv_bound", False):
        return
    L.alore_ltv_default_config.argtypes = [C.POINTER(LtvConfig)]
    L.alore_ltv_default_config.restype = None
    L.alore_ltv_create.argtypes = [C.POINTER(LtvConfig), C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.alore_ltv_destroy.argtypes = [C.c_void_p]
    L.alore_ltv_last_error.argtypes = [C.c_void_p]
    L.alore_ltv_last_error.restype = C.c_char_p
    L.alore_ltv_set_refs.argtypes = [C.c_void_p, C.c_int, DP, DP, C.c_void_p]
    L.alore_ltv_refs_from_store.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, DP, IP, C.c_void_p]
    L.alore_ltv_get_cmd.argtypes = [C.c_void_p, C.c_int, DP, C.c_int, C.c_int, C.c_void_p]
    L.alore_ltv_results.argtypes = [C.c_void_p, C.c_int, DP, DP, IP, IP, C.c_void_p]
    L.alore_ltv_commands.argtypes = [C.c_void_p, C.c_int, DP, IP, C.c_void_p]
    L.alore_ltv_tick.argtypes = [C.c_void_p, C.c_int, DP, C.c_int, C.c_int, DP, IP, C.c_void_p]
    L.alore_ltv_set_state.argtypes = [C.c_void_p, C.c_int, DP, DP, C.c_void_p]
    L._ltv_bound = True


def default_config(**kw) -> LtvConfig:
    L = _lib.load()
    _bind(L)
    c = LtvConfig()
    L.alore_ltv_default_config(C.byref(c))
    for k, v in kw.items():
        if k.startswith("matrix_"):
            for i, x in enumerate(v):
                getattr(c, k)[i] = x
        else:
            setattr(c, k, v)
    return c


def _dp(a):
    return a.ctypes.data_as(DP) if a is not None else None


class LtvError(RuntimeError):
    pass


class BatchedLtvMpc:
    def __init__(self, max_robots: int, config: LtvConfig | None = None, device: int = 0):
        self.L = _lib.load()
        _bind(self.L)
        self.cfg = config or default_config()
        self.h = C.c_void_p()
        rc = self.L.alore_ltv_create(C.byref(self.cfg), device, max_robots, C.byref(self.h))
        if rc != 0:
            raise LtvError(f"alore_ltv_create failed ({rc}): no GPU or bad configuration; there is no CPU path")
        self.B, self.T, self.d = max_robots, self.cfg.predict_steps, self.cfg.delay_num

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.L.alore_ltv_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise LtvError(f"alore_ltv error {rc}: {self.L.alore_ltv_last_error(self.h).decode()}")

    def set_refs(self, xref, dref):
        """xref (B, T, 3): x, y, theta;  dref (B, T, 2): v, omega"""
        xref = np.ascontiguousarray(xref, np.float64); dref = np.ascontiguousarray(dref, np.float64)
        self._n = xref.shape[0]
        self._check(self.L.alore_ltv_set_refs(self.h, self._n, _dp(xref), _dp(dref), None))

    def refs_from_store(self, nmpc_engine, now, est):
        est = np.ascontiguousarray(est, np.float64)
        self._n = est.shape[0]
        goal = np.zeros(self._n, np.int32)
        self._check(self.L.alore_ltv_refs_from_store(self.h, nmpc_engine.h, self._n, float(now), _dp(est), goal.ctypes.data_as(IP), None))
        return goal.astype(bool)

    def set_state(self, output=None, buff=None):
        o = None if output is None else np.ascontiguousarray(output, np.float64)
        b = None if buff is None else np.ascontiguousarray(buff, np.float64)
        n = (o if o is not None else b).shape[0]
        self._check(self.L.alore_ltv_set_state(self.h, n, _dp(o), _dp(b), None))

    def tick(self, now_state, n_relin=5, reset=False):
        """getCmd for all robots, returning only what a control tick publishes: cmd (n, 2) and the status"""
        now_state = np.ascontiguousarray(now_state, np.float64)
        n = now_state.shape[0]
        cmd = np.zeros((n, 2)); st = np.zeros(n, np.int32)
        self._check(self.L.alore_ltv_tick(self.h, n, _dp(now_state), int(n_relin), 1 if reset else 0, _dp(cmd), st.ctypes.data_as(IP), None))
        return cmd, st

    def get_cmd(self, now_state, n_relin=5, reset=False):
        now_state = np.ascontiguousarray(now_state, np.float64)
        n = now_state.shape[0]
        self._check(self.L.alore_ltv_get_cmd(self.h, n, _dp(now_state), int(n_relin), 1 if reset else 0, None))
        out = np.zeros((n, self.T, 2)); xopt = np.zeros((n, self.T + 1, 3))
        sw = np.zeros(n, np.int32); st = np.zeros(n, np.int32)
        self._check(self.L.alore_ltv_results(self.h, n, _dp(out), _dp(xopt), sw.ctypes.data_as(IP), st.ctypes.data_as(IP), None))
        return {"output": out, "cmd": out[:, self.d].copy(), "xopt": xopt, "sweeps": sw, "status": st}
