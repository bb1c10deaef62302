"""ctypes binding of the C++ host layer (alore_legged_manipulator_amd/host/*.hpp -> libalore_nmpc_host.so,
C entry points in include/alore_nmpc_host.h): B instances of the reference's ROS-free `nmpc` node tick
(mpc.cpp CmdCallback), everything numeric on the GPU."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libalore_nmpc_host.so")
DP = C.POINTER(C.c_double)
_lib = None


class MpcParams(C.Structure):
    """alore_host_mpc_params: the node's private parameters (mpc.cpp:11-20, 32-33, 69-70)"""
    _fields_ = [("max_omega", C.c_double), ("max_domega", C.c_double), ("max_vel", C.c_double), ("min_vel", C.c_double),
                ("max_acc", C.c_double), ("cmd_timer_rate", C.c_double), ("max_mpc_time", C.c_double), ("if_mpc", C.c_int),
                ("delay_num", C.c_int), ("state_seq_res", C.c_double), ("Integral_appr_resInt", C.c_int),
                ("matrix_q", C.c_double * 3), ("matrix_r", C.c_double * 2)]


class Command(C.Structure):
    _fields_ = [("right_wheel_ome", C.c_double), ("left_wheel_ome", C.c_double), ("v", C.c_double), ("omega", C.c_double),
                ("a", C.c_double), ("alpha", C.c_double), ("wheel_published", C.c_int), ("state_published", C.c_int)]


def load():
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (one HIP runtime per process, see _lib.py)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: run __graft_entry__.build()")
    L = C.CDLL(LIB_PATH)
    L.alore_host_default_params.argtypes = [C.POINTER(MpcParams)]
    L.alore_host_default_params.restype = None
    L.alore_host_controller_create.restype = C.c_void_p
    L.alore_host_controller_create.argtypes = [C.c_int, C.c_int, C.c_double, C.POINTER(MpcParams), C.c_int, C.c_int, C.c_int]
    L.alore_host_controller_destroy.argtypes = [C.c_void_p]
    L.alore_host_controller_odom.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double]
    L.alore_host_controller_icr.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double]
    L.alore_host_controller_traj.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int] + [DP] * 6
    L.alore_host_controller_emergency_stop.argtypes = [C.c_void_p, C.c_int]
    L.alore_host_controller_robot_state.argtypes = [C.c_void_p, C.c_int] + [C.POINTER(C.c_int)] * 3
    L.alore_host_controller_tick.argtypes = [C.c_void_p, C.c_double, C.POINTER(Command)]
    L.alore_host_controller_references.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.alore_host_controller_prediction.argtypes = [C.c_void_p, C.c_int, DP, DP, C.POINTER(C.c_int)]
    _lib = L
    return L


def _dp(a):
    return a.ctypes.data_as(DP)


def default_params(**kw) -> MpcParams:
    p = MpcParams()
    load().alore_host_default_params(C.byref(p))
    for k, v in kw.items():
        if k in ("matrix_q", "matrix_r"):
            for i, x in enumerate(v):
                getattr(p, k)[i] = x
        else:
            setattr(p, k, v)
    return p


class Polynome:
    """ROS-free carstatemsgs/Polynome: flat-space (theta, s) minimum-jerk trajectory."""

    def __init__(self, innerpoints, t_pts, init_pva, tail_pva, start_position, ICR, traj_start_time=0.0):
        self.innerpoints = np.ascontiguousarray(innerpoints, np.float64).reshape(-1, 2)
        self.t_pts = np.ascontiguousarray(t_pts, np.float64)
        self.init_pva = np.ascontiguousarray(init_pva, np.float64).reshape(6)   # p0 p1 v0 v1 a0 a1
        self.tail_pva = np.ascontiguousarray(tail_pva, np.float64).reshape(6)
        self.start_position = np.ascontiguousarray(start_position, np.float64).reshape(3)
        self.ICR = np.ascontiguousarray(ICR, np.float64).reshape(3)              # (yr, yl, xv) as sent
        self.traj_start_time = float(traj_start_time)
        assert self.innerpoints.shape[0] == self.t_pts.size - 1


class _Robot:
    """The subscriber side of robot b (~odom, ~EKF_ICR, ~traj, /planner/emergency_stop)."""

    def __init__(self, ctl, b):
        self._c, self._b = ctl, b

    def odom(self, x, y, yaw): self._c._chk(self._c.L.alore_host_controller_odom(self._c.h, self._b, x, y, yaw))
    def icr(self, yr, yl, xv): self._c._chk(self._c.L.alore_host_controller_icr(self._c.h, self._b, yr, yl, xv))

    def traj(self, m):
        inner = m.innerpoints if m.innerpoints.size else np.zeros((1, 2))
        self._c._chk(self._c.L.alore_host_controller_traj(self._c.h, self._b, m.traj_start_time, m.t_pts.size, _dp(inner), _dp(m.t_pts),
                                                          _dp(m.init_pva), _dp(m.tail_pva), _dp(m.start_position), _dp(m.ICR)))

    def emergency_stop(self): self._c._chk(self._c.L.alore_host_controller_emergency_stop(self._c.h, self._b))

    def _state(self):
        g, r, o = C.c_int(), C.c_int(), C.c_int()
        self._c._chk(self._c.L.alore_host_controller_robot_state(self._c.h, self._b, C.byref(g), C.byref(r), C.byref(o)))
        return bool(g.value), bool(r.value), bool(o.value)

    @property
    def at_goal(self): return self._state()[0]
    @property
    def receive_traj(self): return self._state()[1]


class BatchedMpcController:
    """B copies of the reference's nmpc node tick (mpc.cpp CmdCallback), GPU-backed."""

    def __init__(self, B, N=20, dt=0.01, matrix_q=(10.0, 10.0, 0.5), matrix_r=(0.1, 0.1), delay_num=1, state_seq_res=0.1,
                 integral_res_int=4, device=0, max_pieces=64, max_checkpoints=1024, params: MpcParams | None = None):
        self.L = load()
        self.B, self.N = B, N
        self.params = params or default_params(matrix_q=matrix_q, matrix_r=matrix_r, delay_num=delay_num,
                                               state_seq_res=state_seq_res, Integral_appr_resInt=integral_res_int)
        self.h = self.L.alore_host_controller_create(B, N, dt, C.byref(self.params), device, max_pieces, max_checkpoints)
        if not self.h:
            raise RuntimeError("controller creation failed (no GPU?)")
        self.robots = [_Robot(self, b) for b in range(B)]

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError("host layer call failed")

    def __del__(self):
        try:
            if self.h:
                self.L.alore_host_controller_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def tick_full(self, now):
        """Everything the B nodes publish on this tick: list of Command."""
        cmd = (Command * self.B)()
        self._chk(self.L.alore_host_controller_tick(self.h, now, cmd))
        return cmd

    def tick(self, now):
        """(B, 2) wheel-speed commands (right, left); zero where nothing was published."""
        cmd = self.tick_full(now)
        return np.array([[c.right_wheel_ome, c.left_wheel_ome] if c.wheel_published else [0.0, 0.0] for c in cmd])

    def references(self):
        """(y, yN, od, x0) as the solver saw them on the last tick (read back from the device)."""
        y = np.zeros((self.B, self.N, 5), np.float32); yN = np.zeros((self.B, 3), np.float32)
        od = np.zeros((self.B, self.N + 1, 3), np.float32); x0 = np.zeros((self.B, 3), np.float32)
        self._chk(self.L.alore_host_controller_references(self.h, y.ctypes.data, yN.ctypes.data, od.ctypes.data, x0.ctypes.data))
        return y, yN, od, x0

    def prediction(self, b):
        s = np.zeros((self.N + 1, 3)); u = np.zeros((self.N, 2)); st = C.c_int(0)
        self._chk(self.L.alore_host_controller_prediction(self.h, b, _dp(s), _dp(u), C.byref(st)))
        return s, u, st.value
