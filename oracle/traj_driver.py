"""ctypes driver of oracle/traj_oracle.hpp (oracle/liboracle_traj.so): the float64 host restatement of the
reference's controller-side reference sampling (TrajAnal / getRefPoints / smooth_yaw).
TEST INFRASTRUCTURE ONLY -- the product samples references on the GPU (csrc/ref_sampler.hip)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "liboracle_traj.so")
DP = C.POINTER(C.c_double)
_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO):
        subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    L = C.CDLL(SO)
    L.orc_sampler_create.restype = C.c_void_p
    L.orc_sampler_create.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double]
    L.orc_sampler_destroy.argtypes = [C.c_void_p]
    L.orc_sampler_traj.argtypes = [C.c_void_p, C.c_double, C.c_int] + [DP] * 6
    L.orc_sampler_odom.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double]
    L.orc_sampler_icr.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double]
    L.orc_sampler_refs.argtypes = [C.c_void_p, C.c_double, C.c_int, DP, DP, C.POINTER(C.c_int)]
    L.orc_sampler_duration.restype = C.c_double
    L.orc_sampler_duration.argtypes = [C.c_void_p]
    L.orc_sampler_state.argtypes = [C.c_void_p, C.c_double, DP, DP, DP]
    L.orc_sampler_flat.argtypes = [C.c_void_p, C.c_double, C.c_int, DP]
    L.orc_sampler_coefficients.argtypes = [C.c_void_p, DP, DP, C.c_int]
    L.orc_sampler_sequence.argtypes = [C.c_void_p, DP, C.c_int]
    L.orc_sampler_at_goal.argtypes = [C.c_void_p]
    L.orc_normlize_theta.argtypes = [DP]
    _lib = L
    return L


def _dp(a):
    return a.ctypes.data_as(DP)


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


class RefSampler:
    """One robot's TrajAnal + getRefPoints + smooth_yaw on the host."""

    def __init__(self, N, dt=0.01, state_seq_res=0.1, integral_res_int=4):
        self.L = load()
        self.N = N
        self.h = self.L.orc_sampler_create(N, dt, state_seq_res, integral_res_int)

    def __del__(self):
        try:
            if self.h:
                self.L.orc_sampler_destroy(self.h)
        except Exception:
            pass

    def traj(self, m):
        inner = m.innerpoints if m.innerpoints.size else np.zeros((1, 2))
        rc = self.L.orc_sampler_traj(self.h, m.traj_start_time, m.t_pts.size, _dp(inner), _dp(m.t_pts), _dp(m.init_pva),
                                     _dp(m.tail_pva), _dp(m.start_position), _dp(m.ICR))
        if rc != 0:
            raise ValueError("bad Polynome")

    def odom(self, x, y, yaw): self.L.orc_sampler_odom(self.h, x, y, yaw)
    def icr(self, yr, yl, xv): self.L.orc_sampler_icr(self.h, yr, yl, xv)
    def duration(self): return self.L.orc_sampler_duration(self.h)

    @property
    def at_goal(self): return bool(self.L.orc_sampler_at_goal(self.h))

    def refs(self, now, smooth=True):
        rs = np.zeros((self.N + 1, 3)); ri = np.zeros((self.N + 1, 2)); g = C.c_int(0)
        if self.L.orc_sampler_refs(self.h, now, 1 if smooth else 0, _dp(rs), _dp(ri), C.byref(g)) != 0:
            raise RuntimeError("getRefPoints failed")
        return rs, ri, bool(g.value)

    def state(self, t):
        p = np.zeros(3); v = np.zeros(2); a = np.zeros(2)
        if self.L.orc_sampler_state(self.h, t, _dp(p), _dp(v), _dp(a)) != 0:
            raise IndexError("t outside the state sequence")
        return p, v, a

    def flat(self, t, orders=(0, 1, 2)):
        out = []
        for o in orders:
            v = np.zeros(2)
            self.L.orc_sampler_flat(self.h, t, o, _dp(v))
            out.append(v)
        return tuple(out)

    def coefficients(self, max_pieces=64):
        dur = np.zeros(max_pieces); coef = np.zeros((max_pieces, 2, 6))
        n = self.L.orc_sampler_coefficients(self.h, _dp(dur), _dp(coef), max_pieces)
        return dur[:n].copy(), coef[:n].copy()

    def sequence(self):
        buf = np.zeros((4096, 4))
        n = self.L.orc_sampler_sequence(self.h, _dp(buf), 4096)
        return buf[:n].copy()


def normlize_theta(th: float) -> float:
    v = C.c_double(th)
    load().orc_normlize_theta(C.byref(v))
    return v.value
