"""ctypes driver of oracle/backend_oracle.c.  TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py's
cpu_baseline leg); the product package never imports it."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "liboracle_backend.so")
DP = C.POINTER(C.c_double)


class LbfgsParam(C.Structure):
    _fields_ = [("mem_size", C.c_int), ("past", C.c_int), ("max_iterations", C.c_int), ("max_linesearch", C.c_int),
                ("g_epsilon", C.c_double), ("delta", C.c_double), ("min_step", C.c_double), ("max_step", C.c_double),
                ("f_dec_coeff", C.c_double), ("s_curv_coeff", C.c_double), ("cautious_factor", C.c_double),
                ("machine_prec", C.c_double)]


class Config(C.Structure):
    _fields_ = [("max_vel", C.c_double), ("min_vel", C.c_double), ("max_acc", C.c_double), ("max_omega", C.c_double),
                ("max_domega", C.c_double), ("max_cen_acc", C.c_double), ("direct_v_omega", C.c_int),
                ("w_time", C.c_double), ("w_acc", C.c_double), ("w_domega", C.c_double), ("w_collision", C.c_double),
                ("w_moment", C.c_double), ("w_mean_time", C.c_double), ("w_cen_acc", C.c_double),
                ("p_time", C.c_double), ("p_bigpath", C.c_double), ("p_moment", C.c_double), ("p_mean_time", C.c_double),
                ("p_acc", C.c_double), ("p_domega", C.c_double),
                ("energy_w", C.c_double * 2),
                ("smooth_eps", C.c_double), ("safe_dis", C.c_double), ("final_min_safe_dis", C.c_double),
                ("final_check_num", C.c_int), ("safe_replan_max", C.c_int),
                ("mean_lo", C.c_double), ("mean_hi", C.c_double),
                ("sparse_res", C.c_int), ("n_check", C.c_int), ("check_pts", (C.c_double * 2) * 8),
                ("icr_xv", C.c_double), ("standard_diff", C.c_int),
                ("lam0", C.c_double * 2), ("rho0", C.c_double * 2), ("rho_max", C.c_double * 2), ("gamma", C.c_double * 2),
                ("tol", C.c_double),
                ("cut_lam0", C.c_double * 2), ("cut_rho0", C.c_double * 2), ("cut_rho_max", C.c_double * 2),
                ("cut_gamma", C.c_double * 2), ("cut_tol", C.c_double),
                ("path_lbfgs", LbfgsParam), ("shot_path_past", C.c_int), ("shot_path_horizon", C.c_double),
                ("lbfgs", LbfgsParam), ("max_alm_rounds", C.c_int), ("exact_chain", C.c_int)]


class Map(C.Structure):
    _fields_ = [("dist", C.c_void_p), ("nx", C.c_int), ("ny", C.c_int), ("x_lo", C.c_double), ("y_lo", C.c_double),
                ("x_hi", C.c_double), ("y_hi", C.c_double), ("res", C.c_double)]


class Problem(C.Structure):
    _fields_ = [("M", C.c_int), ("inner", C.c_void_p), ("init_T", C.c_double), ("positions", C.c_void_p),
                ("head", (C.c_double * 3) * 2), ("tail", (C.c_double * 3) * 2), ("start_xy", C.c_double * 2),
                ("final_xy", C.c_double * 2), ("if_cut", C.c_int)]


class Result(C.Structure):
    _fields_ = [("inner", C.c_void_p), ("T", C.c_void_p), ("coef", C.c_void_p), ("tail_s", C.c_double),
                ("cost", C.c_double), ("xy_err", C.c_double * 2), ("lbfgs_ret", C.c_int), ("path_ret", C.c_int),
                ("evals", C.c_int), ("alm_rounds", C.c_int), ("attempts", C.c_int), ("collision", C.c_int),
                ("min_dist", C.c_double)]


def build() -> None:
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class EsdfGrid:
    """A double ESDF grid in the reference's layout (dist[ix * ny + iy], cell centres at (i + 0.5) res + lo)."""

    def __init__(self, dist: np.ndarray, x_lo: float, y_lo: float, res: float):
        self.dist = np.ascontiguousarray(dist, dtype=np.float64)
        self.nx, self.ny = self.dist.shape
        self.x_lo, self.y_lo, self.res = float(x_lo), float(y_lo), float(res)
        self.x_hi, self.y_hi = self.x_lo + self.nx * self.res, self.y_lo + self.ny * self.res

    @staticmethod
    def free(half: float = 40.0, res: float = 0.1, value: float = 100.0) -> "EsdfGrid":
        n = int(round(2 * half / res))
        return EsdfGrid(np.full((n, n), value), -half, -half, res)

    @staticmethod
    def from_field(fn, half: float = 40.0, res: float = 0.1) -> "EsdfGrid":
        n = int(round(2 * half / res))
        c = (np.arange(n) + 0.5) * res - half
        X, Y = np.meshgrid(c, c, indexing="ij")
        return EsdfGrid(fn(X, Y), -half, -half, res)

    @staticmethod
    def from_occupancy(grid: np.ndarray, x_lo: float, y_lo: float, res: float, odom=(0.0, 0.0), detection_range: float = 1e3) -> "EsdfGrid":
        """SDFmap::updateESDF2d on a uint8 state grid (0 unknown, 1 free, 2 occupied), starting from DBL_MAX everywhere"""
        if not os.path.exists(SO):
            build()
        L = C.CDLL(SO)
        g = np.ascontiguousarray(grid, dtype=np.uint8)
        dist = np.full(g.shape, np.finfo(np.float64).max)
        L.be_update_esdf2d.restype = C.c_int
        L.be_update_esdf2d.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p]
        rc = L.be_update_esdf2d(g.ctypes.data, g.shape[0], g.shape[1], res, x_lo, y_lo, odom[0], odom[1], detection_range, dist.ctypes.data)
        assert rc == 0
        return EsdfGrid(dist, x_lo, y_lo, res)

    def c_map(self) -> Map:
        return Map(_ptr(self.dist), self.nx, self.ny, self.x_lo, self.y_lo, self.x_hi, self.y_hi, self.res)


def predicted_state(T, coef, step, time, start_time=0.0, start_xytheta=(0.0, 0.0, 0.0), standard_diff=True, xv=0.0):
    """MSPlanner::get_the_predicted_state[_and_path] on a plan (T [M], coef [(6 i + k) * 2 + d])"""
    if not os.path.exists(SO):
        build()
    L = C.CDLL(SO)
    L.be_predicted_state.restype = C.c_int
    L.be_predicted_state.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
    T = np.ascontiguousarray(T, np.float64); coef = np.ascontiguousarray(coef, np.float64)
    xyt = np.array(start_xytheta, np.float64); vaj = np.zeros(3); oaj = np.zeros(3)
    fwd = L.be_predicted_state(T.ctypes.data, coef.ctypes.data, len(T), 1 if standard_diff else 0, xv, step, start_time, time,
                               xyt.ctypes.data, vaj.ctypes.data, oaj.ctypes.data)
    return xyt, vaj, oaj, bool(fwd)


def path_points(T, coef, res, start_xy=(0.0, 0.0), standard_diff=True, xv=0.0):
    """MSPlanner::mincoPointPub on a plan: (points [M (res + 1)][2], yaw [M res])"""
    if not os.path.exists(SO):
        build()
    L = C.CDLL(SO)
    L.be_path_points.restype = C.c_int
    L.be_path_points.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    T = np.ascontiguousarray(T, np.float64); coef = np.ascontiguousarray(coef, np.float64)
    M = len(T)
    xy = np.zeros((M * (res + 1), 2)); yaw = np.zeros(M * res); s0 = np.array(start_xy, np.float64)
    n = L.be_path_points(T.ctypes.data, coef.ctypes.data, M, 1 if standard_diff else 0, xv, int(res), s0.ctypes.data, xy.ctypes.data, yaw.ctypes.data)
    assert n == M * (res + 1)
    return xy, yaw


class BackendOracle:
    def __init__(self):
        if not os.path.exists(SO):
            build()
        L = C.CDLL(SO)
        self.L = L
        L.be_energy.restype = C.c_double
        L.be_esdf.restype = C.c_double
        L.be_esdf.argtypes = [C.POINTER(Map), C.c_double, C.c_double, DP, C.c_int, C.c_double]
        L.be_eval.restype = C.c_double
        L.be_eval.argtypes = [C.POINTER(Config), C.POINTER(Map), C.POINTER(Problem), C.c_int, DP, DP, DP, DP, C.c_double,
                              C.c_double, DP, DP]
        L.be_optimize.argtypes = [C.POINTER(Config), C.POINTER(Map), C.POINTER(Problem), C.c_double, C.c_double,
                                  C.POINTER(Result)]
        L.be_minco_plan.argtypes = [C.POINTER(Config), C.POINTER(Map), C.POINTER(Problem), C.POINTER(Result)]
        L.be_lbfgs_run.argtypes = [C.POINTER(Config), C.POINTER(Map), C.POINTER(Problem), C.c_int, DP, DP, DP, DP,
                                   C.c_double, C.c_double, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), DP]
        self.cfg = Config()
        L.be_default_config(C.byref(self.cfg))

    # ---- plumbing
    @staticmethod
    def problem(ft) -> tuple:
        """FlatTraj -> (Problem, keep-alive arrays)"""
        M = ft.pieces
        inner = np.ascontiguousarray(ft.traj_pts[:, :2], dtype=np.float64).reshape(-1)
        pos = np.concatenate([ft.positions[:, :2].reshape(-1), ft.final_xytheta[:2]]).astype(np.float64)
        p = Problem()
        p.M = M
        p.inner = _ptr(inner)
        p.init_T = ft.init_T
        p.positions = _ptr(pos)
        for d in range(2):
            for k in range(3):
                p.head[d][k] = ft.start_state[d, k]
                p.tail[d][k] = ft.final_state[d, k]
        p.start_xy[0], p.start_xy[1] = ft.start_xytheta[0], ft.start_xytheta[1]
        p.final_xy[0], p.final_xy[1] = ft.final_xytheta[0], ft.final_xytheta[1]
        p.if_cut = int(ft.if_cut)
        return p, (inner, pos)

    @staticmethod
    def x0(ft) -> np.ndarray:
        """initial decision vector of MSPlanner::optimizer (optimizer.cpp:277-286)"""
        M = ft.pieces
        T = ft.init_T
        tau = (np.sqrt(2.0 * T - 1.0) - 1.0) if T > 1.0 else (1.0 - np.sqrt(2.0 / T - 1.0))
        return np.concatenate([ft.traj_pts[:, :2].reshape(-1), [ft.final_state[1, 0]], np.full(M, tau)]).astype(np.float64)

    def spline(self, T, inner, head, tail) -> np.ndarray:
        T = np.ascontiguousarray(T, dtype=np.float64)
        M = len(T)
        inner = np.ascontiguousarray(inner, dtype=np.float64).reshape(-1)
        hd = ((C.c_double * 3) * 2)(*[(C.c_double * 3)(*np.asarray(head)[d]) for d in range(2)])
        tl = ((C.c_double * 3) * 2)(*[(C.c_double * 3)(*np.asarray(tail)[d]) for d in range(2)])
        coef = np.zeros((6 * M, 2))
        self.L.be_spline(C.c_int(M), _ptr(T), _ptr(inner), hd, tl, _ptr(coef))
        return coef

    def energy(self, T, coef, w=(0.33, 1.0)):
        T = np.ascontiguousarray(T, dtype=np.float64)
        M = len(T)
        gdC = np.zeros((6 * M, 2))
        gdT = np.zeros(M)
        ww = (C.c_double * 2)(*w)
        coef = np.ascontiguousarray(coef)
        e = self.L.be_energy(C.c_int(M), _ptr(T), _ptr(coef), ww, _ptr(gdC), _ptr(gdT))
        return e, gdC, gdT

    def adjoint(self, T, coef, gdC, gdT):
        T = np.ascontiguousarray(T, dtype=np.float64)
        M = len(T)
        gp = np.zeros(2 * (M - 1))
        gt = np.zeros(M)
        gtail = (C.c_double * 2)()
        self.L.be_spline_adjoint(C.c_int(M), _ptr(T), _ptr(np.ascontiguousarray(coef)), _ptr(np.ascontiguousarray(gdC)),
                                 _ptr(np.ascontiguousarray(gdT)), _ptr(gp), _ptr(gt), gtail)
        return gp, gt, np.array([gtail[0], gtail[1]])

    def esdf(self, grid: EsdfGrid, x, y, mode=0, mindis=0.0):
        m = grid.c_map()
        g = (C.c_double * 2)(0.0, 0.0)
        d = self.L.be_esdf(C.byref(m), x, y, g, mode, mindis)
        return d, np.array([g[0], g[1]])

    def eval(self, grid: EsdfGrid, ft, stage: int, x, lam=(0.0, 0.0), rho=(1e4, 1e4), safe_dis=None, time_weight=None,
             want_nodes=False):
        p, keep = self.problem(ft)
        m = grid.c_map()
        x = np.ascontiguousarray(x, dtype=np.float64)
        g = np.zeros_like(x)
        err = (C.c_double * 2)()
        M = ft.pieces
        nodes = np.zeros((M * (2 * self.cfg.sparse_res + 1), 2)) if want_nodes else None
        f = self.L.be_eval(C.byref(self.cfg), C.byref(m), C.byref(p), stage, x.ctypes.data_as(DP), g.ctypes.data_as(DP),
                           (C.c_double * 2)(*lam), (C.c_double * 2)(*rho),
                           self.cfg.safe_dis if safe_dis is None else safe_dis,
                           self.cfg.w_time if time_weight is None else time_weight, err,
                           nodes.ctypes.data_as(DP) if want_nodes else None)
        out = {"cost": f, "grad": g, "xy_err": np.array([err[0], err[1]])}
        if want_nodes:
            out["nodes"] = nodes
        return out

    def _result(self, M):
        r = Result()
        bufs = (np.zeros(2 * max(M - 1, 1)), np.zeros(M), np.zeros((6 * M, 2)))
        r.inner, r.T, r.coef = _ptr(bufs[0]), _ptr(bufs[1]), _ptr(bufs[2])
        return r, bufs

    @staticmethod
    def _unpack(r, bufs, M):
        return {"inner": bufs[0][:2 * (M - 1)].reshape(-1, 2).copy(), "T": bufs[1].copy(), "coef": bufs[2].copy(),
                "tail_s": r.tail_s, "cost": r.cost, "xy_err": np.array([r.xy_err[0], r.xy_err[1]]),
                "lbfgs_ret": r.lbfgs_ret, "path_ret": r.path_ret, "evals": r.evals, "alm_rounds": r.alm_rounds,
                "attempts": r.attempts, "collision": r.collision, "min_dist": r.min_dist}

    def optimize(self, grid: EsdfGrid, ft, safe_dis=None, time_weight=None):
        p, keep = self.problem(ft)
        m = grid.c_map()
        r, bufs = self._result(ft.pieces)
        self.L.be_optimize(C.byref(self.cfg), C.byref(m), C.byref(p),
                           C.c_double(self.cfg.safe_dis if safe_dis is None else safe_dis),
                           C.c_double(self.cfg.w_time if time_weight is None else time_weight), C.byref(r))
        return self._unpack(r, bufs, ft.pieces)

    def minco_plan(self, grid: EsdfGrid, ft):
        p, keep = self.problem(ft)
        m = grid.c_map()
        r, bufs = self._result(ft.pieces)
        rc = self.L.be_minco_plan(C.byref(self.cfg), C.byref(m), C.byref(p), C.byref(r))
        out = self._unpack(r, bufs, ft.pieces)
        out["ok"] = rc == 0
        return out

    def lbfgs_run(self, grid: EsdfGrid, ft, stage: int, x, lam=(0.0, 0.0), rho=(1e4, 1e4), max_iter=0, safe_dis=None,
                  time_weight=None):
        p, keep = self.problem(ft)
        m = grid.c_map()
        x = np.array(x, dtype=np.float64)
        cost = C.c_double()
        it, ev = C.c_int(), C.c_int()
        err = (C.c_double * 2)()
        rc = self.L.be_lbfgs_run(C.byref(self.cfg), C.byref(m), C.byref(p), stage, x.ctypes.data_as(DP), C.byref(cost),
                                 (C.c_double * 2)(*lam), (C.c_double * 2)(*rho),
                                 self.cfg.safe_dis if safe_dis is None else safe_dis,
                                 self.cfg.w_time if time_weight is None else time_weight, max_iter, C.byref(it), C.byref(ev), err)
        return {"ret": rc, "x": x, "cost": cost.value, "iters": it.value, "evals": ev.value,
                "xy_err": np.array([err[0], err[1]])}
