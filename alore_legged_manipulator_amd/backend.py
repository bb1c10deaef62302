"""Host-side mirror of the reference's back_end entry point for a batch of problems.

``BatchedMSPlanner.minco_plan(flat_trajs)`` is ``MSPlanner::minco_plan(const FlatTrajData&)``
(planning_ddr_opt/back_end/src/optimizer.cpp:169-220) for many FlatTrajData at once, one wavefront per
problem on the GPU, through the C ABI of include/alore_backend.h.  Same parameter names as the reference's
yaml files (``config`` is alore_backend_config); results are what plan_manager reads back to build the
``Polynome`` message (plan_manager.hpp:784-831): ``polynomes()`` returns them in that wire format.
There is no CPU path: the constructor raises without the HIP library or without a GPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

DP = C.POINTER(C.c_double)


class LbfgsParam(C.Structure):
    _fields_ = [("mem_size", C.c_int), ("past", C.c_int), ("max_iterations", C.c_int), ("max_linesearch", C.c_int),
                ("g_epsilon", C.c_double), ("delta", C.c_double), ("min_step", C.c_double), ("max_step", C.c_double),
                ("f_dec_coeff", C.c_double), ("s_curv_coeff", C.c_double), ("cautious_factor", C.c_double),
                ("machine_prec", C.c_double)]


class BackendConfig(C.Structure):
    """alore_backend_config (include/alore_backend.h)"""
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
                ("lbfgs", LbfgsParam), ("max_alm_rounds", C.c_int)]


class FlatTrajC(C.Structure):
    _fields_ = [("n_pieces", C.c_int), ("traj_pts", C.c_void_p), ("init_T", C.c_double), ("positions", C.c_void_p),
                ("start_state", (C.c_double * 3) * 2), ("final_state", (C.c_double * 3) * 2),
                ("start_xytheta", C.c_double * 3), ("final_xytheta", C.c_double * 3), ("if_cut", C.c_int)]


class StatusC(C.Structure):
    _fields_ = [("ok", C.c_int), ("attempts", C.c_int), ("alm_rounds", C.c_int), ("evals", C.c_int), ("lbfgs_ret", C.c_int),
                ("path_ret", C.c_int), ("collision", C.c_int), ("n_pieces", C.c_int), ("cost", C.c_double),
                ("xy_err", C.c_double * 2), ("min_dist", C.c_double), ("tail_s", C.c_double)]


class DeviceView(C.Structure):
    _fields_ = [("max_pieces", C.c_int), ("n_pieces", C.c_void_p), ("inner", C.c_void_p), ("T", C.c_void_p),
                ("coef", C.c_void_p), ("head", C.c_void_p), ("tail", C.c_void_p), ("start_xytheta", C.c_void_p),
                ("ok", C.c_void_p)]


def _bind(L):
    if getattr(L, "_backend_bound", False):
        return
    L.alore_backend_default_config.argtypes = [C.POINTER(BackendConfig)]
    L.alore_backend_default_config.restype = None
    L.alore_backend_create.argtypes = [C.POINTER(BackendConfig), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.alore_backend_destroy.argtypes = [C.c_void_p]
    L.alore_backend_last_error.argtypes = [C.c_void_p]
    L.alore_backend_last_error.restype = C.c_char_p
    L.alore_backend_set_map.argtypes = [C.c_void_p, DP, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double]
    L.alore_backend_build_esdf.argtypes = [C.c_void_p, C.POINTER(C.c_ubyte), C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                           C.c_double, C.c_double, C.c_double, DP]
    L.alore_backend_predicted_state.argtypes = [C.c_void_p, C.c_int, C.c_double, DP, DP, DP, DP, DP, DP, C.POINTER(C.c_int)]
    L.alore_backend_path_points.argtypes = [C.c_void_p, C.c_int, C.c_int, DP, DP, C.POINTER(C.c_int)]
    L.alore_backend_set_problems.argtypes = [C.c_void_p, C.c_int, C.POINTER(FlatTrajC), C.c_void_p]
    L.alore_backend_plan.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.alore_backend_results.argtypes = [C.c_void_p, C.c_int, C.POINTER(StatusC), DP, DP, DP, C.c_void_p]
    L.alore_backend_device_results.argtypes = [C.c_void_p, C.POINTER(DeviceView)]
    L.alore_backend_eval.argtypes = [C.c_void_p, C.c_int, C.c_int, DP, DP, DP, C.c_double, C.c_double, DP, DP, DP, C.c_void_p]
    L.alore_backend_lbfgs.argtypes = [C.c_void_p, C.c_int, C.c_int, DP, DP, DP, C.c_double, C.c_double, C.c_int, DP,
                                      C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), DP, C.c_void_p]
    L.alore_backend_last_plan_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    L._backend_bound = True


class BackendError(RuntimeError):
    pass


def default_config() -> BackendConfig:
    L = _lib.load()
    _bind(L)
    c = BackendConfig()
    L.alore_backend_default_config(C.byref(c))
    return c


def _dp(a):
    return a.ctypes.data_as(DP) if a is not None else None


class BatchedMSPlanner:
    """MSPlanner for a batch.  ``max_pieces``: the longest trajectory (pieces) the handle accepts (<= 32)."""

    def __init__(self, max_problems: int, max_pieces: int = 16, config: BackendConfig | None = None, device: int = 0):
        self.L = _lib.load()
        _bind(self.L)
        self.cfg = config or default_config()
        self.h = C.c_void_p()
        rc = self.L.alore_backend_create(C.byref(self.cfg), device, max_pieces, max_problems, C.byref(self.h))
        if rc != 0:
            raise BackendError(f"alore_backend_create failed ({rc}): no GPU, or unsupported configuration; there is no CPU path")
        self.P = 16 if max_pieces <= 16 else 32
        self.B = max_problems
        self.count = 0
        self._keep = None

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.L.alore_backend_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise BackendError(f"alore_backend error {rc}: {self.L.alore_backend_last_error(self.h).decode()}")

    # plan_env::SDFmap: dist[ix, iy] double grid, cell centres at (i + 0.5) res + lo
    def set_map(self, dist: np.ndarray, x_lo: float, y_lo: float, res: float):
        dist = np.ascontiguousarray(dist, dtype=np.float64)
        self._check(self.L.alore_backend_set_map(self.h, _dp(dist), dist.shape[0], dist.shape[1], x_lo, y_lo, res))

    def build_esdf(self, grid: np.ndarray, x_lo: float, y_lo: float, res: float, odom=(0.0, 0.0), detection_range: float = 1e3) -> np.ndarray:
        """SDFmap::updateESDF2d on the device from a uint8 state grid (0 unknown, 1 free, 2 occupied); the result becomes
        the planner's map and is returned."""
        g = np.ascontiguousarray(grid, dtype=np.uint8)
        dist = np.zeros(g.shape)
        self._check(self.L.alore_backend_build_esdf(self.h, g.ctypes.data_as(C.POINTER(C.c_ubyte)), g.shape[0], g.shape[1], x_lo, y_lo, res,
                                                    float(odom[0]), float(odom[1]), float(detection_range), _dp(dist)))
        return dist

    def predicted_state(self, times, resolution: float = 0.01, start_times=None, start_xytheta=None):
        """MSPlanner::get_the_predicted_state[_and_path] for the plans of the last launch"""
        t = np.ascontiguousarray(times, np.float64)
        n = t.shape[0]
        st = None if start_times is None else np.ascontiguousarray(start_times, np.float64)
        sx = None if start_xytheta is None else np.ascontiguousarray(start_xytheta, np.float64)
        xyt, vaj, oaj, fwd = np.zeros((n, 3)), np.zeros((n, 3)), np.zeros((n, 3)), np.zeros(n, np.int32)
        self._check(self.L.alore_backend_predicted_state(self.h, n, float(resolution), _dp(st), _dp(t), _dp(sx), _dp(xyt), _dp(vaj), _dp(oaj),
                                                         fwd.ctypes.data_as(C.POINTER(C.c_int))))
        return {"xytheta": xyt, "vaj": vaj, "oaj": oaj, "forward": fwd.astype(bool)}

    def path_points(self, panels_per_piece: int = 3, count: int | None = None):
        """MSPlanner::mincoPointPub for the plans of the last launch: list of (points [n][2], yaw [n_pieces panels]) per plan"""
        n = int(count or self.count)
        per = self.P * (panels_per_piece + 1)
        xy = np.zeros((n, per, 2)); yaw = np.zeros((n, self.P * panels_per_piece)); npts = np.zeros(n, np.int32)
        self._check(self.L.alore_backend_path_points(self.h, n, int(panels_per_piece), _dp(xy), _dp(yaw), npts.ctypes.data_as(C.POINTER(C.c_int))))
        return [(xy[b, :npts[b]].copy(), yaw[b, :(npts[b] // (panels_per_piece + 1)) * panels_per_piece].copy()) for b in range(n)]

    def set_free_map(self, half: float = 40.0, res: float = 0.1, value: float = 100.0):
        n = int(round(2 * half / res))
        self.set_map(np.full((n, n), value), -half, -half, res)

    def set_problems(self, flat_trajs):
        n = len(flat_trajs)
        arr = (FlatTrajC * n)()
        keep = []
        for b, ft in enumerate(flat_trajs):
            pts = np.ascontiguousarray(ft.traj_pts, dtype=np.float64).reshape(-1, 3)
            pos = np.ascontiguousarray(ft.positions, dtype=np.float64).reshape(-1, 3)
            keep += [pts, pos]
            a = arr[b]
            a.n_pieces = len(pts) + 1
            a.traj_pts = pts.ctypes.data
            a.positions = pos.ctypes.data
            a.init_T = ft.init_T
            for d in range(2):
                for k in range(3):
                    a.start_state[d][k] = ft.start_state[d, k]
                    a.final_state[d][k] = ft.final_state[d, k]
            for k in range(3):
                a.start_xytheta[k] = ft.start_xytheta[k]
                a.final_xytheta[k] = ft.final_xytheta[k]
            a.if_cut = int(ft.if_cut)
        self._keep = keep
        self._check(self.L.alore_backend_set_problems(self.h, n, arr, None))
        self.count = n
        self._pieces = np.array([a.n_pieces for a in arr])

    def plan(self, count: int | None = None):
        self._check(self.L.alore_backend_plan(self.h, count or self.count, None))

    def last_plan_ms(self) -> float:
        ms = C.c_float()
        self._check(self.L.alore_backend_last_plan_ms(self.h, C.byref(ms)))
        return ms.value

    def results(self, count: int | None = None) -> dict:
        n = count or self.count
        st = (StatusC * n)()
        inner = np.zeros((n, self.P - 1, 2))
        T = np.zeros((n, self.P))
        coef = np.zeros((n, self.P * 6, 2))
        self._check(self.L.alore_backend_results(self.h, n, st, _dp(inner), _dp(T), _dp(coef), None))
        out = {k: np.array([getattr(s, k) for s in st]) for k in
               ("ok", "attempts", "alm_rounds", "evals", "lbfgs_ret", "path_ret", "collision", "n_pieces", "cost", "min_dist", "tail_s")}
        out["xy_err"] = np.array([[s.xy_err[0], s.xy_err[1]] for s in st])
        out["inner"], out["T"], out["coef"] = inner, T, coef
        return out

    def minco_plan(self, flat_trajs) -> dict:
        self.set_problems(flat_trajs)
        self.plan()
        return self.results()

    def device_view(self) -> DeviceView:
        v = DeviceView()
        self._check(self.L.alore_backend_device_results(self.h, C.byref(v)))
        return v

    # ---- pieces of the optimiser for parity tests
    def pack_x(self, xs) -> np.ndarray:
        out = np.zeros((len(xs), 3 * self.P))
        for b, x in enumerate(xs):
            out[b, :len(x)] = x
        return out

    def eval(self, stage: int, xs, lam=None, rho=None, safe_dis=None, time_weight=None):
        n = len(xs)
        X = self.pack_x(xs)
        cost = np.zeros(n); grad = np.zeros_like(X); err = np.zeros((n, 2))
        lam = None if lam is None else np.ascontiguousarray(np.broadcast_to(lam, (n, 2)), dtype=np.float64)
        rho = None if rho is None else np.ascontiguousarray(np.broadcast_to(rho, (n, 2)), dtype=np.float64)
        self._check(self.L.alore_backend_eval(self.h, n, stage, _dp(X), _dp(lam), _dp(rho),
                                              self.cfg.safe_dis if safe_dis is None else safe_dis,
                                              self.cfg.w_time if time_weight is None else time_weight, _dp(cost), _dp(grad),
                                              _dp(err), None))
        return {"cost": cost, "grad": [grad[b, :len(xs[b])] for b in range(n)], "xy_err": err}

    def lbfgs(self, stage: int, xs, lam=None, rho=None, max_iter=0, safe_dis=None, time_weight=None):
        n = len(xs)
        X = self.pack_x(xs)
        cost = np.zeros(n); err = np.zeros((n, 2))
        ret = (C.c_int * n)(); it = (C.c_int * n)(); ev = (C.c_int * n)()
        lam = None if lam is None else np.ascontiguousarray(np.broadcast_to(lam, (n, 2)), dtype=np.float64)
        rho = None if rho is None else np.ascontiguousarray(np.broadcast_to(rho, (n, 2)), dtype=np.float64)
        self._check(self.L.alore_backend_lbfgs(self.h, n, stage, _dp(X), _dp(lam), _dp(rho),
                                               self.cfg.safe_dis if safe_dis is None else safe_dis,
                                               self.cfg.w_time if time_weight is None else time_weight, max_iter, _dp(cost),
                                               ret, it, ev, _dp(err), None))
        return {"x": [X[b, :len(xs[b])] for b in range(n)], "cost": cost, "ret": np.array(ret[:]), "iters": np.array(it[:]),
                "evals": np.array(ev[:]), "xy_err": err}

    # ---- the message plan_manager publishes from the result (plan_manager.hpp:784-831)
    def polynomes(self, flat_trajs, res: dict, traj_start_time: float = 0.0) -> list:
        """One dict per problem with the fields of carstatemsgs/Polynome."""
        out = []
        for b, ft in enumerate(flat_trajs):
            M = int(res["n_pieces"][b])
            tail = ft.final_state.copy()
            tail[1, 0] = res["tail_s"][b]
            icr = (0.0, 0.0, 0.0) if self.cfg.standard_diff else (0.0, 0.0, self.cfg.icr_xv)
            out.append({"innerpoints": res["inner"][b, :M - 1].copy(), "t_pts": res["T"][b, :M].copy(),
                        "init_pva": np.array([ft.start_state[0, 0], ft.start_state[1, 0], ft.start_state[0, 1], ft.start_state[1, 1],
                                              ft.start_state[0, 2], ft.start_state[1, 2]]),
                        "tail_pva": np.array([tail[0, 0], tail[1, 0], tail[0, 1], tail[1, 1], tail[0, 2], tail[1, 2]]),
                        "start_position": np.array(ft.start_xytheta, dtype=np.float64), "ICR": np.array(icr),
                        "traj_start_time": traj_start_time})
        return out
