"""Host side of the batched NMPC engine (Python over the C ABI).

Two layers:

* ``BatchedNmpc`` -- thin owner of a solver handle plus a set of torch tensors
  laid out like the reference's ``ACADOvariables`` members with a leading batch
  dimension (include/alore_nmpc.h).  torch is used for device memory and
  streams only; the numerics are the HIP kernels behind ``alore_nmpc_rti``.
* ``BatchedMpcWrapper`` -- the batched mirror of the reference's
  ``Tracked_nmpc::MpcWrapper`` (planning_ddr_opt/nmpc_controller/
  include/nmpc_controller/mpc_wrapper.h:43-141, src/mpc_wrapper.cpp:33-410):
  same method names and argument meaning, every argument carrying one extra
  leading batch dimension.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import numpy as np

from . import _lib
from ._lib import BATCH_FLOAT_MEMBERS, BATCH_MEMBERS, Batch, Config, LaunchInfo, LinOut, NmpcError

NX, NU, NOD, NY, NYN = 3, 2, 3, 5, 3


def member_shapes(B: int, N: int) -> dict:
    return {
        "x": (B, N + 1, NX), "u": (B, N, NU), "od": (B, N + 1, NOD), "y": (B, N, NY), "yN": (B, NYN),
        "W": (B, N, NY, NY), "WN": (B, NYN, NYN), "x0": (B, NX), "lbValues": (B, N, NU),
        "ubValues": (B, N, NU), "dual": (B, N, NU), "status": (B,), "n_iter": (B,), "kkt": (B,), "obj": (B,),
    }


class BatchedNmpc:
    """B independent NMPC instances on one GPU.

    ``self.t[name]`` are the device tensors (float32; status/n_iter int32) of
    slot 0.  ``slots`` > 1 allocates that many independent copies of the whole
    batch (``self.ts[name]`` has a leading slot dimension); the benchmark uses
    one slot per timed step so that every step reads its inputs from HBM and
    starts from the same iterate."""

    def __init__(self, B: int, N: int = 20, dt: float = 0.01, device: int = 0, max_as_iter: int = 0,
                 lanes_per_problem: int = 0, slots: int = 1, warm_start_steps: int = -1, diagnostics: bool = True):
        import torch  # device memory + streams
        self.torch = torch
        self.lib = _lib.load()  # raises if the HIP library is missing: no fallback
        if not torch.cuda.is_available():
            raise _lib.NmpcLibraryError("no GPU visible to torch: the NMPC engine has no CPU path")
        self.B, self.N, self.dt, self.slots = int(B), int(N), float(dt), int(slots)
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        cfg = Config(self.N, self.dt, device, max_as_iter, lanes_per_problem, warm_start_steps)
        h = C.c_void_p()
        rc = self.lib.alore_nmpc_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise NmpcError(rc, "alore_nmpc_create failed")
        self.h = h
        self.ts = {}
        for k, shp in member_shapes(self.B, self.N).items():
            dt_ = torch.int32 if k in ("status", "n_iter") else torch.float32
            self.ts[k] = torch.zeros((self.slots,) + shp, dtype=dt_, device=self.device)
        self.t = {k: v[0] for k, v in self.ts.items()}
        # diagnostics=False: kkt / obj are not asked for (NULL pointers) and the kernel skips them, like the
        # reference's tick, which never calls acado_getKKT / acado_getObjective
        skip = () if diagnostics else ("kkt", "obj")
        self._batches = [Batch(**{k: (None if k in skip else self.ts[k][s].data_ptr()) for k in BATCH_MEMBERS})
                         for s in range(self.slots)]
        self._batch = self._batches[0]
        for b in self._batches:
            self._check(self.lib.alore_nmpc_batch_default_bounds(self.h, C.byref(b), self.B, self._stream()))

    # -- plumbing
    def _stream(self):
        raw = getattr(self.torch._C, "_cuda_getCurrentRawStream", None)   # the raw handle without building a Stream object (a third of the call's cost)
        if raw is not None:
            return C.c_void_p(raw(self.device.index))
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def _check(self, rc):
        if rc != 0:
            raise NmpcError(rc, self.lib.alore_nmpc_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.alore_nmpc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- data
    def load(self, batch: dict, slot: Optional[int] = 0) -> None:
        """Copy host arrays (any subset of the float members) to the device;
        ``slot=None`` fills every slot."""
        torch = self.torch
        shapes = member_shapes(self.B, self.N)
        for k, v in batch.items():
            if k not in BATCH_FLOAT_MEMBERS:
                continue
            a = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32).reshape(shapes[k]))
            if slot is None:
                d = a.to(self.device)
                self.ts[k].copy_(d.unsqueeze(0).expand_as(self.ts[k]))
            else:
                self.ts[k][slot].copy_(a, non_blocking=False)

    def fetch(self, names=("x", "u", "dual", "status", "n_iter", "kkt", "obj"), slot: int = 0) -> dict:
        self.torch.cuda.synchronize(self.device)
        return {k: self.ts[k][slot].detach().cpu().numpy().copy() for k in names}

    # -- the hot path
    def rti(self, n_sqp: int = 1, slot: int = 0) -> None:
        """n_sqp x (acado_preparationStep + acado_feedbackStep) for the whole batch, one launch."""
        self._check(self.lib.alore_nmpc_rti(self.h, C.byref(self._batches[slot]), self.B, int(n_sqp), self._stream()))

    def rti_sync(self, n_sqp: int = 1, slot: int = 0) -> None:
        """rti() and wait for it (alore_nmpc_rti + alore_nmpc_synchronize on the current stream): the synchronous control tick"""
        st = self._stream()
        rc = self.lib.alore_nmpc_rti(self.h, C.byref(self._batches[slot]), self.B, n_sqp, st)
        if rc == 0:
            rc = self.lib.alore_nmpc_synchronize(self.h, st)
        if rc != 0:
            self._check(rc)

    def input_column(self, node: int, slot: int = 0):
        """inputs of one node of every problem and the status (alore_nmpc_input_column): (cmd [B][2] float32, status [B])"""
        import numpy as np
        cmd = np.empty((self.B, 2), np.float32)
        st = np.empty(self.B, np.int32)
        self._check(self.lib.alore_nmpc_input_column(self.h, C.byref(self._batches[slot]), self.B, int(node),
                                                     cmd.ctypes.data_as(C.POINTER(C.c_float)), st.ctypes.data_as(C.POINTER(C.c_int)), self._stream()))
        return cmd, st

    def set_launch_overlap(self, ways: int) -> None:
        """launches of independent slots kept in flight at once by rti_range (alore_nmpc_set_launch_overlap; 1 = in order)"""
        self._check(self.lib.alore_nmpc_set_launch_overlap(self.h, int(ways)))

    def condense(self, slot: int = 0) -> dict:
        """The condensed QP of the batch as it stands (alore_nmpc_condense: acadoWorkspace.H / g / lb / ub of the reference):
        H (B, 2N, 2N), g, lb, ub (B, 2N), device tensors."""
        torch = self.torch
        n = 2 * self.N
        out = {"H": torch.empty((self.B, n, n), dtype=torch.float32, device=self.device)}
        for k in ("g", "lb", "ub"):
            out[k] = torch.empty((self.B, n), dtype=torch.float32, device=self.device)
        qp = (C.c_void_p * 4)(out["H"].data_ptr(), out["g"].data_ptr(), out["lb"].data_ptr(), out["ub"].data_ptr())
        self._check(self.lib.alore_nmpc_condense(self.h, C.addressof(self._batches[slot]), self.B, C.addressof(qp), self._stream()))
        return out

    def dense_qp(self, H, g, lb, ub, y0=None):
        """acado_solve() for a batch of dense box QPs (alore_nmpc_dense_qp): min 1/2 x'Hx + g'x, lb <= x <= ub; device tensors
        H (B, n, n), g / lb / ub (B, n); returns x, y (multipliers, > 0 lower / < 0 upper), status, n_iter."""
        torch = self.torch
        Bq, n = g.shape
        x = torch.empty_like(g)
        y = torch.zeros_like(g) if y0 is None else y0.clone().contiguous()
        st = torch.empty(Bq, dtype=torch.int32, device=g.device); ni = torch.empty(Bq, dtype=torch.int32, device=g.device)
        H, g, lb, ub = (t.contiguous() for t in (H, g, lb, ub))
        qp = (C.c_void_p * 4)(H.data_ptr(), g.data_ptr(), lb.data_ptr(), ub.data_ptr())
        self._check(self.lib.alore_nmpc_dense_qp(self.h, Bq, n, C.addressof(qp), x.data_ptr(), y.data_ptr(), st.data_ptr(), ni.data_ptr(),
                                                 self._stream()))
        return x, y, st, ni

    def set_problem_mask(self, mask) -> None:
        """mask [B] (1 = solve, 0 = leave the problem exactly as it is) for the following rti() calls, None = all
        (alore_nmpc_set_problem_mask); kept on the device by this object"""
        if mask is None:
            self._mask = None
            self._check(self.lib.alore_nmpc_set_problem_mask(self.h, None))
            return
        m = self.torch.as_tensor(np.ascontiguousarray(mask, dtype=np.uint8), device=self.device).reshape(self.B).contiguous()
        self._mask = m
        self._check(self.lib.alore_nmpc_set_problem_mask(self.h, C.c_void_p(m.data_ptr())))

    def set_many_mode(self, mode: str) -> None:
        """how rti_range keeps independent slots in flight (alore_nmpc_set_many_mode): "groups" -- up to 24 slots per grid,
        the default -- or "streams" -- one launch per slot on forked streams"""
        self._check(self.lib.alore_nmpc_set_many_mode(self.h, {"groups": 0, "streams": 1}[mode]))

    def set_two_phase(self, mode) -> None:
        """two-phase grids of rti_range (alore_nmpc_set_two_phase): "auto" / -1, False / 0 (never), True / 1 (wherever the build exists)"""
        m = {"auto": -1, False: 0, True: 1, -1: -1, 0: 0, 1: 1}[mode]
        self._check(self.lib.alore_nmpc_set_two_phase(self.h, m))

    def two_phase_info(self) -> dict:
        from ._lib import TwoPhaseInfo
        ti = TwoPhaseInfo()
        self._check(self.lib.alore_nmpc_get_two_phase_info(self.h, C.byref(ti)))
        return {k: getattr(ti, k) for k, _ in TwoPhaseInfo._fields_}

    def rti_range(self, first: int, count: int, n_sqp: int = 1) -> None:
        """slots first .. first + count - 1 by ONE call into the library (independent slots are solved together, see
        alore_nmpc_rti_many)"""
        args = getattr(self, "_range_args", {}).get((first, count))   # made by prepare_range: nothing but the call is left
        if args is None:
            if first < 0 or count < 1 or first + count > self.slots:
                raise ValueError(f"rti_range: slots {first} .. {first + count - 1} outside 0 .. {self.slots - 1}")
            if not hasattr(self, "_batch_array"):
                self._batch_array = (Batch * self.slots)(*self._batches)
            args = (C.cast(C.byref(self._batch_array, first * C.sizeof(Batch)), C.POINTER(Batch)), C.c_int(count), C.c_int(self.B))
        rc = self.lib.alore_nmpc_rti_many(self.h, args[0], args[1], args[2], n_sqp, self._stream())
        if rc != 0:
            self._check(rc)

    def prepare_range(self, first: int, count: int) -> None:
        """the independence check of rti_range(first, count) ahead of time (alore_nmpc_rti_many_prepare): later calls on these
        slots, or on any contiguous run of them, go straight to the launch"""
        if first < 0 or count < 1 or first + count > self.slots:
            raise ValueError(f"prepare_range: slots {first} .. {first + count - 1} outside 0 .. {self.slots - 1}")
        if not hasattr(self, "_batch_array"):
            self._batch_array = (Batch * self.slots)(*self._batches)
        ptr = C.cast(C.byref(self._batch_array, first * C.sizeof(Batch)), C.POINTER(Batch))
        self._check(self.lib.alore_nmpc_rti_many_prepare(self.h, ptr, int(count), self.B))
        if not hasattr(self, "_range_args"):
            self._range_args = {}
        self._range_args[(first, count)] = (ptr, C.c_int(count), C.c_int(self.B))

    def linearize(self) -> dict:
        torch = self.torch
        d = torch.empty((self.B, self.N, 3), dtype=torch.float32, device=self.device)
        gx = torch.empty((self.B, self.N, 9), dtype=torch.float32, device=self.device)
        gu = torch.empty((self.B, self.N, 6), dtype=torch.float32, device=self.device)
        out = LinOut(d.data_ptr(), gx.data_ptr(), gu.data_ptr())
        self._check(self.lib.alore_nmpc_linearize(self.h, C.byref(self._batch), self.B, C.byref(out), self._stream()))
        torch.cuda.synchronize(self.device)
        return {"d": d.cpu().numpy(), "evGx": gx.cpu().numpy(), "evGu": gu.cpu().numpy()}

    def forward_simulate(self) -> None:
        self._check(self.lib.alore_nmpc_forward_simulate(self.h, C.byref(self._batch), self.B, self._stream()))

    def shift(self, strategy: int = 0, xEnd=None, uEnd=None) -> None:
        torch = self.torch
        xe = torch.as_tensor(np.asarray(xEnd, np.float32), device=self.device).reshape(self.B, 3).contiguous() \
            if xEnd is not None else None
        ue = torch.as_tensor(np.asarray(uEnd, np.float32), device=self.device).reshape(self.B, 2).contiguous() \
            if uEnd is not None else None
        self._check(self.lib.alore_nmpc_shift(self.h, C.byref(self._batch), self.B, int(strategy),
                                              C.c_void_p(xe.data_ptr()) if xe is not None else None,
                                              C.c_void_p(ue.data_ptr()) if ue is not None else None, self._stream()))

    # -- reference sampling on the device (include/alore_nmpc.h: alore_nmpc_refs_*)
    def refs_init(self, max_pieces: int = 64, max_checkpoints: int = 1024) -> None:
        self._refs_shape = (int(max_pieces), int(max_checkpoints))
        self._check(self.lib.alore_nmpc_refs_init(self.h, self.B, max_pieces, max_checkpoints))

    def refs_set_from_backend(self, planner, count: int | None = None, traj_start_time: float = 0.0, xv: float = 0.0,
                              state_seq_res: float = 0.1, integral_res_int: int = 4) -> None:
        """Trajectory store <- the plans of a BatchedMSPlanner on the same GPU, device to device (problem t -> slot t)."""
        view = planner.device_view()
        self._check(self.lib.alore_nmpc_refs_set_from_backend(self.h, C.addressof(view), int(count or planner.count),
                                                              float(traj_start_time), float(xv), float(state_seq_res),
                                                              int(integral_res_int), self._stream()))

    def refs_set_polynomes(self, robots, msgs, state_seq_res: float = 0.1, integral_res_int: int = 4) -> None:
        """msgs: objects with the fields of Polynome.msg (alore_legged_manipulator_amd.host.Polynome)."""
        n = len(msgs)
        arr = (_lib.PolynomeMsg * n)()
        keep = []
        for q, m in zip(arr, msgs):
            inner = np.ascontiguousarray(m.innerpoints, np.float64).reshape(-1)
            t_pts = np.ascontiguousarray(m.t_pts, np.float64)
            keep += [inner, t_pts]
            q.n_pieces = t_pts.size
            q.innerpoints = inner.ctypes.data if inner.size else None
            q.t_pts = t_pts.ctypes.data
            pva0, pva1 = np.asarray(m.init_pva, np.float64), np.asarray(m.tail_pva, np.float64)
            for d in range(2):
                q.init_p[d], q.init_v[d], q.init_a[d] = pva0[d], pva0[2 + d], pva0[4 + d]
                q.tail_p[d], q.tail_v[d], q.tail_a[d] = pva1[d], pva1[2 + d], pva1[4 + d]
            for k in range(3):
                q.start_position[k] = float(m.start_position[k]); q.ICR[k] = float(m.ICR[k])
            q.traj_start_time = float(m.traj_start_time)
        rb = np.ascontiguousarray(robots, np.int32)
        self._check(self.lib.alore_nmpc_refs_set_polynomes(self.h, n, rb.ctypes.data, C.addressof(arr) if n else None,
                                                           float(state_seq_res), int(integral_res_int), self._stream()))

    def refs_download(self, robot: int) -> dict:
        P, Cn = self._refs_shape
        meta = np.zeros(8); dur = np.zeros(P); coef = np.zeros((P, 2, 6)); ck = np.zeros((Cn, 2))
        self._check(self.lib.alore_nmpc_refs_download(self.h, int(robot), meta.ctypes.data, dur.ctypes.data,
                                                      coef.ctypes.data, ck.ctypes.data))
        n, c = int(meta[4]), int(meta[5])
        return {"start_time": meta[0], "duration": meta[1], "xv": meta[2], "res": meta[3], "valid": bool(meta[6]),
                "durations": dur[:n], "coeffs": coef[:n], "checkpoints": ck[:c]}

    def refs_sample(self, now: float, est, icr, smooth: bool = True, slot: int = 0):
        """Writes y, yN, od, x0 of `slot` from the stored trajectories; returns the at_goal flags."""
        est = np.ascontiguousarray(est, np.float64).reshape(self.B, 3)
        icr = np.ascontiguousarray(icr, np.float64).reshape(self.B, 3)
        goal = np.zeros(self.B, np.int32)
        self._check(self.lib.alore_nmpc_refs_sample(self.h, C.byref(self._batches[slot]), self.B, float(now), est.ctypes.data,
                                                    icr.ctypes.data, 1 if smooth else 0, goal.ctypes.data, self._stream()))
        return goal.astype(bool)

    # -- closed loop on the device (include/alore_nmpc.h: alore_nmpc_plant_*, alore_nmpc_closed_loop_tick)
    def plant_init(self, max_acc: float = 2.0, max_domega: float = 4.0, pose_pub_rate: float = 100.0,
                   state_propa_rate: float = 500.0, substeps: int = 5) -> None:
        p = _lib.PlantParams(max_acc, max_domega, 1.0 / pose_pub_rate, 1.0 / state_propa_rate, substeps)
        self._check(self.lib.alore_nmpc_plant_init(self.h, C.byref(p)))

    def plant_set_state(self, pose, icr, vw=None) -> None:
        pose = np.ascontiguousarray(pose, np.float64).reshape(self.B, 3)
        icr = np.ascontiguousarray(icr, np.float64).reshape(self.B, 3)
        vwp = None
        if vw is not None:
            vw = np.ascontiguousarray(vw, np.float64).reshape(self.B, 2)
            vwp = vw.ctypes.data
        self._check(self.lib.alore_nmpc_plant_set_state(self.h, self.B, pose.ctypes.data, vwp, icr.ctypes.data, self._stream()))

    def plant_get_state(self):
        pose = np.zeros((self.B, 3)); vw = np.zeros((self.B, 2)); goal = np.zeros(self.B, np.int32)
        self._check(self.lib.alore_nmpc_plant_get_state(self.h, self.B, pose.ctypes.data, vw.ctypes.data, goal.ctypes.data,
                                                        self._stream()))
        return pose, vw, goal.astype(bool)

    def closed_loop_reset(self, mask=None, slot: int = 0) -> None:
        """MpcWrapper::solve's cold start from the plant's pose (x <- pose replicated, u <- 0) for the masked robots."""
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        self._check(self.lib.alore_nmpc_closed_loop_reset(self.h, C.byref(self._batches[slot]), self.B,
                                                          None if m is None else m.ctypes.data, self._stream()))

    def refs_eval(self, now: float) -> np.ndarray:
        """(B, 4): theta', s', theta'', s'' of every stored trajectory at `now` (zeros without a trajectory)."""
        out = np.zeros((self.B, 4))
        self._check(self.lib.alore_nmpc_refs_eval(self.h, self.B, float(now), out.ctypes.data, self._stream()))
        return out

    def closed_loop_tick(self, now: float, delay_num: int = 1, slot: int = 0) -> None:
        """References from the plant's pose -> one real-time iteration -> command to the plant; no host sync."""
        self._check(self.lib.alore_nmpc_closed_loop_tick(self.h, C.byref(self._batches[slot]), self.B, float(now),
                                                         int(delay_num), self._stream()))

    def closed_loop_run(self, t0: float, dt_tick: float, n_ticks: int, delay_num: int = 1, slot: int = 0) -> None:
        """n_ticks closed-loop ticks at t0, t0 + dt_tick, ... enqueued back to back by the library."""
        self._check(self.lib.alore_nmpc_closed_loop_run(self.h, C.byref(self._batches[slot]), self.B, float(t0), float(dt_tick),
                                                        int(n_ticks), int(delay_num), self._stream()))

    def set_shared_members(self, W: bool = False, bounds: bool = False, od: bool = False) -> None:
        """Members that are ONE copy for the whole batch (row 0 of the corresponding tensors is what every problem reads)."""
        self._check(self.lib.alore_nmpc_set_shared_members(self.h, (1 if W else 0) | (2 if bounds else 0) | (4 if od else 0)))

    def set_timing(self, enable: bool) -> None:
        self._check(self.lib.alore_nmpc_set_timing(self.h, 1 if enable else 0))

    def launch_info(self) -> dict:
        li = LaunchInfo()
        self._check(self.lib.alore_nmpc_get_launch_info(self.h, C.byref(li)))
        return {k: getattr(li, k) for k, _ in LaunchInfo._fields_}


class BatchedMpcWrapper:
    """Batched mirror of ``Tracked_nmpc::MpcWrapper`` (reference:
    nmpc_controller/src/mpc_wrapper.cpp).  Array arguments take the reference's
    Eigen shapes with one leading batch dimension; a shape without the batch
    dimension is broadcast to every problem.

    Differences, all forced by batching/fusion and documented in DESIGN.md:
    the preparation step is fused into the next feedback launch, so ``prepare``
    only flips the readiness flag the reference keeps (``acado_is_prepared_``).
    """

    kSamples = None  # set per instance (reference: ACADO_N, compile-time 50)
    kStateSize, kRefSize, kEndRefSize, kInputSize, kCostSize, kOdSize = NX, NY, NYN, NU, NY - NU, NOD

    def __init__(self, B: int, N: int = 20, Q=None, R=None, dt: float = 0.01, device: int = 0, **kw):
        self.kSamples = N
        self.B = B
        self.eng = BatchedNmpc(B, N, dt=dt, device=device, **kw)
        self.dt_ = dt
        self.W_ = np.zeros((NY, NY))
        self.WN_ = np.zeros((NYN, NYN))
        # mpc_wrapper.cpp:33-93: zero everything, default ICR (0,-0.2,0.2), forward-simulate, prepare
        if Q is not None and R is not None:
            self.setCosts(Q, R)
        self.setICRParameters(np.array([0.0, -0.2, 0.2]))
        self.eng.forward_simulate()
        self.acado_is_prepared_ = True

    def _b(self, a, tail):
        a = np.asarray(a, dtype=np.float64)
        if a.shape == tuple(tail):
            a = np.broadcast_to(a, (self.B,) + tuple(tail))
        assert a.shape == (self.B,) + tuple(tail), (a.shape, tail)
        return a

    # mpc_wrapper.cpp:106-138
    def setCosts(self, Q, R, state_cost_scaling: float = 0.0, input_cost_scaling: float = 0.0) -> bool:
        if state_cost_scaling < 0.0 or input_cost_scaling < 0.0:
            return False
        N = self.kSamples
        self.W_[:3, :3] = np.asarray(Q, np.float64)
        self.W_[3:, 3:] = np.asarray(R, np.float64)
        self.WN_ = self.W_[:3, :3].copy()
        W = np.zeros((N, NY, NY), np.float32)
        state_scale = np.float32(1.0)
        for i in range(N):
            state_scale = np.float32(math.exp(-np.float32(i) / np.float32(N) * np.float32(state_cost_scaling)))
            input_scale = np.float32(math.exp(-np.float32(i) / np.float32(N) * np.float32(input_cost_scaling)))
            W[i, :3, :3] = self.W_[:3, :3].astype(np.float32) * state_scale
            W[i, 3:, 3:] = self.W_[3:, 3:].astype(np.float32) * input_scale
        WN = self.WN_.astype(np.float32) * state_scale
        self.eng.load({"W": np.broadcast_to(W, (self.B,) + W.shape), "WN": np.broadcast_to(WN, (self.B, 3, 3))})
        return True

    # mpc_wrapper.cpp:200-207 -- (xv, yr, yl) broadcast to all N+1 nodes
    def setICRParameters(self, p_B_I) -> bool:
        p = self._b(p_B_I, (3,))
        self.eng.load({"od": np.broadcast_to(p[:, None, :], (self.B, self.kSamples + 1, 3))})
        return True

    # mpc_wrapper.cpp:219-239
    def setReferencePose(self, state) -> bool:
        s = self._b(state, (3,))
        N = self.kSamples
        y = np.zeros((self.B, N, NY))
        y[:, :, :3] = s[:, None, :]
        self.eng.load({"y": y, "yN": s})
        self.eng.forward_simulate()
        return True

    # mpc_wrapper.cpp:242-264 -- states [B,3,N+1], inputs [B,2,N+1] (Eigen shapes)
    def setTrajectory(self, states, inputs) -> bool:
        N = self.kSamples
        st = self._b(states, (3, N + 1))
        inp = self._b(inputs, (2, N + 1))
        y = np.concatenate([st[:, :, :N].transpose(0, 2, 1), inp[:, :, :N].transpose(0, 2, 1)], axis=2)
        self.eng.load({"y": y, "yN": st[:, :, N]})
        return True

    # mpc_wrapper.cpp:267-275
    def solve(self, state) -> bool:
        s = self._b(state, (3,))
        N = self.kSamples
        self.eng.load({"x": np.broadcast_to(s[:, None, :], (self.B, N + 1, 3)), "u": np.zeros((self.B, N, 2))})
        return self.update(s)

    # mpc_wrapper.cpp:279-373
    def update(self, state, do_preparation: bool = True) -> bool:
        if not self.acado_is_prepared_:
            return False
        self.eng.load({"x0": self._b(state, (3,))})
        self.eng.rti(1)
        self.acado_is_prepared_ = False
        if do_preparation:
            self.acado_is_prepared_ = True
        return True

    # mpc_wrapper.cpp:377-383
    def prepare(self) -> bool:
        self.acado_is_prepared_ = True
        return True

    # mpc_wrapper.cpp:386-410 (double out, Eigen shapes)
    def getState(self, node_index: int):
        return self.eng.t["x"][:, node_index, :].double().cpu().numpy()

    def getStates(self):
        return self.eng.t["x"].double().cpu().numpy().transpose(0, 2, 1)

    def getInput(self, node_index: int):
        return self.eng.t["u"][:, node_index, :].double().cpu().numpy()

    def getInputs(self):
        return self.eng.t["u"].double().cpu().numpy().transpose(0, 2, 1)

    def getTimestep(self) -> float:
        return self.dt_

    def getStatus(self):
        """Per-problem QP status (the reference computes and drops it, mpc_wrapper.cpp:298)."""
        return self.eng.t["status"].cpu().numpy()
