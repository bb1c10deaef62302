"""oracle/ltv_mpc_oracle.py -- TEST INFRASTRUCTURE ONLY.

float64 NumPy restatement of the reference's linear time-varying MPC (planning_ddr_opt/mpc_controller/src/mpc.cpp):
  predict_motion      :233-269   stateTrans + predictMotion (rollout of the last output; its clamping quirks kept)
  linear_model        :217-231   getLinearModel (unicycle, Euler, linearised about the rollout)
  build_qp            :304-493   solveMPCV's matrices, written densely: Hessian, gradient, the equality /
                                 box / rate rows and their bounds, in the reference's variable order
                                 z = (states 3 (T - d), inputs 2 (T - d))
  solve_qp            the reference hands the QP to OSQP (external, un-vendored, version unpinned; eps 1e-6):
                      PARITY UNPINNED.  Here the QP is solved to 1e-10 by an independent dense method (ADMM on the
                      full KKT system with numpy.linalg, then a KKT certificate), which is what the GPU solver
                      (Riccati-structured ADMM) is compared with.
  get_cmd             :569-614   the relinearisation loop with a FIXED iteration count (the reference stops on a
                                 9.7 ms wall clock, i.e. after a machine-dependent number of iterations)
  ref_points          :634-690, 538-567   getRefPoints + smooth_yaw on top of the trajectory oracle
"""
from __future__ import annotations

import math

import numpy as np


class LtvParams:
    """mpc_controller/config/mpc3ms.yaml + plan_manager/config/car3ms.yaml"""

    def __init__(self, **kw):
        self.dt = 0.01
        self.T = 30            # predict_steps
        self.delay_num = 1
        self.Q = (15.0, 15.0, 0.0, 1.0)   # x, y, v, theta
        self.R = (0.0, 0.0)
        self.Rd = (1.0, 0.05)
        self.max_speed = 3.0
        self.min_speed = 0.0
        self.max_omega = 3.0
        self.max_accel = 2.0
        self.max_domega = 4.0
        self.__dict__.update(kw)

    @property
    def max_cv(self): return self.max_accel * self.dt
    @property
    def max_comega(self): return self.max_domega * self.dt


def state_trans(s, a, yaw_dot, p: LtvParams):
    """stateTrans: s = [x, y, theta, v]; note the speed clamp acts on the OLD s.v, which is then overwritten."""
    yaw_dot = min(max(yaw_dot, -p.max_omega), p.max_omega)
    x, y, th, v = s
    x = x + a * math.cos(th) * p.dt
    y = y + a * math.sin(th) * p.dt
    th = th + yaw_dot * p.dt
    return [x, y, th, a]


def predict_motion(now_state, output, p: LtvParams):
    xbar = [list(now_state)]
    tmp = list(now_state)
    for i in range(1, p.T + 1):
        tmp = state_trans(tmp, output[0, i - 1], output[1, i - 1], p)
        xbar.append(list(tmp))
    return np.array(xbar)


def linear_model(s, p: LtvParams):
    th, v = s[2], s[3]
    B = np.zeros((3, 2)); B[0, 0] = math.cos(th) * p.dt; B[1, 0] = math.sin(th) * p.dt; B[2, 1] = p.dt
    A = np.eye(3); A[0, 2] = -B[1, 0] * v; A[1, 2] = B[0, 0] * v
    C = np.zeros(3); C[0] = -A[0, 2] * th; C[1] = -A[1, 2] * th
    return A, B, C


def build_qp(xbar, xref, dref, p: LtvParams):
    """P, q, Aeq rows etc. exactly as solveMPCV lays them out.  Returns (P, q, A, l, u, K)."""
    K = p.T - p.delay_num
    dimx, dimu = 3 * K, 2 * K
    nx = dimx + dimu
    q = np.zeros(nx)
    for i in range(K):
        j = p.delay_num + i
        q[3 * i] = -2 * p.Q[0] * xref[0, j]
        q[3 * i + 1] = -2 * p.Q[1] * xref[1, j]
        q[3 * i + 2] = -2 * p.Q[3] * xref[2, j]
        q[dimx + 2 * i] = -2 * p.Q[2] * dref[0, j]
    P = np.zeros((nx, nx))
    for i in range(K):
        P[3 * i, 3 * i] = 2 * p.Q[0]; P[3 * i + 1, 3 * i + 1] = 2 * p.Q[1]; P[3 * i + 2, 3 * i + 2] = 2 * p.Q[3]
    for i in range(K):
        ends = (i == 0 or i == K - 1)
        P[dimx + 2 * i, dimx + 2 * i] = 2 * (p.R[0] + (1 if ends else 2) * p.Rd[0] + p.Q[2])
        P[dimx + 2 * i + 1, dimx + 2 * i + 1] = 2 * (p.R[1] + (1 if ends else 2) * p.Rd[1])
    for i in range(K - 1):
        for c in range(2):
            P[dimx + 2 * (i + 1) + c, dimx + 2 * i + c] = P[dimx + 2 * i + c, dimx + 2 * (i + 1) + c] = -2 * p.Rd[c]
    mx, my, mz = dimu, dimx, 2 * K - 2
    A = np.zeros((mx + my + mz, nx)); l = np.zeros(mx + my + mz); u = np.zeros(mx + my + mz)
    for i in range(K):  # input box
        A[2 * i, dimx + 2 * i] = 1; A[2 * i + 1, dimx + 2 * i + 1] = 1
        l[2 * i], u[2 * i] = -p.max_speed, p.max_speed
        l[2 * i + 1], u[2 * i + 1] = -p.max_omega, p.max_omega
    for jj in range(K):  # dynamics, linearised about xbar[delay + jj]
        Am, Bm, Cm = linear_model(xbar[p.delay_num + jj], p)
        r = mx + 3 * jj
        A[r:r + 3, 3 * jj:3 * jj + 3] = np.eye(3)
        A[r:r + 3, dimx + 2 * jj:dimx + 2 * jj + 2] = -Bm
        if jj == 0:
            b = Am @ xbar[p.delay_num][:3] + Cm
        else:
            A[r:r + 3, 3 * (jj - 1):3 * jj] = -Am
            b = Cm
        l[r:r + 3] = u[r:r + 3] = b
    for i in range(K - 1):  # rate limits
        for c in range(2):
            r = mx + my + 2 * i + c
            A[r, dimx + 2 * i + c] = -1.0; A[r, dimx + 2 * (i + 1) + c] = 1.0
        l[mx + my + 2 * i], u[mx + my + 2 * i] = -p.max_cv, p.max_cv
        l[mx + my + 2 * i + 1], u[mx + my + 2 * i + 1] = -p.max_comega, p.max_comega
    return P, q, A, l, u, K


def solve_qp(P, q, A, l, u, tol=1e-10, max_iter=20000):
    """min 0.5 z'Pz + q'z  s.t. l <= Az <= u, by ADMM on the dense KKT system (rho per row: large for equalities),
    to `tol`; returns (z, y, info) with info = KKT residuals (the certificate the tests check)."""
    n, m = P.shape[0], A.shape[0]
    eq = np.abs(u - l) < 1e-12
    z = np.zeros(n); w = np.zeros(m); y = np.zeros(m)
    rho_base, sigma = 10.0, 1e-6
    for outer in range(12):
        rho = np.where(eq, 1e3 * rho_base, rho_base)
        KKT = np.block([[P + sigma * np.eye(n), A.T], [A, -np.diag(1.0 / rho)]])
        lu = np.linalg.inv(KKT)
        done = False
        for it in range(max_iter // 12):
            rhs = np.concatenate([sigma * z - q, w - y / rho])
            sol = lu @ rhs
            zt, nu = sol[:n], sol[n:]
            wt = w + (nu - y) / rho
            z = zt
            w_new = np.clip(wt + y / rho, l, u)
            y = y + rho * (wt - w_new)
            w = w_new
            rp = np.max(np.abs(A @ z - w)); rd = np.max(np.abs(P @ z + q + A.T @ y))
            if rp < tol and rd < tol:
                done = True
                break
        if done:
            break
        # residual balancing
        rp = max(rp, 1e-30); rd = max(rd, 1e-30)
        rho_base *= math.sqrt(rp / rd) if rd > 0 else 1.0
        rho_base = min(max(rho_base, 1e-4), 1e6)
    Az = A @ z
    info = {"primal": float(max(np.max(np.maximum(l - Az, 0)), np.max(np.maximum(Az - u, 0)))),
            "dual": float(np.max(np.abs(P @ z + q + A.T @ y))),
            "compl": float(max(np.max(np.abs(np.minimum(y, 0) * (Az - l))), np.max(np.abs(np.maximum(y, 0) * (u - Az)))))}
    return z, y, info


def solve_mpcv(now_state, output, output_buff, xref, dref, p: LtvParams):
    """one predictMotion + solveMPCV: returns the new output (2 x T), the QP solution and its certificate"""
    xbar = predict_motion(now_state, output, p)
    P, q, A, l, u, K = build_qp(xbar, xref, dref, p)
    z, y, info = solve_qp(P, q, A, l, u)
    out = output.copy()
    for i in range(p.delay_num):
        out[:, i] = output_buff[i]
    for j in range(K):
        out[0, j + p.delay_num] = z[3 * K + 2 * j]
        out[1, j + p.delay_num] = z[3 * K + 2 * j + 1]
    return out, z, info, xbar


def get_cmd(now_state, output, output_buff, xref, dref, p: LtvParams, n_relin: int):
    infos = []
    for _ in range(n_relin):
        output, z, info, xbar = solve_mpcv(now_state, output, output_buff, xref, dref, p)
        infos.append(info)
    buff = list(output_buff)
    if p.delay_num > 0:
        buff = buff[1:] + [output[:, p.delay_num].copy()]
    return output, buff, infos


def ref_points(sampler, now, est_theta, p: LtvParams):
    """getRefPoints (x, y, theta normalised, v = s', w = theta' at t_cur + (i + 1) dt, clamped at the end) followed
    by smooth_yaw; `sampler` is an oracle.traj_driver.RefSampler whose trajectory started at time 0."""
    dur = sampler.duration()
    xref = np.zeros((3, p.T)); dref = np.zeros((2, p.T))
    for i in range(p.T):
        t = min(now + (i + 1) * p.dt, dur) if now + (i + 1) * p.dt <= dur else dur
        pose, v, _ = sampler.state(t)
        th = pose[2]
        while th > math.pi: th -= 2 * math.pi
        while th < -math.pi: th += 2 * math.pi
        xref[:, i] = (pose[0], pose[1], th)
        dref[:, i] = (v[1], v[0])
    dy = xref[2, 0] - est_theta
    while dy >= math.pi / 2: xref[2, 0] -= 2 * math.pi; dy = xref[2, 0] - est_theta
    while dy <= -math.pi / 2: xref[2, 0] += 2 * math.pi; dy = xref[2, 0] - est_theta
    for i in range(p.T - 1):
        dy = xref[2, i + 1] - xref[2, i]
        while dy >= math.pi / 2: xref[2, i + 1] -= 2 * math.pi; dy = xref[2, i + 1] - xref[2, i]
        while dy <= -math.pi / 2: xref[2, i + 1] += 2 * math.pi; dy = xref[2, i + 1] - xref[2, i]
    return xref, dref, now > dur + 1.0
