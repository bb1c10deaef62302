#!/usr/bin/env python3
"""Prototype (NumPy): the backward Riccati recursion of one planar NMPC stage problem as an ASSOCIATIVE SCAN over the stages
(Sarkka & Garcia-Fernandez, "Temporal parallelization of dynamic programming and linear quadratic control", IEEE TAC 2023), in
float32, against the sequential recursion in float64 -- is it accurate enough to replace the 20 sequential stage steps of a sweep
(13 k of the 40 k cycles of a one-batch launch) by 5 combine levels over the lanes?

Stage k: x+ = A x + B u + d, cost 1/2 x'Qx + q'x + x'Su + 1/2 u'Ru + r'u, inputs of the working set held at given values.
Element of the interval [i, j): (A, b, C, eta, J) of the conditional value function
    V(xi, xj) = max_l  1/2 xi' J xi + eta' xi - 1/2 l' C l - l' (xj - A xi - b)
(eta carries the linear term with the sign used here), combine rule of the paper; the suffix products give P_k = J, p_k = eta of
the cost-to-go 1/2 x' P_k x + p_k' x at every stage at once."""
import sys
import numpy as np


def stage_element(A, B, d, Q, q, S, R, r, held, dt):
    """element of one stage with the inputs `held` (dict index -> value) eliminated"""
    free = [i for i in range(B.shape[1]) if i not in held]
    Ae, de, Qe, qe = A.copy(), d.copy(), Q.copy(), q.copy()
    re = r.copy()
    for i, v in held.items():
        de = de + B[:, i] * v
        qe = qe + S[:, i] * v
        re = re + R[:, i] * v
    if not free:
        return Ae, de, np.zeros_like(Q), qe, Qe
    Bf, Rf, Sf, rf = B[:, free], R[np.ix_(free, free)], S[:, free], re[free]
    Ri = np.linalg.inv(Rf)
    # u = ut - Ri (Sf' x + rf): removes the cross term and the linear input term
    Ae = Ae - Bf @ Ri @ Sf.T
    de = de - Bf @ Ri @ rf
    C = Bf @ Ri @ Bf.T
    J = Qe - Sf @ Ri @ Sf.T
    eta = qe - Sf @ Ri @ rf
    return Ae, de, C, eta, J


def combine(e1, e2, dt):
    """[i, j) then [j, k)"""
    A1, b1, C1, h1, J1 = e1
    A2, b2, C2, h2, J2 = e2
    n = A1.shape[0]
    I = np.eye(n, dtype=dt)
    M = np.linalg.inv(I + C1 @ J2).astype(dt)           # (I + C1 J2)^-1
    A = A2 @ M @ A1
    b = A2 @ M @ (b1 - C1 @ h2) + b2
    C = A2 @ M @ C1 @ A2.T + C2
    Mt = np.linalg.inv(I + J2 @ C1).astype(dt)
    h = A1.T @ Mt @ (h2 + J2 @ b1) + h1
    J = A1.T @ Mt @ J2 @ A1 + J1
    return tuple(x.astype(dt) for x in (A, b, C, h, J))


def sequential(stages, QN, qN, held_sets):
    """float64 reference: P_k, p_k for k = 0 .. N"""
    N = len(stages)
    P, p = [None] * (N + 1), [None] * (N + 1)
    P[N], p[N] = QN.copy(), qN.copy()
    for k in range(N - 1, -1, -1):
        A, B, d, Q, q, S, R, r = stages[k]
        Ae, de, C, eta, J = stage_element(A, B, d, Q, q, S, R, r, held_sets[k], np.float64)
        # V_k(x) = 1/2 x'Jx + eta'x + min over the free inputs folded into C: P = J + Ae' (P+ ^-1 + C)^-1 Ae in dual form
        Pn, pn = P[k + 1], p[k + 1]
        M = np.linalg.inv(np.eye(3) + C @ Pn)
        P[k] = J + Ae.T @ Pn @ M @ Ae
        P[k] = 0.5 * (P[k] + P[k].T)
        p[k] = eta + Ae.T @ np.linalg.inv(np.eye(3) + Pn @ C) @ (pn + Pn @ de)
    return P, p


def scan(stages, QN, qN, held_sets, dt):
    N = len(stages)
    el = []
    for k in range(N):
        A, B, d, Q, q, S, R, r = [a.astype(dt) for a in stages[k]]
        el.append(tuple(x.astype(dt) for x in stage_element(A, B, d, Q, q, S, R, r, held_sets[k], dt)))
    z = np.zeros((3, 3), dt)
    el.append((z.copy(), np.zeros(3, dt), z.copy(), qN.astype(dt), QN.astype(dt)))   # terminal node
    # Hillis-Steele suffix scan: after the level with stride s, el[k] covers [k, min(k + 2 s, N + 1))
    s = 1
    while s < N + 1:
        new = list(el)
        for k in range(N + 1):
            if k + s < N + 1:
                new[k] = combine(el[k], el[k + s], dt)
        el = new
        s *= 2
    return [e[4] for e in el], [e[3] for e in el]


def main():
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    N, h = 20, 0.01
    worst = {np.float32: 0.0, np.float64: 0.0}
    for trial in range(400):
        stages, held = [], []
        th, v = rng.uniform(-3, 3), rng.uniform(0.2, 1.5)
        for k in range(N):
            th += rng.normal(0, 0.05)
            c, s_ = np.cos(th), np.sin(th)
            yr, yl, xv = -0.3 + rng.normal(0, 0.01), 0.3 + rng.normal(0, 0.01), 0.1
            A = np.eye(3); A[0, 2] = -h * v * s_; A[1, 2] = h * v * c
            # wheel speeds -> (v, omega): v = (vr yl - vl yr) / (yl - yr), w = (vr - vl) / (yl - yr)
            dv = np.array([yl, -yr]) / (yl - yr); dw = np.array([1.0, -1.0]) / (yl - yr)
            B = np.zeros((3, 2)); B[0] = h * (c * dv + xv * s_ * dw); B[1] = h * (s_ * dv - xv * c * dw); B[2] = h * dw
            d = rng.normal(0, 1e-3, 3)
            W = np.diag([10, 10, 0.5, 0.1, 0.1]) if trial % 2 == 0 else None
            if W is None:
                G = rng.normal(0, 1, (5, 5)); W = G @ G.T * 0.2 + np.diag([10, 10, 0.5, 0.1, 0.1])
            e = rng.normal(0, 0.2, 5)
            Q, S, R = W[:3, :3], W[:3, 3:], W[3:, 3:]
            q, r = W[:3] @ e, W[3:] @ e
            stages.append((A, B, d, Q, q, S, R, r))
            hs = {}
            for i in range(2):
                if rng.random() < 0.35: hs[i] = rng.uniform(-0.5, 0.5)
            held.append(hs)
        QN = np.diag([10, 10, 0.5]) * (1 + rng.random()); qN = rng.normal(0, 1, 3)
        Pr, pr = sequential(stages, QN, qN, held)
        for dt in (np.float32, np.float64):
            Ps, ps = scan(stages, QN, qN, held, dt)
            err = max(max(np.max(np.abs(Ps[k] - Pr[k])) / np.max(np.abs(Pr[k])) for k in range(N + 1)),
                      max(np.max(np.abs(ps[k] - pr[k])) / max(1e-9, np.max(np.abs(pr[k]))) for k in range(N + 1)))
            worst[dt] = max(worst[dt], err)
    print(f"400 random stage problems (N = 20, 35 % of the inputs held): worst relative deviation of (P_k, p_k), all k, from the "
          f"sequential float64 recursion: scan in float64 {worst[np.float64]:.2e}, scan in float32 {worst[np.float32]:.2e}")


if __name__ == "__main__":
    main()
