"""Generates tests/golden/traj_polynomes.npz: planner messages (carstatemsgs/Polynome) that are NOT constant-twist
arcs, with answers computed independently of every C / C++ / HIP file in this repository:

  * the quintic coefficients of the minimum-jerk spline from the dense 6M x 6M system of the reference
    (planning_ddr_opt/back_end/include/gcopter/minco.hpp:817-898), written out in NumPy and solved with numpy.linalg;
  * the pose (x, y) at sample times by adaptive quadrature (scipy.integrate.quad, piecewise) of
        x' = s' cos(theta) + theta' xv sin(theta),   y' = s' sin(theta) - theta' xv cos(theta)
    (nmpc_controller/include/nmpc_controller/traj_anal.hpp:55-130) -- the EXACT integral, so a Simpson-based
    implementation is expected to differ by its quadrature error (bounded in the tests);
  * jerk and snap at both sides of every junction (the C3 / C4 rows that define MINCO_S3NU).

Run:  python oracle/gen_traj_golden.py        (NumPy + SciPy only; deterministic)
"""
import os
import sys

import numpy as np
from scipy.integrate import quad

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "traj_polynomes.npz")
MAXP = 12


def dense_system(T, inner, init_pva, tail_pva):
    M = len(T)
    A = np.zeros((6 * M, 6 * M)); b = np.zeros((6 * M, 2))
    A[0, 0] = 1.0; A[1, 1] = 1.0; A[2, 2] = 2.0
    b[0], b[1], b[2] = init_pva[0:2], init_pva[2:4], init_pva[4:6]
    for i in range(M - 1):
        t = T[i]; r = 6 * i
        A[r + 3, r + 3:r + 6] = [6.0, 24.0 * t, 60.0 * t * t]; A[r + 3, r + 9] = -6.0
        A[r + 4, r + 4:r + 6] = [24.0, 120.0 * t]; A[r + 4, r + 10] = -24.0
        A[r + 5, r:r + 6] = [t ** k for k in range(6)]
        A[r + 6, r:r + 6] = [t ** k for k in range(6)]; A[r + 6, r + 6] = -1.0
        A[r + 7, r + 1:r + 6] = [k * t ** (k - 1) for k in range(1, 6)]; A[r + 7, r + 7] = -1.0
        A[r + 8, r + 2:r + 6] = [k * (k - 1) * t ** (k - 2) for k in range(2, 6)]; A[r + 8, r + 8] = -2.0
        b[r + 5] = inner[i]
    t = T[-1]; e = 6 * M
    A[e - 3, e - 6:e] = [t ** k for k in range(6)]
    A[e - 2, e - 5:e] = [k * t ** (k - 1) for k in range(1, 6)]
    A[e - 1, e - 4:e] = [k * (k - 1) * t ** (k - 2) for k in range(2, 6)]
    b[e - 3], b[e - 2], b[e - 1] = tail_pva[0:2], tail_pva[2:4], tail_pva[4:6]
    return A, b


def deriv(c, t, order):
    out = np.zeros(2)
    for k in range(order, 6):
        f = 1.0
        for q in range(order):
            f *= (k - q)
        out += f * c[k] * t ** (k - order)
    return out


def main():
    rng = np.random.default_rng(20260206)
    n = 10
    msgs = []
    for i in range(n):
        M = int(rng.integers(1, MAXP + 1)) if i else MAXP
        T = rng.uniform(0.3, 0.7, M)
        th = np.cumsum(rng.uniform(-0.6, 0.6, M + 1)) + rng.uniform(-3, 3)
        sl = np.cumsum(rng.uniform(0.1, 0.8, M + 1))
        inner = np.stack([th[1:M], sl[1:M]], 1)
        init = np.array([th[0], sl[0], rng.uniform(-1, 1), rng.uniform(0.2, 1.5), rng.uniform(-1, 1), rng.uniform(-1, 1)])
        tail = np.array([th[M], sl[M], rng.uniform(-1, 1), rng.uniform(0.2, 1.5), rng.uniform(-1, 1), rng.uniform(-1, 1)])
        start = rng.uniform(-1, 1, 3)
        icr = np.array([-0.3, 0.3, rng.uniform(0, 0.3)])
        msgs.append(dict(M=M, T=T, inner=inner, init=init, tail=tail, start=start, icr=icr, t0=float(rng.uniform(0, 0.05))))
    out = {"n": n, "max_pieces": MAXP}
    S = 25  # sample times per message
    for i, m in enumerate(msgs):
        M = m["M"]
        A, b = dense_system(m["T"], m["inner"], m["init"], m["tail"])
        assert np.linalg.cond(A) < 1e12
        c = np.linalg.solve(A, b).reshape(M, 6, 2)          # [piece][power][dim]
        edges = np.concatenate([[0.0], np.cumsum(m["T"])])
        xv = m["icr"][2]

        def integrand(tt, k, which):
            p = deriv(c[k], tt, 0); v = deriv(c[k], tt, 1)
            if which == 0:
                return v[1] * np.cos(p[0]) + v[0] * xv * np.sin(p[0])
            return v[1] * np.sin(p[0]) - v[0] * xv * np.cos(p[0])

        times = np.sort(np.concatenate([[0.0, edges[-1]], rng.uniform(0, edges[-1], S - 2)]))
        xy = np.zeros((S, 2)); flat = np.zeros((S, 3, 2))
        piece_int = np.zeros((M, 2))
        for k in range(M):
            for w in range(2):
                piece_int[k, w] = quad(integrand, 0.0, m["T"][k], args=(k, w), epsabs=1e-13, epsrel=1e-13)[0]
        for j, t in enumerate(times):
            k = min(int(np.searchsorted(edges, t, side="right")) - 1, M - 1)
            tl = t - edges[k]
            for w in range(2):
                xy[j, w] = m["start"][w] + piece_int[:k, w].sum() + quad(integrand, 0.0, tl, args=(k, w), epsabs=1e-13, epsrel=1e-13)[0]
            for o in range(3):
                flat[j, o] = deriv(c[k], tl, o)
        junction = np.zeros((max(M - 1, 1), 2, 2, 2))       # [junction][jerk|snap][left|right][dim]
        for k in range(M - 1):
            for o, order in enumerate((3, 4)):
                junction[k, o, 0] = deriv(c[k], m["T"][k], order)
                junction[k, o, 1] = deriv(c[k + 1], 0.0, order)
            assert np.allclose(junction[k, :, 0], junction[k, :, 1], rtol=1e-7, atol=1e-7)
        pre = f"m{i}_"
        out.update({pre + "T": m["T"], pre + "inner": m["inner"], pre + "init": m["init"], pre + "tail": m["tail"],
                    pre + "start": m["start"], pre + "icr": m["icr"], pre + "t0": m["t0"], pre + "coef": c,
                    pre + "times": times, pre + "xy": xy, pre + "flat": flat, pre + "junction": junction})
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    sys.exit(main())
