"""csrc/minco_spline.h compiled for the host: the knot-state formulation the kernels use (SPD 2 x 2 block-tridiagonal
system + Hermite pieces) against the oracle's restatement of the reference's 6M x 6M band LU and against a dense NumPy
solve; Hermite map derivatives by finite differences."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tests.test_backend_oracle import dense_minco_system, random_spline

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "harness", "minco_spline_harness.cpp")
SO = os.path.join(ROOT, "tests", "harness", "libminco_spline_harness.so")
DP = C.POINTER(C.c_double)


@pytest.fixture(scope="module")
def H():
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", SO, SRC])
    L = C.CDLL(SO)
    L.harness_spline.argtypes = [C.c_int, DP, DP, DP, DP, DP]
    L.harness_hermite.argtypes = [C.c_double, DP, DP, DP, DP]
    L.harness_hermite_adjoint.argtypes = [C.c_double, DP, DP, DP]
    return L


def dp(a):
    return a.ctypes.data_as(DP)


@pytest.mark.parametrize("M", [1, 2, 3, 5, 12, 16, 32, 64])
def test_knot_state_spline_equals_band_lu_and_dense_solve(H, M):
    from oracle.backend_driver import BackendOracle
    rng = np.random.default_rng(40 + M)
    T, inner, head, tail = random_spline(rng, M)
    coef = np.zeros((6 * M, 2))
    H.harness_spline(M, dp(T), dp(np.ascontiguousarray(inner if M > 1 else np.zeros((1, 2)))), dp(np.ascontiguousarray(head)),
                     dp(np.ascontiguousarray(tail)), dp(coef))
    ref = BackendOracle().spline(T, inner, head, tail)
    A, b = dense_minco_system(T, inner, head, tail)
    dense = np.linalg.solve(A, b)
    scale = max(1.0, np.max(np.abs(dense)))
    assert np.max(np.abs(coef - dense)) <= 1e-9 * scale
    assert np.max(np.abs(coef - ref)) <= 1e-9 * scale
    # the knot-state form is the better conditioned of the two: it is at least as close to the dense solve
    assert np.max(np.abs(coef - dense)) <= 10 * np.max(np.abs(ref - dense)) + 1e-13 * scale


def test_hermite_derivatives(H):
    rng = np.random.default_rng(2)
    for _ in range(20):
        T = rng.uniform(0.2, 2.0)
        z0, z1 = rng.normal(size=3), rng.normal(size=3)
        c = np.zeros(6); dc = np.zeros(6)
        H.harness_hermite(T, dp(z0), dp(z1), dp(c), dp(dc))
        # interpolation conditions
        t = T
        p = sum(c[k] * t ** k for k in range(6)); v = sum(k * c[k] * t ** (k - 1) for k in range(1, 6))
        a = sum(k * (k - 1) * c[k] * t ** (k - 2) for k in range(2, 6))
        assert np.allclose([c[0], c[1], 2 * c[2], p, v, a], np.concatenate([z0, z1]), rtol=1e-10, atol=1e-10)
        # d c / d T by central differences
        h = 1e-6
        cp = np.zeros(6); cm = np.zeros(6); junk = np.zeros(6)
        H.harness_hermite(T + h, dp(z0), dp(z1), dp(cp), dp(junk))
        H.harness_hermite(T - h, dp(z0), dp(z1), dp(cm), dp(junk))
        assert np.max(np.abs((cp - cm) / (2 * h) - dc)) <= 1e-6 * max(1.0, np.max(np.abs(dc)))
        # adjoint = transpose of the (linear) map z -> c
        G = rng.normal(size=6); g0 = np.zeros(3); g1 = np.zeros(3)
        H.harness_hermite_adjoint(T, dp(G), dp(g0), dp(g1))
        J = np.zeros((6, 6))
        for j in range(6):
            e = np.zeros(6); e[j] = 1.0
            cj = np.zeros(6)
            H.harness_hermite(T, dp(e[:3].copy()), dp(e[3:].copy()), dp(cj), dp(junk))
            J[:, j] = cj
        assert np.allclose(J.T @ G, np.concatenate([g0, g1]), rtol=1e-10, atol=1e-10)
