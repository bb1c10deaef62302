"""The kernels' rigid-body source (csrc/wb_dynamics.h: 3-D Newton-Euler, chain by chain) compiled for the host and
compared with the spatial-algebra oracle -- two formulations of the same physics, 1e-11."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle.wb_oracle import Model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "harness", "wb_dynamics_harness.cpp")
SO = os.path.join(ROOT, "tests", "harness", "libwb_dynamics_harness.so")
DP = C.POINTER(C.c_double)


@pytest.fixture(scope="module")
def H():
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", SO, SRC])
    return C.CDLL(SO)


def dp(a):
    return a.ctypes.data_as(DP)


def test_rnea_matches_the_spatial_oracle(H):
    m = Model()
    rng = np.random.default_rng(8)
    for _ in range(20):
        q = np.concatenate([rng.uniform(-1, 1, 3), rng.uniform(-0.8, 0.8, 3), rng.uniform(m.lower, m.upper)])
        v, a, f = rng.normal(0, 1, 24), rng.normal(0, 2, 24), rng.normal(0, 80, 12)
        tau = np.zeros(24)
        H.wbh_rnea(dp(q), dp(v), dp(a), dp(f), C.c_double(m.g), dp(tau))
        ref = m.rnea(q, v, a, f.reshape(4, 3))
        assert np.max(np.abs(tau - ref)) < 1e-11 * max(1.0, np.max(np.abs(ref)))
        H.wbh_rnea(dp(q), dp(v), dp(a), dp(f), C.c_double(0.0), dp(tau))
        assert np.max(np.abs(tau - m.rnea(q, v, a, f.reshape(4, 3), gravity=False))) < 1e-10


def test_mass_matrix_columns(H):
    m = Model()
    rng = np.random.default_rng(9)
    q = np.concatenate([rng.uniform(-1, 1, 3), rng.uniform(-0.8, 0.8, 3), rng.uniform(m.lower, m.upper)])
    M = m.crba(q)
    for j in range(24):
        col = np.zeros(24)
        H.wbh_mass_column(dp(q), j, dp(col))
        assert np.max(np.abs(col - M[:, j])) < 1e-11


def test_aba_matches_the_spatial_oracle(H):
    """csrc/wb_aba.h (the GPU's articulated-body algorithm, 3 x 3 block form) against the oracle's 6-D ABA and against
    M^-1 (tau - bias): three routes to the same accelerations."""
    m = Model()
    rng = np.random.default_rng(10)
    for _ in range(20):
        q = np.concatenate([rng.uniform(-1, 1, 3), rng.uniform(-0.8, 0.8, 3), rng.uniform(m.lower, m.upper)])
        v, tau, f = rng.normal(0, 1, 24), rng.normal(0, 8, 18), rng.normal(0, 80, 12)
        acc = np.zeros(24)
        H.wbh_aba(dp(q), dp(v), dp(tau), dp(f), C.c_double(m.g), dp(acc))
        ref = m.aba(q, v, tau, f.reshape(4, 3))
        assert np.max(np.abs(acc - ref)) < 1e-9 * max(1.0, np.max(np.abs(ref))), np.max(np.abs(acc - ref))
        assert np.max(np.abs(acc - m.forward_dynamics(q, v, tau, f.reshape(4, 3)))) < 1e-8 * max(1.0, np.max(np.abs(ref)))
