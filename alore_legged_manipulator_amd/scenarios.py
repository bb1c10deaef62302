"""Seeded synthetic NMPC batches ("planar class").

Every problem is one instance of the reference's planar ICR skid-steer NMPC
(NX=3, NU=2, NOD=3, NY=5, NYN=3; reference
planning_ddr_opt/nmpc_controller/UAV_CAR_model/UAV_CAR_model.cpp:29-46) laid out
exactly like the reference's ``ACADOvariables`` members
(.../quadrotor_mpc_codegen/acado_common.h:104-162), stacked over the batch:

    x  [B, N+1, 3]   u  [B, N, 2]    od [B, N+1, 3]   y  [B, N, 5]   yN [B, 3]
    W  [B, N, 5, 5]  WN [B, 3, 3]    x0 [B, 3]        lbValues/ubValues [B, N, 2]
    dual [B, N, 2]   (previous QP dual = acadoWorkspace.y, seeds the working set)

The reference trajectory is a constant-twist arc sampled the way
``MpcController::getRefPoints`` samples it (nmpc_controller/src/mpc.cpp:407-461:
node j gets t = (j+1)*dt; wheel speeds vr = v - w*yr, vl = v - w*yl, :442-443),
weights are the node defaults (config/mpc3ms.yaml:4-5) and the start iterate is
the cold start of ``MpcWrapper::solve`` (src/mpc_wrapper.cpp:267-275): x <- x0
replicated, u <- 0.
"""
from __future__ import annotations

import numpy as np

SEED = 20260206
DT = 0.01
# (xv, yr, yl) classes: launch default planner_sim.launch:41-43, EKF init :200-202
ICR_CLASSES = np.array([(0.0, -0.3, 0.3), (0.2, -0.3, 0.3), (0.1, -0.25, 0.25), (0.3, -0.35, 0.35)],
                       dtype=np.float64)
Q_DIAG = (10.0, 10.0, 0.5)
R_DIAG = (0.1, 0.1)
U_BOUND = 3.0  # UAV_CAR_model.cpp:97-101


def arc_pose(v, w, xv, t):
    """Exact flow of the ICR model for constant body twist (v, w) from (0,0,0)."""
    v, w, xv, t = np.broadcast_arrays(*(np.asarray(a, np.float64) for a in (v, w, xv, t)))
    th = w * t
    small = np.abs(w) < 1e-9
    ws = np.where(small, 1.0, w)
    s_over = np.where(small, t, np.sin(th) / ws)           # sin(wt)/w
    c_over = np.where(small, 0.0, (1.0 - np.cos(th)) / ws)  # (1-cos(wt))/w
    px = v * s_over + xv * (1.0 - np.cos(th))
    py = v * c_over - xv * np.sin(th)
    return np.stack([px, py, th], axis=-1)


def make_batch(B: int, N: int, seed: int = SEED, dt: float = DT, fast_tail: float = 0.05,
               offset: int = 0) -> dict:
    """Problems ``offset .. offset+B-1`` of the seeded stream (float32 arrays).

    Problem b depends only on (seed, b), so a rank that owns a slice of a large
    batch generates exactly the same problems as a single process would.
    """
    idx = np.arange(offset, offset + B)
    # counter-based: one independent generator per problem index
    draws = np.empty((B, 6), np.float64)
    for i, b in enumerate(idx):
        draws[i] = np.random.default_rng([seed, int(b)]).random(6)
    cls = idx % 4
    od1 = ICR_CLASSES[cls]                                   # [B,3]
    fast = draws[:, 5] < fast_tail
    v = np.where(fast, 2.5 + draws[:, 0] * 1.0, 0.3 + draws[:, 0] * 1.7)
    w = -1.5 + draws[:, 1] * 3.0
    t = (np.arange(N + 1) + 1.0) * dt                        # node j <- (j+1) dt
    pose = arc_pose(v[:, None], w[:, None], od1[:, None, 0], t[None, :])  # [B,N+1,3]
    vr = v - w * od1[:, 1]
    vl = v - w * od1[:, 2]
    y = np.concatenate([pose[:, :N, :], np.broadcast_to(np.stack([vr, vl], -1)[:, None, :], (B, N, 2))], -1)
    yN = pose[:, N, :]
    x0 = np.stack([-0.3 + 0.6 * draws[:, 2], -0.3 + 0.6 * draws[:, 3], -0.5 + 1.0 * draws[:, 4]], -1)
    W = np.zeros((B, N, 5, 5))
    W[:, :, np.arange(5), np.arange(5)] = np.array(Q_DIAG + R_DIAG)
    WN = np.zeros((B, 3, 3))
    WN[:, np.arange(3), np.arange(3)] = np.array(Q_DIAG)
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    return {
        "x": f32(np.broadcast_to(x0[:, None, :], (B, N + 1, 3))),
        "u": np.zeros((B, N, 2), np.float32),
        "od": f32(np.broadcast_to(od1[:, None, :], (B, N + 1, 3))),
        "y": f32(y), "yN": f32(yN), "W": f32(W), "WN": f32(WN), "x0": f32(x0),
        "lbValues": np.full((B, N, 2), -U_BOUND, np.float32),
        "ubValues": np.full((B, N, 2), U_BOUND, np.float32),
        "dual": np.zeros((B, N, 2), np.float32),
    }


def make_wide_batch(B: int, N: int, seed: int, dt: float = DT) -> dict:
    """Stress distribution (not the benchmark's): far-off initial poses, references beyond the wheel-speed bounds,
    random ICR geometry, log-uniform weights (a third of the problems with full symmetric positive definite
    blocks), random bounds.  Exercises long working-set iterations and the active-set safeguard."""
    r = np.random.default_rng(seed)
    xv = r.uniform(0.0, 0.4, B); half = r.uniform(0.15, 0.5, B); skew = r.uniform(-0.05, 0.05, B)
    od1 = np.stack([xv, -half + skew, half + skew], -1)
    v = r.uniform(-1.0, 4.5, B); w = r.uniform(-3.0, 3.0, B)
    t = (np.arange(N + 1) + 1.0) * dt
    pose = arc_pose(v[:, None], w[:, None], od1[:, None, 0], t[None, :])
    vr = v - w * od1[:, 1]; vl = v - w * od1[:, 2]
    y = np.concatenate([pose[:, :N, :], np.broadcast_to(np.stack([vr, vl], -1)[:, None, :], (B, N, 2))], -1)
    x0 = np.stack([r.uniform(-2, 2, B), r.uniform(-2, 2, B), r.uniform(-3, 3, B)], -1)
    qd = 10.0 ** r.uniform(-1, 2, (B, 3)); rd = 10.0 ** r.uniform(-2, 1, (B, 2))
    W = np.zeros((B, N, 5, 5)); WN = np.zeros((B, 3, 3))
    for i in range(3): W[:, :, i, i] = qd[:, None, i]; WN[:, i, i] = qd[:, i] * r.uniform(0.5, 5, B)
    for i in range(2): W[:, :, 3 + i, 3 + i] = rd[:, None, i]
    # a third of the problems: full symmetric positive definite weights
    full = r.random(B) < 0.33
    for b in np.nonzero(full)[0]:
        A = r.normal(size=(5, 5)) * 0.3
        M = A @ A.T + np.diag(np.concatenate([qd[b], rd[b]]))
        M[:3, 3:] = 0.0; M[3:, :3] = 0.0            # the generated code ignores state-control cross terms of the Hessian
        W[b, :] = M
        A3 = r.normal(size=(3, 3)); WN[b] = A3 @ A3.T + np.diag(qd[b])
    ub = r.uniform(0.5, 5.0, (B, 1, 2)) * np.ones((B, N, 2)); lb = -r.uniform(0.5, 5.0, (B, 1, 2)) * np.ones((B, N, 2))
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    return {"x": f32(np.broadcast_to(x0[:, None, :], (B, N + 1, 3))), "u": np.zeros((B, N, 2), np.float32),
            "od": f32(np.broadcast_to(od1[:, None, :], (B, N + 1, 3))), "y": f32(y), "yN": f32(pose[:, N, :]),
            "W": f32(W), "WN": f32(WN), "x0": f32(x0), "lbValues": f32(lb), "ubValues": f32(ub),
            "dual": np.zeros((B, N, 2), np.float32)}


def problem(batch: dict, b: int) -> dict:
    """Problem b of a batch as flat reference-layout arrays."""
    return {k: np.ascontiguousarray(v[b]).reshape(-1) for k, v in batch.items()}
