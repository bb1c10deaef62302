"""Pins of the float64 back_end oracle (oracle/backend_oracle.c).  The reference's back_end cannot be
compiled here (Eigen / ROS / PCL absent) and has no tests: parity unpinned against the binary; these are the
self-checks SURVEY.md section 8(c) prescribes, each independent of the C code it checks."""
import numpy as np
import pytest
from scipy.integrate import quad

from alore_legged_manipulator_amd.flat_traj import monte_carlo_goals, straight_goal, waypoint_path
from oracle.backend_driver import BackendOracle, EsdfGrid


@pytest.fixture(scope="module")
def orc():
    return BackendOracle()


def dense_minco_system(T, inner, head, tail):
    """The 6M x 6M minimum-jerk system written out densely in NumPy from the reference's row list
    (back_end/include/gcopter/minco.hpp:829-892); solved by numpy.linalg, not by a banded LU."""
    M = len(T)
    A = np.zeros((6 * M, 6 * M))
    b = np.zeros((6 * M, 2))
    A[0, 0] = 1.0; A[1, 1] = 1.0; A[2, 2] = 2.0
    b[0], b[1], b[2] = head[:, 0], head[:, 1], head[:, 2]
    pw = lambda t, k: t ** k
    for i in range(M - 1):
        t = T[i]; r = 6 * i
        A[r + 3, r + 3:r + 6] = [6.0, 24.0 * t, 60.0 * t * t]; A[r + 3, r + 9] = -6.0
        A[r + 4, r + 4:r + 6] = [24.0, 120.0 * t]; A[r + 4, r + 10] = -24.0
        A[r + 5, r:r + 6] = [pw(t, k) for k in range(6)]
        A[r + 6, r:r + 6] = [pw(t, k) for k in range(6)]; A[r + 6, r + 6] = -1.0
        A[r + 7, r + 1:r + 6] = [k * pw(t, k - 1) for k in range(1, 6)]; A[r + 7, r + 7] = -1.0
        A[r + 8, r + 2:r + 6] = [k * (k - 1) * pw(t, k - 2) for k in range(2, 6)]; A[r + 8, r + 8] = -2.0
        b[r + 5] = inner[i]
    t = T[-1]; e = 6 * M
    A[e - 3, e - 6:e] = [pw(t, k) for k in range(6)]
    A[e - 2, e - 5:e] = [k * pw(t, k - 1) for k in range(1, 6)]
    A[e - 1, e - 4:e] = [k * (k - 1) * pw(t, k - 2) for k in range(2, 6)]
    b[e - 3], b[e - 2], b[e - 1] = tail[:, 0], tail[:, 1], tail[:, 2]
    return A, b


def random_spline(rng, M):
    T = rng.uniform(0.25, 1.6, M)
    inner = np.cumsum(rng.normal(0.4, 0.6, (M - 1, 2)), axis=0)
    head = rng.normal(0, 1, (2, 3)); tail = rng.normal(0, 1, (2, 3))
    tail[:, 0] += inner[-1] if M > 1 else 0
    return T, inner, head, tail


def poly_derivs(c, t, order):
    """derivative `order` of the quintic with ascending coefficients c (6, 2) at t"""
    out = np.zeros(2)
    for k in range(order, 6):
        f = 1.0
        for q in range(order):
            f *= (k - q)
        out += f * c[k] * t ** (k - order)
    return out


@pytest.mark.parametrize("M", [1, 2, 3, 7, 16, 40])
def test_spline_matches_dense_solve_and_is_c4(orc, M):
    rng = np.random.default_rng(100 + M)
    T, inner, head, tail = random_spline(rng, M)
    coef = orc.spline(T, inner, head, tail)
    A, b = dense_minco_system(T, inner, head, tail)
    ref = np.linalg.solve(A, b)
    assert np.max(np.abs(coef - ref)) <= 1e-9 * max(1.0, np.max(np.abs(ref)))
    c = coef.reshape(M, 6, 2)
    for d in range(3):  # boundary conditions
        assert np.allclose(poly_derivs(c[0], 0.0, d), head[:, d], atol=1e-10)
        assert np.allclose(poly_derivs(c[-1], T[-1], d), tail[:, d], atol=1e-8)
    for i in range(M - 1):  # C0..C4 and interpolation
        assert np.allclose(poly_derivs(c[i], T[i], 0), inner[i], atol=1e-9)
        for d in range(5):
            lhs, rhs = poly_derivs(c[i], T[i], d), poly_derivs(c[i + 1], 0.0, d)
            assert np.allclose(lhs, rhs, atol=1e-7 * max(1.0, np.max(np.abs(rhs)))), (i, d)


def test_energy_is_weighted_squared_jerk(orc):
    rng = np.random.default_rng(5)
    T, inner, head, tail = random_spline(rng, 5)
    coef = orc.spline(T, inner, head, tail)
    w = (0.33, 1.0)
    e, gdC, gdT = orc.energy(T, coef, w)
    c = coef.reshape(5, 6, 2)
    ref = 0.0
    for i in range(5):
        ref += quad(lambda t: float(np.sum(np.array(w) * poly_derivs(c[i], t, 3) ** 2)), 0.0, T[i], epsabs=1e-12, epsrel=1e-12)[0]
    assert abs(e - ref) <= 1e-9 * abs(ref)
    # partial gradients by finite differences of the closed form
    for idx in [(3, 0), (4, 1), (5 + 6 * 2, 0), (29, 1)]:
        h = 1e-6
        cp = coef.copy(); cp[idx] += h; cm = coef.copy(); cm[idx] -= h
        fd = (orc.energy(T, cp, w)[0] - orc.energy(T, cm, w)[0]) / (2 * h)
        assert abs(fd - gdC[idx]) <= 1e-6 * max(1.0, abs(fd))
    for i in range(5):
        h = 1e-6
        Tp = T.copy(); Tp[i] += h; Tm = T.copy(); Tm[i] -= h
        fd = (orc.energy(Tp, coef, w)[0] - orc.energy(Tm, coef, w)[0]) / (2 * h)
        assert abs(fd - gdT[i]) <= 1e-6 * max(1.0, abs(fd))


def test_adjoint_matches_dense_transpose_solve_and_fd(orc):
    rng = np.random.default_rng(9)
    M = 6
    T, inner, head, tail = random_spline(rng, M)
    coef = orc.spline(T, inner, head, tail)
    G = rng.normal(0, 1, (6 * M, 2))  # dJ/dcoef of the linear functional J = <G, coef>
    gp, gt, gtail = orc.adjoint(T, coef, G, np.zeros(M))
    A, _ = dense_minco_system(T, inner, head, tail)
    lam = np.linalg.solve(A.T, G)
    assert np.allclose(gp.reshape(-1, 2), lam[[6 * i + 5 for i in range(M - 1)]], rtol=1e-9, atol=1e-9)
    assert np.allclose(gtail, lam[6 * M - 3], rtol=1e-9, atol=1e-9)
    J = lambda T_, inner_, tail_: float(np.sum(G * orc.spline(T_, inner_, head, tail_)))
    for i in range(M):
        h = 1e-6
        Tp = T.copy(); Tp[i] += h; Tm = T.copy(); Tm[i] -= h
        fd = (J(Tp, inner, tail) - J(Tm, inner, tail)) / (2 * h)
        assert abs(fd - gt[i]) <= 2e-6 * max(1.0, abs(fd)), i
    for i in range(M - 1):
        for d in range(2):
            h = 1e-6
            ip = inner.copy(); ip[i, d] += h; im = inner.copy(); im[i, d] -= h
            fd = (J(T, ip, tail) - J(T, im, tail)) / (2 * h)
            assert abs(fd - gp[2 * i + d]) <= 1e-6 * max(1.0, abs(fd))
    h = 1e-6
    tp = tail.copy(); tp[1, 0] += h; tm = tail.copy(); tm[1, 0] -= h
    assert abs((J(T, inner, tp) - J(T, inner, tm)) / (2 * h) - gtail[1]) <= 1e-6 * max(1.0, abs(gtail[1]))


def circle_grid(cx, cy, r, half=12.0, res=0.1):
    return EsdfGrid.from_field(lambda X, Y: np.hypot(X - cx, Y - cy) - r, half=half, res=res)


def test_esdf_bilinear_against_analytic_field(orc):
    g = circle_grid(1.0, -0.5, 0.8)
    rng = np.random.default_rng(3)
    # exact at cell centres, second-order in between, analytic gradient direction
    assert abs(orc.esdf(g, 2.05, 1.05)[0] - (np.hypot(1.05, 1.55) - 0.8)) < 1e-12
    for _ in range(200):
        x, y = rng.uniform(-8, 8, 2)
        if np.hypot(x - 1.0, y + 0.5) < 0.5:
            continue
        d, gr = orc.esdf(g, x, y, mode=0)
        exact = np.hypot(x - 1.0, y + 0.5) - 0.8
        assert abs(d - exact) < 0.1 ** 2 / (2 * 0.5)  # |f''| h^2 / 8 per axis, curvature <= 1 / 0.5
        n = np.array([x - 1.0, y + 0.5]) / np.hypot(x - 1.0, y + 0.5)
        assert np.linalg.norm(gr - n) < 0.25
        # the gradient is the exact derivative of the interpolant (inside one cell)
        h = 1e-7
        fx = (orc.esdf(g, x + h, y)[0] - orc.esdf(g, x - h, y)[0]) / (2 * h)
        fy = (orc.esdf(g, x, y + h)[0] - orc.esdf(g, x, y - h)[0]) / (2 * h)
        cell = lambda v: np.floor((v + 12.0) / 0.1 - 0.5)
        if cell(x + h) == cell(x - h) and cell(y + h) == cell(y - h):
            assert abs(fx - gr[0]) < 1e-6 and abs(fy - gr[1]) < 1e-6
    # outside the map: 100 / 1e10 / 1e10 and a zero gradient (sdf_map.cpp:761-765, 797-801, 838-840)
    assert orc.esdf(g, 13.0, 0.0, mode=0)[0] == 100.0
    assert orc.esdf(g, 13.0, 0.0, mode=1, mindis=0.6)[0] == 1e10
    assert orc.esdf(g, 0.0, -12.5, mode=2)[0] == 1e10
    # last cell row has no upper neighbour: treated as outside (idx >= size - 1)
    assert orc.esdf(g, 11.97, 0.0, mode=0)[0] == 100.0
    # mode 1 leaves the gradient alone beyond mindis
    d, gr = orc.esdf(g, 5.0, 5.0, mode=1, mindis=0.6)
    assert d > 0.6 and np.all(gr == 0.0)


def fd_grad(orc, grid, ft, stage, x, **kw):
    g = np.zeros_like(x)
    for i in range(len(x)):
        h = 1e-6 * max(1.0, abs(x[i]))
        xp = x.copy(); xp[i] += h; xm = x.copy(); xm[i] -= h
        g[i] = (orc.eval(grid, ft, stage, xp, **kw)["cost"] - orc.eval(grid, ft, stage, xm, **kw)["cost"]) / (2 * h)
    return g


@pytest.mark.parametrize("stage", [1, 2])
@pytest.mark.parametrize("standard_diff", [1, 0])
def test_cost_gradient_matches_finite_differences_free_map(orc, stage, standard_diff):
    orc.cfg.standard_diff = standard_diff
    # ICR model: the reference's d(dy)/d(yaw) has a sign slip (oracle header); the exact form is what FD can pin
    orc.cfg.exact_chain = 0 if standard_diff else 1
    try:
        grid = EsdfGrid.free(half=20.0)
        rng = np.random.default_rng(11 + stage)
        for ft in monte_carlo_goals(3, seed=77):
            x = orc.x0(ft) + rng.normal(0, 0.05, 3 * ft.pieces - 1)
            # stage 1 charges two different time weights in cost and gradient (reference quirk): make them equal
            kw = dict(time_weight=orc.cfg.p_time) if stage == 1 else dict(lam=(3.0, -2.0), rho=(5e3, 2e4))
            r = orc.eval(grid, ft, stage, x, **kw)
            fd = fd_grad(orc, grid, ft, stage, x, **kw)
            assert np.max(np.abs(fd - r["grad"])) <= 1e-6 * np.max(np.abs(fd)), (stage, standard_diff)
            if not standard_diff:  # the reference's form differs, only in the yaw rows, by the documented term
                orc.cfg.exact_chain = 0
                ref = orc.eval(grid, ft, stage, x, **kw)
                orc.cfg.exact_chain = 1
                assert ref["cost"] == r["cost"]
                M = ft.pieces
                dev = np.abs(ref["grad"] - r["grad"])
                assert 0 < dev.max() < np.abs(r["grad"]).max()
    finally:
        orc.cfg.standard_diff = 1
        orc.cfg.exact_chain = 0


def obstacle_case():
    # a path that brushes a disc: clearance violated on some nodes, not on all
    ft = waypoint_path([[0.0, 0.0], [6.0, 0.5]], 0.1, 0.4)
    grid = circle_grid(3.0, 1.3, 0.9)
    return ft, grid


def test_obstacle_gradient_exact_chain_matches_fd_and_reference_chain_is_close(orc):
    ft, grid = obstacle_case()
    x = orc.x0(ft)
    kw = dict(lam=(0.0, 0.0), rho=(1e4, 1e4))
    base = orc.eval(grid, ft, 2, x, **kw)
    free = orc.eval(EsdfGrid.free(half=12.0), ft, 2, x, **kw)
    assert base["cost"] > free["cost"] + 1.0  # the obstacle term is active
    orc.cfg.exact_chain = 1
    try:
        r = orc.eval(grid, ft, 2, x, **kw)
        fd = fd_grad(orc, grid, ft, 2, x, **kw)
    finally:
        orc.cfg.exact_chain = 0
    assert r["cost"] == base["cost"]
    # the smoothed-L1 / bilinear field is C1 only: central differences see the kinks at 1e-4
    assert np.max(np.abs(fd - r["grad"])) <= 2e-4 * np.max(np.abs(fd))
    # the reference's chain (node counted with its full Simpson weight) is an O(step) perturbation of that
    dev = np.max(np.abs(base["grad"] - r["grad"])) / np.max(np.abs(r["grad"]))
    assert 0.0 < dev < 0.05


def test_simpson_positions_against_quadrature(orc):
    orc.cfg.standard_diff = 0
    try:
        ft = straight_goal((0.3, -0.2, 0.5), (4.0, 3.0, -0.7))
        grid = EsdfGrid.free(half=12.0)
        x = orc.x0(ft)
        r = orc.eval(grid, ft, 2, x, want_nodes=True)
        M = ft.pieces
        tau = x[2 * (M - 1) + 1:]
        T = np.where(tau > 0, (0.5 * tau + 1) * tau + 1, 1 / ((0.5 * tau - 1) * tau + 1))
        tail = ft.final_state.copy(); tail[1, 0] = x[2 * (M - 1)]
        coef = orc.spline(T, x[:2 * (M - 1)].reshape(-1, 2), ft.start_state, tail).reshape(M, 6, 2)
        xv = orc.cfg.icr_xv
        pos = np.array(ft.start_xytheta[:2], dtype=float)
        S = 2 * orc.cfg.sparse_res
        for i in range(M):
            fx = lambda t: poly_derivs(coef[i], t, 1)[1] * np.cos(poly_derivs(coef[i], t, 0)[0]) + poly_derivs(coef[i], t, 1)[0] * xv * np.sin(poly_derivs(coef[i], t, 0)[0])
            fy = lambda t: poly_derivs(coef[i], t, 1)[1] * np.sin(poly_derivs(coef[i], t, 0)[0]) - poly_derivs(coef[i], t, 1)[0] * xv * np.cos(poly_derivs(coef[i], t, 0)[0])
            for j in range(0, S + 1, 2):
                t = T[i] * j / S
                ex = pos + np.array([quad(fx, 0, t, epsabs=1e-12)[0], quad(fy, 0, t, epsabs=1e-12)[0]])
                assert np.max(np.abs(r["nodes"][i * (S + 1) + j] - ex)) < 2e-6, (i, j)  # Simpson, 8 panels of <= 0.08 s
            pos = pos + np.array([quad(fx, 0, T[i], epsabs=1e-12)[0], quad(fy, 0, T[i], epsabs=1e-12)[0]])
        assert np.max(np.abs(pos - ft.final_xytheta[:2] - r["xy_err"])) < 1e-5
    finally:
        orc.cfg.standard_diff = 1


def test_path_points_against_quadrature_and_the_predicted_state(orc):
    """be_path_points (MSPlanner::mincoPointPub): the marker points are the integral of the planar velocity -- SciPy
    quadrature per panel, both ICR models -- every piece ends with a doubled point, and the last point is what
    be_predicted_state reaches at the end of the plan when it integrates with the same panel width."""
    from oracle.backend_driver import path_points, predicted_state
    for standard, xv in ((True, 0.0), (False, 0.2)):
        ft = straight_goal((0.3, -0.2, 0.5), (4.0, 3.0, -0.7))
        x = orc.x0(ft)
        M = ft.pieces
        tau = x[2 * (M - 1) + 1:]
        T = np.where(tau > 0, (0.5 * tau + 1) * tau + 1, 1 / ((0.5 * tau - 1) * tau + 1))
        tail = ft.final_state.copy(); tail[1, 0] = x[2 * (M - 1)]
        coef = orc.spline(T, x[:2 * (M - 1)].reshape(-1, 2), ft.start_state, tail).reshape(M, 6, 2)
        res = 3
        xy, yaw = path_points(T, coef.reshape(-1), res, ft.start_xytheta[:2], standard_diff=standard, xv=xv)
        assert xy.shape == (M * (res + 1), 2) and yaw.shape == (M * res,)
        pos = np.array(ft.start_xytheta[:2], dtype=float)
        k = 0
        for i in range(M):
            fx = lambda t: poly_derivs(coef[i], t, 1)[1] * np.cos(poly_derivs(coef[i], t, 0)[0]) + poly_derivs(coef[i], t, 1)[0] * xv * np.sin(poly_derivs(coef[i], t, 0)[0])
            fy = lambda t: poly_derivs(coef[i], t, 1)[1] * np.sin(poly_derivs(coef[i], t, 0)[0]) - poly_derivs(coef[i], t, 1)[0] * xv * np.cos(poly_derivs(coef[i], t, 0)[0])
            for j in range(1, res + 1):
                t = T[i] * j / res
                ex = pos + np.array([quad(fx, 0, t, epsabs=1e-12)[0], quad(fy, 0, t, epsabs=1e-12)[0]])
                assert np.max(np.abs(xy[k] - ex)) < 5e-4, (i, j)                       # Simpson, panels of ~0.2 s
                assert abs(yaw[i * res + j - 1] - poly_derivs(coef[i], t, 0)[0]) < 1e-12
                k += 1
            assert np.array_equal(xy[k], xy[k - 1])                                    # the doubled end point of the piece
            k += 1
            pos = pos + np.array([quad(fx, 0, T[i], epsabs=1e-12)[0], quad(fy, 0, T[i], epsabs=1e-12)[0]])
        # equal pieces: one panel width for the whole plan, which get_the_predicted_state then walks as well
        if np.allclose(T, T[0]):
            xyt, _, _, _ = predicted_state(T, coef.reshape(-1), T[0] / res, float(T.sum()), start_xytheta=ft.start_xytheta, standard_diff=standard, xv=xv)
            assert np.max(np.abs(xyt[:2] - xy[-1])) < 1e-9


def test_time_map_and_quirks(orc):
    ft = straight_goal((0, 0, 0), (4, 0, 0))
    grid = EsdfGrid.free(half=12.0)
    x = orc.x0(ft)
    # |x| > 1e4: cost 0 (the reference's `inf` macro is 1 >> 30) and an untouched gradient
    big = x.copy(); big[0] = 2e4
    r = orc.eval(grid, ft, 2, big)
    assert r["cost"] == 0.0 and np.all(r["grad"] == 0.0)
    # stage 1: cost uses PathpenaltyWeights.time_weight, the gradient penaltyWeights.time_weight
    a = orc.eval(grid, ft, 1, x, time_weight=50.0)
    b = orc.eval(grid, ft, 1, x, time_weight=20.0)
    assert a["cost"] == b["cost"] and not np.allclose(a["grad"], b["grad"])


def test_optimiser_converges_and_respects_limits(orc):
    grid = EsdfGrid.free(half=20.0)
    for ft in monte_carlo_goals(12, seed=5):
        r = orc.minco_plan(grid, ft)
        assert r["ok"] and r["attempts"] == 1
        assert np.hypot(*r["xy_err"]) < orc.cfg.tol
        assert r["lbfgs_ret"] in (0, 1) and r["path_ret"] in (0, 1)
        x0 = orc.x0(ft)
        assert r["cost"] < orc.eval(grid, ft, 2, x0)["cost"]
        c = r["coef"].reshape(ft.pieces, 6, 2)
        for i in range(ft.pieces):
            for t in np.linspace(0, r["T"][i], 9):
                v = poly_derivs(c[i], t, 1); a = poly_derivs(c[i], t, 2)
                # soft constraints (smoothed-L1 penalties): small violations are allowed, gross ones are not
                assert abs(a[1]) < orc.cfg.max_acc * 1.15 and abs(a[0]) < orc.cfg.max_domega * 1.15
                assert orc.cfg.max_vel * abs(v[0]) + orc.cfg.max_omega * v[1] < orc.cfg.max_vel * orc.cfg.max_omega * 1.1


def test_obstacle_is_avoided(orc):
    ft, grid = obstacle_case()
    r = orc.minco_plan(grid, ft)
    assert r["ok"]
    assert r["min_dist"] > orc.cfg.final_min_safe_dis
    x = np.concatenate([r["inner"].reshape(-1), [r["tail_s"]], [np.sqrt(2 * t - 1) - 1 if t > 1 else 1 - np.sqrt(2 / t - 1) for t in r["T"]]])
    nodes = orc.eval(grid, ft, 2, x, want_nodes=True)["nodes"][::2]
    # body check points keep (almost) safe_dis from the disc; the penalty is soft
    assert np.min(np.hypot(nodes[:, 0] - 3.0, nodes[:, 1] - 1.3) - 0.9) > orc.cfg.safe_dis - 0.35


def test_optimiser_is_sensitive_to_the_last_digits(orc):
    """Documents why whole plans cannot be compared digit for digit with ANY other implementation (another
    compiler, another libm, the GPU): perturbing the inputs in the 14th digit changes the iteration on which the
    reference's stopping rule fires, and with it the final cost by 1e-3..5e-2."""
    import copy
    grid = EsdfGrid.free(half=20.0)
    rng = np.random.default_rng(0)
    dev, same = [], 0
    for ft in monte_carlo_goals(24, seed=20260206):
        a = orc.minco_plan(grid, ft)
        f2 = copy.deepcopy(ft)
        f2.traj_pts = ft.traj_pts * (1 + 1e-14 * rng.standard_normal(ft.traj_pts.shape))
        b = orc.minco_plan(grid, f2)
        dev.append(abs(a["cost"] - b["cost"]) / abs(a["cost"]))
        same += a["evals"] == b["evals"]
        assert np.hypot(*b["xy_err"]) < orc.cfg.tol
    dev = np.array(dev)
    assert same < len(dev)          # some runs end on a different iteration
    assert dev.max() > 1e-4         # ... with a visibly different final cost
    assert dev.max() < 0.2          # ... that is still the same valley


def test_optimiser_outcomes_are_multimodal():
    """The reference's stopping rule (relative decrease < 5e-4 over 3 iterations) is loose enough that the SAME problem,
    perturbed in the 14th digit, ends on clearly separated costs: problem 33 of the Monte-Carlo seed used by the GPU
    whole-plan test lands on ~310 or ~342 (10 % apart).  This is why whole plans are compared statistically."""
    import copy
    from alore_legged_manipulator_amd.flat_traj import monte_carlo_goals
    from oracle.backend_driver import BackendOracle, EsdfGrid
    orc = BackendOracle()
    grid = EsdfGrid.free(half=20.0)
    ft = monte_carlo_goals(96, seed=20260206)[33]
    rng = np.random.default_rng(1)
    costs = []
    for _ in range(16):
        f2 = copy.deepcopy(ft)
        f2.traj_pts = ft.traj_pts * (1 + 1e-14 * rng.standard_normal(ft.traj_pts.shape))
        costs.append(orc.minco_plan(grid, f2)["cost"])
    costs = np.sort(costs)
    assert (costs[-1] - costs[0]) / costs[0] > 0.05, costs


def test_esdf_construction_equals_the_exact_euclidean_transform():
    """be_update_esdf2d (SDFmap::updateESDF2d + fillESDF restated, with the reference's stride-Y scratch indexing) against
    SciPy's exact Euclidean distance transform: positive part outside obstacles, -(inside distance) + res inside."""
    from scipy import ndimage
    from oracle.backend_driver import EsdfGrid
    rng = np.random.default_rng(0)
    n, res = 120, 0.1
    g = np.ones((n, n), np.uint8)
    for _ in range(12):
        x, y = rng.integers(5, n - 15, 2)
        w, h = rng.integers(2, 10, 2)
        g[x:x + w, y:y + h] = 2
    g[40:44, 60:63] = 0                                      # unknown cells count as free for the inside distance
    E = EsdfGrid.from_occupancy(g, -6.0, -6.0, res)
    occ = g == 2
    exp = np.where(occ, -ndimage.distance_transform_edt(occ) * res + res, ndimage.distance_transform_edt(~occ) * res)
    inner = (slice(0, n - 2), slice(0, n - 2))               # the combine step leaves the last row / column untouched
    assert np.max(np.abs(E.dist[inner] - exp[inner])) < 1e-12
    assert np.all(E.dist[n - 1, :] > 1e300) and np.all(E.dist[:, n - 1] > 1e300)
    # a window around the robot only: cells outside keep the initial DBL_MAX
    W = EsdfGrid.from_occupancy(g, -6.0, -6.0, res, odom=(0.0, 0.0), detection_range=3.0)
    assert W.dist[0, 0] > 1e300 and W.dist[60, 60] < 1e3
