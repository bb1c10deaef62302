"""Batched back_end on the GPU (include/alore_backend.h) against the float64 CPU oracle
(oracle/backend_oracle.c), through the C ABI.  Tolerances: the optimiser is float64 on both sides; one cost
evaluation differs by summation order and libm-vs-device sin/cos only (1e-10 relative asserted, ~1e-13
measured); whole optimisations are compared on what the reference's own stopping rule pins (delta = 5e-4
relative cost decrease over 3 iterations): final cost and terminal error, not iterate-for-iterate."""
import numpy as np
import pytest

from alore_legged_manipulator_amd.flat_traj import monte_carlo_goals, straight_goal, waypoint_path

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def orc():
    from oracle.backend_driver import BackendOracle
    return BackendOracle()


def free_grid(half=20.0):
    from oracle.backend_driver import EsdfGrid
    return EsdfGrid.free(half=half)


def circle_grid(cx, cy, r, half=12.0, res=0.1):
    from oracle.backend_driver import EsdfGrid
    return EsdfGrid.from_field(lambda X, Y: np.hypot(X - cx, Y - cy) - r, half=half, res=res)


def planner_for(grid, n, max_pieces=16, cfg=None):
    from alore_legged_manipulator_amd.backend import BatchedMSPlanner
    pl = BatchedMSPlanner(n, max_pieces, cfg)
    pl.set_map(grid.dist, grid.x_lo, grid.y_lo, grid.res)
    return pl


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1e-300, np.max(np.abs(b))))


@pytest.mark.parametrize("stage", [1, 2])
def test_cost_and_gradient_match_oracle_free_map(orc, stage):
    grid = free_grid()
    fts = monte_carlo_goals(48, seed=901)
    rng = np.random.default_rng(5)
    xs = [orc.x0(ft) + rng.normal(0, 0.05, 3 * ft.pieces - 1) for ft in fts]
    pl = planner_for(grid, len(fts))
    pl.set_problems(fts)
    kw = dict(lam=(3.0, -2.0), rho=(5e3, 2e4))
    out = pl.eval(stage, xs, **kw)
    for b, ft in enumerate(fts):
        ref = orc.eval(grid, ft, stage, xs[b], **kw)
        assert abs(out["cost"][b] - ref["cost"]) <= 1e-10 * abs(ref["cost"]), b
        assert rel(out["grad"][b], ref["grad"]) <= 1e-10, b
        assert np.max(np.abs(out["xy_err"][b] - ref["xy_err"])) <= 1e-11


def test_cost_and_gradient_match_oracle_icr_model_and_obstacle(orc):
    from alore_legged_manipulator_amd.backend import default_config
    cfg = default_config()
    cfg.standard_diff = 0
    orc.cfg.standard_diff = 0
    try:
        grid = circle_grid(3.0, 1.3, 0.9)
        fts = [waypoint_path([[0.0, 0.0], [6.0, 0.5]], 0.1, 0.4), waypoint_path([[0.5, 2.5], [5.5, 0.2]], -0.3, 0.0),
               straight_goal((0, 0, 0), (6, 2.2, 0.5))]
        xs = [orc.x0(ft) for ft in fts]
        pl = planner_for(grid, len(fts), cfg=cfg)
        pl.set_problems(fts)
        out = pl.eval(2, xs)
        free = planner_for(free_grid(12.0), len(fts), cfg=cfg)
        free.set_problems(fts)
        assert np.any(out["cost"] > free.eval(2, xs)["cost"] + 1.0)  # the obstacle term is active
        for b, ft in enumerate(fts):
            ref = orc.eval(grid, ft, 2, xs[b])
            assert abs(out["cost"][b] - ref["cost"]) <= 1e-10 * abs(ref["cost"])
            assert rel(out["grad"][b], ref["grad"]) <= 1e-10
    finally:
        orc.cfg.standard_diff = 1


def test_norm_guard_returns_zero_cost_like_the_reference(orc):
    grid = free_grid()
    ft = straight_goal((0, 0, 0), (4, 0, 0))
    x = orc.x0(ft)
    x[0] = 2e4
    pl = planner_for(grid, 1)
    pl.set_problems([ft])
    out = pl.eval(2, [x])
    assert out["cost"][0] == 0.0 and np.all(out["grad"][0] == 0.0)


@pytest.mark.parametrize("stage,iters", [(1, 3), (2, 5), (2, 25)])
def test_lbfgs_iterations_track_the_oracle(orc, stage, iters):
    """Same start, same number of L-BFGS iterations: iterates agree far below the optimiser's own tolerance."""
    grid = free_grid()
    fts = monte_carlo_goals(24, seed=77)
    xs = [orc.x0(ft) for ft in fts]
    pl = planner_for(grid, len(fts))
    pl.set_problems(fts)
    out = pl.lbfgs(stage, xs, max_iter=iters)
    for b, ft in enumerate(fts):
        ref = orc.lbfgs_run(grid, ft, stage, xs[b], lam=(0.0, 0.0), rho=(1e4, 1e4), max_iter=iters)
        assert out["ret"][b] == ref["ret"] and out["iters"][b] == ref["iters"] and out["evals"][b] == ref["evals"], b
        assert abs(out["cost"][b] - ref["cost"]) <= 1e-7 * abs(ref["cost"]), b
        assert np.max(np.abs(out["x"][b] - ref["x"])) <= 1e-6, b


@pytest.mark.parametrize("mem,iters", [(5, 25), (8, 30), (12, 25)])
def test_lbfgs_with_a_short_history_tracks_the_oracle(orc, mem, iters):
    """mem_size smaller than the run: the history wraps.  The kernel runs the two-loop recursion on aligned chunks with stored
    cross products while the history sits in slots 0 .. bound - 1 and pair by pair once it has wrapped; both forms, and the
    iteration at which one hands over to the other, against the oracle's plain recursion (lbfgs.hpp:704-735)."""
    from alore_legged_manipulator_amd.backend import default_config
    grid = free_grid()
    fts = monte_carlo_goals(12, seed=78)
    xs = [orc.x0(ft) for ft in fts]
    cfg = default_config()
    cfg.lbfgs.mem_size = mem
    cfg.path_lbfgs.mem_size = mem
    keep = (orc.cfg.lbfgs.mem_size, orc.cfg.path_lbfgs.mem_size)
    orc.cfg.lbfgs.mem_size = mem
    orc.cfg.path_lbfgs.mem_size = mem
    try:
        pl = planner_for(grid, len(fts), cfg=cfg)
        pl.set_problems(fts)
        out = pl.lbfgs(2, xs, max_iter=iters)
        wrapped = 0
        for b, ft in enumerate(fts):
            ref = orc.lbfgs_run(grid, ft, 2, xs[b], lam=(0.0, 0.0), rho=(1e4, 1e4), max_iter=iters)
            assert out["ret"][b] == ref["ret"] and out["iters"][b] == ref["iters"] and out["evals"][b] == ref["evals"], b
            assert np.max(np.abs(out["x"][b] - ref["x"])) <= 1e-6, b
            wrapped += int(ref["iters"] > mem + 2)
        assert wrapped >= len(fts) // 2   # the runs are long enough to wrap
    finally:
        orc.cfg.lbfgs.mem_size, orc.cfg.path_lbfgs.mem_size = keep


def test_minco_plan_matches_oracle_on_monte_carlo_goals(orc):
    """Whole plans.  The reference's optimiser is chaotic in the last digits of its inputs: L-BFGS with a loose
    stopping rule (relative cost decrease < 5e-4 over 3 iterations) ends on a different iteration when an input
    changes by 1e-14 (tests/test_backend_oracle.py::test_optimiser_is_sensitive_to_the_last_digits: median 3e-3,
    up to 5e-2 in the final cost).  So plans are compared (i) exactly where the algorithm is deterministic enough
    -- validity, terminal error, returned spline -- and (ii) statistically: the GPU-vs-oracle spread must stay
    within the oracle's own spread under a 1e-14 input perturbation, measured here on the same problems."""
    import copy
    grid = free_grid()
    fts = monte_carlo_goals(96, seed=20260206)
    pl = planner_for(grid, len(fts))
    res = pl.minco_plan(fts)
    assert np.all(res["ok"] == 1) and np.all(res["attempts"] == 1)
    rng = np.random.default_rng(0)
    dev_cost, dev_T, self_cost, self_T = [], [], [], []
    for b, ft in enumerate(fts):
        ref = orc.minco_plan(grid, ft)
        assert ref["ok"]
        M = ft.pieces
        assert res["n_pieces"][b] == M
        assert np.hypot(*res["xy_err"][b]) < orc.cfg.tol
        assert res["lbfgs_ret"][b] in (0, 1) and res["path_ret"][b] in (0, 1)
        dev_cost.append(abs(res["cost"][b] - ref["cost"]) / abs(ref["cost"]))
        dev_T.append(abs(res["T"][b, :M].sum() - ref["T"].sum()) / ref["T"].sum())
        # the returned coefficients are the spline of the returned way-points / durations
        tail = ft.final_state.copy(); tail[1, 0] = res["tail_s"][b]
        coef = orc.spline(res["T"][b, :M], res["inner"][b, :M - 1], ft.start_state, tail)
        assert np.max(np.abs(res["coef"][b, :6 * M] - coef)) <= 1e-9 * max(1.0, np.max(np.abs(coef)))
        # the oracle against itself with inputs perturbed in the 14th digit
        f2 = copy.deepcopy(ft)
        f2.traj_pts = ft.traj_pts * (1 + 1e-14 * rng.standard_normal(ft.traj_pts.shape))
        per = orc.minco_plan(grid, f2)
        self_cost.append(abs(per["cost"] - ref["cost"]) / abs(ref["cost"]))
        self_T.append(abs(per["T"].sum() - ref["T"].sum()) / ref["T"].sum())
    dev_cost, dev_T, self_cost, self_T = map(np.array, (dev_cost, dev_T, self_cost, self_T))
    print(f"cost dev GPU-vs-oracle median {np.median(dev_cost):.2e} max {dev_cost.max():.2e}; oracle self median "
          f"{np.median(self_cost):.2e} max {self_cost.max():.2e}")
    assert np.median(dev_cost) <= 2.0 * np.median(self_cost) + 1e-6
    assert np.quantile(dev_cost, 0.9) <= 2.0 * np.quantile(self_cost, 0.9) + 1e-6
    assert np.median(dev_T) <= 2.0 * np.median(self_T) + 1e-6
    # The tail: the loose stopping rule makes some problems bimodal -- the oracle itself, under 1e-14 perturbations, ends
    # on either of two costs 10 % apart in about half the runs of problem 33 of this seed and on a plateau 45 % above
    # its usual cost in 1 of 24 runs of problem 40 (tests/test_backend_oracle.py::test_optimiser_outcomes_are_multimodal
    # pins that on the CPU).  So a few GPU plans may sit outside twice the one-perturbation spread; each of them
    # must still be a plan the reference accepts (converged L-BFGS return code, terminal error below its tolerance
    # -- asserted for every problem above) with a cost in the range the stopping rule can produce.
    outliers = np.nonzero(dev_cost > 2.0 * self_cost.max() + 1e-6)[0]
    print("outliers beyond twice the oracle's one-perturbation spread:", [(int(b), round(float(dev_cost[b]), 3)) for b in outliers])
    assert len(outliers) <= 4, outliers
    assert dev_cost.max() < 0.6


def test_obstacle_is_avoided_and_retry_logic_runs(orc):
    grid = circle_grid(3.0, 1.3, 0.9)
    fts = [waypoint_path([[0.0, 0.0], [6.0, 0.5]], 0.1, 0.4), straight_goal((0, 0, 0), (6, 2.2, 0.5))]
    pl = planner_for(grid, len(fts))
    res = pl.minco_plan(fts)
    for b, ft in enumerate(fts):
        ref = orc.minco_plan(grid, ft)
        assert bool(res["ok"][b]) == ref["ok"]
        assert res["attempts"][b] == ref["attempts"]
        assert res["min_dist"][b] > orc.cfg.final_min_safe_dis
        assert abs(res["cost"][b] - ref["cost"]) <= 0.1 * abs(ref["cost"])  # chaotic iteration, see above


def test_results_are_bit_reproducible_and_independent_of_batch_mates(orc):
    grid = free_grid()
    fts = monte_carlo_goals(40, seed=3)
    pl = planner_for(grid, len(fts))
    a = pl.minco_plan(fts)
    b = pl.minco_plan(fts)
    for k in ("inner", "T", "coef", "cost", "evals"):
        assert np.array_equal(a[k], b[k]), k
    sub = [fts[i] for i in (7, 3, 31)]
    c = planner_for(grid, 3).minco_plan(sub)
    for j, i in enumerate((7, 3, 31)):
        assert np.array_equal(c["coef"][j], a["coef"][i]) and c["cost"][j] == a["cost"][i]


def test_long_paths_use_the_32_piece_kernel(orc):
    grid = free_grid(40.0)
    ft = waypoint_path([[-6.0, -4.0], [0.0, -1.0], [6.0, 4.0]], 0.0, 1.0)
    assert 16 < ft.pieces <= 32
    pl = planner_for(grid, 1, max_pieces=32)
    pl.set_problems([ft])
    x = orc.x0(ft)
    out = pl.eval(2, [x])
    ref = orc.eval(grid, ft, 2, x)
    assert abs(out["cost"][0] - ref["cost"]) <= 1e-10 * abs(ref["cost"])
    assert rel(out["grad"][0], ref["grad"]) <= 1e-10
    res = pl.minco_plan([ft])
    assert res["ok"][0] == 1 and np.hypot(*res["xy_err"][0]) < orc.cfg.tol


def test_lbfgs_of_the_32_piece_kernel_tracks_the_oracle(orc):
    """The 32-piece build keeps the pair-at-a-time two-loop recursion (two registers per vector): its iterates against the
    oracle's, like test_lbfgs_iterations_track_the_oracle for the 16-piece build."""
    grid = free_grid(40.0)
    ft = waypoint_path([[-6.0, -4.0], [0.0, -1.0], [6.0, 4.0]], 0.0, 1.0)
    assert 16 < ft.pieces <= 32
    x0 = orc.x0(ft)
    pl = planner_for(grid, 1, max_pieces=32)
    pl.set_problems([ft])
    for stage, iters in ((1, 4), (2, 20)):
        out = pl.lbfgs(stage, [x0], max_iter=iters)
        ref = orc.lbfgs_run(grid, ft, stage, x0, lam=(0.0, 0.0), rho=(1e4, 1e4), max_iter=iters)
        assert out["ret"][0] == ref["ret"] and out["iters"][0] == ref["iters"] and out["evals"][0] == ref["evals"], stage
        assert np.max(np.abs(out["x"][0] - ref["x"])) <= 1e-6, stage


def test_too_many_pieces_is_refused():
    from alore_legged_manipulator_amd.backend import BackendError, BatchedMSPlanner
    grid = free_grid(40.0)
    ft = waypoint_path([[-6.0, -4.0], [0.0, -1.0], [6.0, 4.0]], 0.0, 1.0)
    pl = BatchedMSPlanner(1, 16)
    pl.set_map(grid.dist, grid.x_lo, grid.y_lo, grid.res)
    with pytest.raises(BackendError):
        pl.set_problems([ft])


def test_esdf_built_on_the_device_is_bit_identical_to_the_oracle(orc):
    """alore_backend_build_esdf (SDFmap::updateESDF2d on the GPU, one thread per line) against the oracle's restatement:
    all intermediate values are sums of squared integers, so the fields must agree bit for bit -- whole map and a
    window around the odometry -- and the planner must then avoid an obstacle given only as occupancy."""
    from oracle.backend_driver import EsdfGrid
    rng = np.random.default_rng(3)
    n, res, lo = 240, 0.1, -12.0
    g = np.ones((n, n), np.uint8)
    for _ in range(25):
        x, y = rng.integers(5, n - 25, 2)
        w, h = rng.integers(2, 18, 2)
        g[x:x + w, y:y + h] = 2
    g[100:104, 30:36] = 0
    from alore_legged_manipulator_amd.backend import BatchedMSPlanner
    pl = BatchedMSPlanner(2, 16)            # no map yet: the first build starts from DBL_MAX everywhere (sdf_map.h:160)
    for kw in (dict(), dict(odom=(1.0, -2.0), detection_range=5.0)):
        dev = pl.build_esdf(g, lo, lo, res, **kw) if not kw else None
        if kw:   # the windowed update refreshes the handle's existing field in place: compare with the same sequence
            dev = pl.build_esdf(g, lo, lo, res, **kw)
            ref = EsdfGrid.from_occupancy(g, lo, lo, res).dist
            from oracle.backend_driver import SO
            import ctypes as C
            L = C.CDLL(SO)
            L.be_update_esdf2d.restype = C.c_int
            L.be_update_esdf2d.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p]
            assert L.be_update_esdf2d(g.ctypes.data, n, n, res, lo, lo, kw["odom"][0], kw["odom"][1], kw["detection_range"], ref.ctypes.data) == 0
        else:
            ref = EsdfGrid.from_occupancy(g, lo, lo, res).dist
        assert np.array_equal(dev, ref)
    # an obstacle known only as occupied cells: disc of radius 0.9 m at (3.0, 1.3)
    c = (np.arange(n) + 0.5) * res + lo
    X, Y = np.meshgrid(c, c, indexing="ij")
    disc = np.where(np.hypot(X - 3.0, Y - 1.3) < 0.9, 2, 1).astype(np.uint8)
    pl2 = BatchedMSPlanner(1, 16)
    pl2.build_esdf(disc, lo, lo, res)
    ft = waypoint_path([[0.0, 0.0], [6.0, 0.5]], 0.1, 0.4)
    res2 = pl2.minco_plan([ft])
    assert res2["ok"][0] == 1 and res2["min_dist"][0] > orc.cfg.final_min_safe_dis


def test_path_points_match_the_oracle(orc):
    """alore_backend_path_points (MSPlanner::mincoPointPub: what the ALORE FSM follows) on the GPU's own plans against the
    oracle's restatement fed with the same coefficients: the same Simpson panels in the same order of additions, 1e-10;
    a plan the optimiser rejected has no points."""
    from oracle.backend_driver import path_points
    fts = monte_carlo_goals(24, seed=9)
    pl = planner_for(free_grid(), len(fts))
    res = pl.minco_plan(fts)
    for panels in (3, 8):
        got = pl.path_points(panels)
        for b, ft in enumerate(fts):
            M = res["n_pieces"][b]
            xy, yaw = got[b]
            if not res["ok"][b]:
                assert xy.shape[0] == 0
                continue
            T, coef = res["T"][b, :M], res["coef"][b, :6 * M].reshape(-1)
            exy, eyaw = path_points(T, coef, panels, ft.start_xytheta[:2])
            assert xy.shape == exy.shape and np.max(np.abs(xy - exy)) < 1e-10 and np.max(np.abs(yaw - eyaw)) < 1e-10
            assert np.hypot(*(xy[-1] - np.array(ft.final_xytheta[:2]))) < 0.05      # the path ends at the goal


def test_predicted_state_matches_the_oracle(orc):
    """alore_backend_predicted_state (MSPlanner::get_the_predicted_state / _and_path) on the GPU's own plans against the
    oracle's restatement fed with the same coefficients: same Simpson steps in the same order, 1e-10."""
    from oracle.backend_driver import predicted_state
    fts = monte_carlo_goals(24, seed=8)
    pl = planner_for(free_grid(), len(fts))
    res = pl.minco_plan(fts)
    rng = np.random.default_rng(2)
    total = np.array([res["T"][b, :res["n_pieces"][b]].sum() for b in range(len(fts))])
    times = total * rng.uniform(0.05, 1.3, len(fts))           # some beyond the end of the plan (clamped)
    out = pl.predicted_state(times, resolution=0.01)
    st = total * rng.uniform(0.0, 0.4, len(fts))
    sx = rng.uniform(-1, 1, (len(fts), 3))
    out2 = pl.predicted_state(np.maximum(times, st + 0.05), resolution=0.02, start_times=st, start_xytheta=sx)
    for b, ft in enumerate(fts):
        M = res["n_pieces"][b]
        T, coef = res["T"][b, :M], res["coef"][b, :6 * M].reshape(-1)
        xyt, vaj, oaj, fwd = predicted_state(T, coef, 0.01, times[b], start_xytheta=ft.start_xytheta)
        assert np.max(np.abs(out["xytheta"][b] - xyt)) < 1e-10 and np.max(np.abs(out["vaj"][b] - vaj)) < 1e-9
        assert np.max(np.abs(out["oaj"][b] - oaj)) < 1e-9 and bool(out["forward"][b]) == fwd
        xyt, vaj, oaj, fwd = predicted_state(T, coef, 0.02, max(times[b], st[b] + 0.05), start_time=st[b], start_xytheta=sx[b])
        assert np.max(np.abs(out2["xytheta"][b] - xyt)) < 1e-10 and bool(out2["forward"][b]) == fwd
    # at the end of the plan the predicted pose is the goal the optimiser was asked to reach (terminal error < tol)
    end = pl.predicted_state(total + 1.0)
    goal = np.array([ft.final_xytheta for ft in fts])
    assert np.max(np.hypot(end["xytheta"][:, 0] - goal[:, 0], end["xytheta"][:, 1] - goal[:, 1])) < 0.05
