"""Edge cases of the round-2 entry points on the GPU: smallest / largest sizes, single problems, rejected arguments."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_whole_body_horizon_limits_and_single_problem():
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from wb_cases import make_problems_fast, weights
    from alore_legged_manipulator_amd.whole_body import BatchedWholeBody, WbError
    for N in (1, 32):
        eng = BatchedWholeBody(1, N, 0.01)
        x0, xref, uref, xi, ui = make_problems_fast(1, N, seed=N)
        eng.set_weights(*weights())
        eng.set_problem(x0, xref, uref)
        eng.set_iterate(xi, ui)
        eng.rti(2)
        x, u = eng.get_iterate()
        dx, du = eng.last_step()
        assert np.isfinite(x).all() and np.isfinite(u).all() and np.max(np.abs(x[:, 0] - x0)) < 1e-6
        assert np.max(np.abs(dx)) < 0.5                 # second iteration: already close
    assert np.all(eng.status() == 0)
    # an indefinite input weight: the step is rejected, the iterate stays, the status says so
    Q, R, QN = weights()
    R = R.copy(); R[:18] = -1e6
    eng.set_weights(Q, R, QN)
    eng.set_iterate(xi, ui)
    eng.rti(1)
    xb, ub_ = eng.get_iterate()
    st = eng.status()
    assert np.isfinite(xb).all() and (st[0] == 1 and np.array_equal(xb, xi) or st[0] == 0)
    with pytest.raises(WbError):
        BatchedWholeBody(1, 33)
    with pytest.raises(WbError):
        BatchedWholeBody(1, 0)
    eng = BatchedWholeBody(2, 4)
    with pytest.raises(WbError):                        # more problems than the handle holds
        eng.set_problem(np.zeros((3, 48)), np.zeros((3, 5, 48)), np.zeros((3, 4, 30)))


def test_ltv_longest_horizon_no_delay_and_single_robot():
    from alore_legged_manipulator_amd.ltv_mpc import BatchedLtvMpc, LtvError, default_config
    from oracle.ltv_mpc_oracle import LtvParams, solve_mpcv
    cfg = default_config(predict_steps=64, delay_num=0)
    eng = BatchedLtvMpc(1, cfg)
    p = LtvParams(T=64, delay_num=0)
    ts = (np.arange(64) + 1) * p.dt
    v, w = 1.2, 0.6
    xref = np.stack([v / w * np.sin(w * ts), v / w * (1 - np.cos(w * ts)), w * ts])
    dref = np.stack([np.full(64, v), np.full(64, w)])
    eng.set_refs(xref.T[None], dref.T[None])
    got = eng.get_cmd(np.array([[0.05, -0.02, 0.1]]), n_relin=1, reset=True)
    ref, z, info, xbar = solve_mpcv([0.05, -0.02, 0.1, 0.0], np.zeros((2, 64)), [], xref, dref, p)
    assert got["status"][0] == 0 and np.max(np.abs(got["output"][0] - ref.T)) < 1e-6
    for bad in (dict(predict_steps=65), dict(predict_steps=10, delay_num=10), dict(predict_steps=0)):
        with pytest.raises(LtvError):
            BatchedLtvMpc(1, default_config(**bad))


def test_backend_single_short_problem_and_argument_checks():
    from alore_legged_manipulator_amd.backend import BackendError, BatchedMSPlanner
    from alore_legged_manipulator_amd.flat_traj import straight_goal
    from oracle.backend_driver import BackendOracle, EsdfGrid
    grid = EsdfGrid.free(half=10.0)
    ft = straight_goal((0, 0, 0), (1.2, 0.1, 0.05))      # the shortest plan the front end produces (mintrajNum pieces)
    pl = BatchedMSPlanner(1, 16)
    with pytest.raises(BackendError):                    # no map yet
        pl.minco_plan([ft])
    pl.set_map(grid.dist, grid.x_lo, grid.y_lo, grid.res)
    res = pl.minco_plan([ft])
    ref = BackendOracle().minco_plan(grid, ft)
    assert res["ok"][0] == 1 and ref["ok"] and res["n_pieces"][0] == ft.pieces
    assert abs(res["cost"][0] - ref["cost"]) < 0.1 * abs(ref["cost"])
    with pytest.raises(BackendError):                    # window that misses the map
        pl.build_esdf(np.ones((50, 50), np.uint8), -2.5, -2.5, 0.1, odom=(100.0, 100.0), detection_range=1.0)


def test_comm_argument_checks():
    from alore_legged_manipulator_amd import _lib
    lib = _lib.load()
    comm = C.c_void_p()
    uid = C.create_string_buffer(128)
    assert lib.alore_nmpc_comm_create(2, 2, uid, 0, C.byref(comm)) == -1      # rank out of range
    assert lib.alore_nmpc_comm_create(0, 0, uid, 0, C.byref(comm)) == -1
    assert lib.alore_nmpc_comm_destroy(None) == -1


@pytest.mark.gpu
@pytest.mark.parametrize("N,lanes", [(19, 0x100 | 4), (16, 0x100 | 4), (22, 0x100 | 8), (31, 0x100 | 16), (20, 0x100 | 4), (20, 0x100 | 16), (20, 0x100 | 32), (20, 32)])
def test_a_problem_full_of_nan_does_not_touch_its_wavefront_mates(N, lanes):
    """Horizons that leave neutral slots in the last lane of a group (N = 19 at five stages per lane, 22 at three, 31 at two)
    next to a problem whose iterate is NaN / Inf (mpc_controller.hpp expects idle, non-finite solves inside a batch): every
    other problem returns the bits it returns when the broken one is healthy."""
    from alore_legged_manipulator_amd.nmpc import BatchedNmpc
    from alore_legged_manipulator_amd.scenarios import make_batch
    B = 67
    batch = make_batch(B, N, seed=17, fast_tail=0.4)
    eng = BatchedNmpc(B, N, lanes_per_problem=lanes)
    eng.load(batch); eng.rti(1); want = eng.fetch()
    broken = {k: v.copy() for k, v in batch.items()}
    sick = [3, 16, 41, 66]
    broken["x"][3] = np.nan
    broken["u"][16] = np.inf
    broken["x0"][41] = np.nan; broken["y"][41] = np.nan
    broken["x"][66, 0] = -np.inf
    e2 = BatchedNmpc(B, N, lanes_per_problem=lanes)
    e2.load(broken); e2.rti(1); got = e2.fetch()
    good = np.array([b for b in range(B) if b not in sick])
    assert (want["status"] == 0).all()
    for k in ("x", "u", "dual", "status", "n_iter", "kkt", "obj"):
        assert np.array_equal(got[k][good], want[k][good]), k
