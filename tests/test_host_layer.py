"""Host layer (C++: alore_legged_manipulator_amd/host/, C entry points include/alore_nmpc_host.h) and the
device-side reference path (csrc/ref_sampler.hip) against
  * the float64 host restatement of the reference's sampling (oracle/traj_oracle.hpp, pinned by
    tests/test_traj_oracle.py), and
  * golden trajectories computed with NumPy / SciPy only (tests/golden/traj_polynomes.npz), so that the device
    spline (csrc/minco_spline.h: knot-state system) is checked against something that shares no code with it."""
import math
import os

import numpy as np
import pytest

from tests.test_traj_oracle import arc_polynome, exact_pose, golden_messages

pytestmark = pytest.mark.gpu


def as_host_msg(m):
    from alore_legged_manipulator_amd.host import Polynome
    return Polynome(m.innerpoints, m.t_pts, m.init_pva, m.tail_pva, m.start_position, m.ICR, m.traj_start_time)


def wiggly_polynome(rng, pieces, t0=0.0):
    """A trajectory that is not a constant-twist arc: random inner points, boundary velocities and accelerations."""
    from oracle.traj_driver import Polynome
    M = len(pieces)
    th = np.cumsum(rng.uniform(-0.6, 0.6, M + 1)) + rng.uniform(-3, 3)
    sl = np.cumsum(rng.uniform(0.1, 0.8, M + 1))
    inner = np.stack([th[1:M], sl[1:M]], 1)
    init = [th[0], sl[0], rng.uniform(-1, 1), rng.uniform(0.2, 1.5), rng.uniform(-1, 1), rng.uniform(-1, 1)]
    tail = [th[M], sl[M], rng.uniform(-1, 1), rng.uniform(0.2, 1.5), rng.uniform(-1, 1), rng.uniform(-1, 1)]
    return Polynome(inner, pieces, init, tail, rng.uniform(-1, 1, 3), [-0.3, 0.3, rng.uniform(0, 0.3)], t0)


def test_controller_tick_matches_oracle_pipeline():
    """BatchedMpcController.tick == (oracle RefSampler refs -> MpcWrapper::solve semantics -> oracle RTI tick)."""
    from alore_legged_manipulator_amd.host import BatchedMpcController
    from oracle.drivers import Oracle
    from oracle.traj_driver import RefSampler
    B, N, dt = 6, 20, 0.01
    rng = np.random.default_rng(9)
    ctl = BatchedMpcController(B, N, dt, delay_num=1)
    lone = []
    for b in range(B):
        v, w, xv = rng.uniform(0.5, 1.8), rng.uniform(-1.2, 1.2), rng.uniform(0, 0.3)
        m = arc_polynome(v, w, 0.0, [0.4, 0.4, 0.4], xv=xv, t0=0.0)
        yr, yl = -rng.uniform(0.2, 0.35), rng.uniform(0.2, 0.35)
        od = (rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), rng.uniform(-0.3, 0.3))
        r = ctl.robots[b]
        r.traj(as_host_msg(m)); r.odom(*od); r.icr(yr, yl, xv)
        s = RefSampler(N, dt); s.traj(m); s.odom(*od); s.icr(yr, yl, xv)
        lone.append((s, od, (xv, yr, yl)))
    cmd = ctl.tick(0.123)
    orc = Oracle(N)
    for b in range(B):
        s, od, icr = lone[b]
        rs, ri, _ = s.refs(0.123, smooth=True)
        p = {"x": np.tile(np.float32(od), N + 1), "u": np.zeros(2 * N, np.float32),
             "od": np.tile(np.float32(icr), N + 1),
             "y": np.concatenate([rs[:N], ri[:N]], 1).astype(np.float32).reshape(-1), "yN": rs[N].astype(np.float32),
             "W": np.tile(np.diag([10, 10, 0.5, 0.1, 0.1]).astype(np.float32).reshape(-1), N),
             "WN": np.diag([10, 10, 0.5]).astype(np.float32).reshape(-1), "x0": np.float32(od)}
        orc.reset(); orc.initialize_solver(); orc.load(p)
        orc.preparation_step()
        assert orc.feedback_step() == 0
        u = orc.v["u"].reshape(N, 2)
        assert np.max(np.abs(cmd[b] - u[1])) < 1e-4 * max(1.0, np.max(np.abs(u))), (b, cmd[b], u[1])
        ps, pu, st = ctl.prediction(b)
        assert st == 0 and np.max(np.abs(ps.reshape(-1) - orc.v["x"])) < 1e-4 * max(1.0, np.max(np.abs(orc.v["x"])))


def test_device_reference_sampling_matches_oracle_sampler():
    """alore_nmpc_refs_* (ref_sampler.hip) against the float64 oracle RefSampler: y, yN, od, x0 as float32,
    across heading wrap-around, the end of the trajectory, a trajectory swap and a robot without trajectory."""
    from alore_legged_manipulator_amd.host import BatchedMpcController
    from oracle.traj_driver import RefSampler
    B, N, dt = 9, 20, 0.01
    rng = np.random.default_rng(31)
    dev = BatchedMpcController(B, N, dt, max_pieces=8, max_checkpoints=64)
    lone = [RefSampler(N, dt) for _ in range(B)]

    spec = []
    for b in range(B - 1):                       # robot B-1 never gets a trajectory
        v, w, xv = rng.uniform(0.5, 1.8), rng.uniform(-1.2, 1.2), rng.uniform(0, 0.3)
        th0 = math.pi - 0.1 if b == 0 else rng.uniform(-3, 3)   # b = 0: heading crosses +pi at once
        if b == 0:
            w = 1.5
        pieces = [0.4] * int(rng.integers(1, 5))
        m = arc_polynome(v, w, th0, pieces, xv=xv, t0=0.0)
        yr, yl = -rng.uniform(0.2, 0.35), rng.uniform(0.2, 0.35)
        od = (rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), th0 + rng.uniform(-0.3, 0.3))
        dev.robots[b].traj(as_host_msg(m)); dev.robots[b].odom(*od); dev.robots[b].icr(yr, yl, xv)
        lone[b].traj(m); lone[b].odom(*od); lone[b].icr(yr, yl, xv)
        spec.append((od, (xv, yr, yl)))
    dev.robots[B - 1].odom(0.1, 0.2, 0.3)

    def check(now, active=range(B - 1)):
        cd = dev.tick(now)
        yd, yNd, odd, x0d = dev.references()
        for b in active:
            rs, ri, goal = lone[b].refs(now, smooth=True)
            want = np.concatenate([rs[:N], ri[:N]], 1).astype(np.float32)
            tol = 2e-6 * max(1.0, float(np.max(np.abs(want))))
            assert np.max(np.abs(yd[b] - want)) <= tol, (now, b, np.max(np.abs(yd[b] - want)))
            assert np.max(np.abs(yNd[b] - rs[N].astype(np.float32))) <= tol
            assert np.array_equal(odd[b], np.tile(np.float32(spec[b][1]), (N + 1, 1)))
            assert np.array_equal(x0d[b], np.float32(spec[b][0]))
            assert dev.robots[b].at_goal == goal
        assert np.all(cd[B - 1] == 0.0)
        return cd

    check(0.123)                                  # first tick (solve from scratch)
    check(0.133)                                  # warm tick
    check(0.35)                                   # horizons of the 1-piece robots run past the end (0.4 s)
    # a new trajectory arrives for robot 2, starting in the future: old one keeps being tracked until then
    m2 = arc_polynome(1.1, -0.7, 0.4, [0.5, 0.5], xv=0.1, t0=0.5)
    dev.robots[2].traj(as_host_msg(m2)); lone[2].traj(m2)
    check(0.45)
    check(0.55)                                   # swapped in
    cd = check(5.0)                               # every trajectory ended more than 1 s ago: at_goal gets set ...
    assert all(dev.robots[b].at_goal for b in range(B - 1)) and np.any(cd[:B - 1] != 0.0)
    full = dev.tick_full(5.01)                    # ... and the NEXT tick publishes the zero CarState, the last wheel command
    for b in range(B - 1):                        # once more (mpc.cpp:184-203), and forgets the trajectory
        assert full[b].state_published and full[b].v == 0.0 and full[b].omega == 0.0
        assert full[b].wheel_published and abs(full[b].right_wheel_ome - cd[b, 0]) < 1e-12
        assert not dev.robots[b].receive_traj
    assert np.all(dev.tick(5.02) == 0.0)          # nothing is published any more


def test_device_built_trajectory_store_matches_golden_and_oracle():
    """alore_nmpc_refs_set_polynomes (knot-state spline + Simpson checkpoints on the GPU) against (i) the golden
    trajectories (NumPy dense solve, SciPy quadrature: no shared code) and (ii) the oracle TrajAnal (same Simpson
    rule: checkpoints to 1e-11); then the sampled references (float32) against the oracle getRefPoints."""
    from alore_legged_manipulator_amd.nmpc import BatchedNmpc
    from oracle.traj_driver import RefSampler
    N, dt = 20, 0.01
    gold = golden_messages()
    rng = np.random.default_rng(5)
    extra = [wiggly_polynome(rng, list(rng.uniform(0.3, 0.7, int(rng.integers(1, 13)))), t0=float(rng.uniform(0, 0.05)))
             for _ in range(300 - len(gold))]        # > one chunk of 256 messages
    msgs = [m for m, _ in gold] + extra
    B = len(msgs)
    eng = BatchedNmpc(B, N, dt)
    eng.refs_init(max_pieces=12, max_checkpoints=80)
    order = rng.permutation(B)                   # store slots need not follow message order
    eng.refs_set_polynomes(order, msgs)
    # (i) golden
    for i, (m, ans) in enumerate(gold):
        d = eng.refs_download(int(order[i]))
        M = len(m.t_pts)
        assert d["valid"] and np.array_equal(d["durations"], m.t_pts)
        ref = np.transpose(ans["coef"], (0, 2, 1))
        assert np.max(np.abs(d["coeffs"][:M] - ref)) <= 1e-9 * max(1.0, np.max(np.abs(ref))), i
        # checkpoints (every 0.1 s) against the exact integral at those times
        edges = np.concatenate([[0.0], np.cumsum(m.t_pts)])
        for k, t in enumerate(ans["times"]):
            j = int(np.floor(t / 0.1 + 1e-9))
            if abs(j * 0.1 - t) < 1e-12 and j < d["checkpoints"].shape[0]:
                assert np.max(np.abs(d["checkpoints"][j] - ans["xy"][k])) < 2e-5
    # (ii) oracle TrajAnal
    hosts = {}
    for i in (0, 1, 2, 17, 255, 256, B - 1):
        r = int(order[i]); m = msgs[i]
        s = RefSampler(N, dt); s.traj(m)
        hosts[r] = (s, m)
        d = eng.refs_download(r)
        assert d["valid"] and abs(d["duration"] - m.t_pts.sum()) < 1e-12 and d["xv"] == m.ICR[2]
        seq = s.sequence()
        assert d["checkpoints"].shape[0] == seq.shape[0]
        assert np.max(np.abs(d["checkpoints"] - seq[:, :2])) < 1e-10
        _, coef = s.coefficients()
        assert np.max(np.abs(d["coeffs"][:len(m.t_pts)] - coef)) <= 1e-9 * max(1.0, np.max(np.abs(coef)))
    # sampled references
    est = rng.uniform(-0.3, 0.3, (B, 3)); icr = np.tile([0.1, -0.3, 0.3], (B, 1))
    now = 0.21
    for r, (s, m) in hosts.items():
        est[r, 2] += m.init_pva[0]               # yaw estimate near the trajectory's heading
    eng.refs_sample(now, est, icr)
    out = eng.fetch(names=("y", "yN", "od", "x0"))
    for r, (s, m) in hosts.items():
        s.odom(*est[r]); s.icr(-0.3, 0.3, 0.1)
        rs, ri, _ = s.refs(now, smooth=True)
        want = np.concatenate([rs[:N], ri[:N]], 1).astype(np.float32)
        tol = 2e-6 * max(1.0, float(np.max(np.abs(want))))
        assert np.max(np.abs(out["y"][r] - want)) <= tol and np.max(np.abs(out["yN"][r] - rs[N].astype(np.float32))) <= tol
        assert np.array_equal(out["x0"][r], est[r].astype(np.float32))
    # capacity errors are reported, the slot is invalidated
    from alore_legged_manipulator_amd._lib import NmpcError
    with pytest.raises(NmpcError):
        eng.refs_set_polynomes([0], [wiggly_polynome(rng, [0.5] * 13)])           # 13 pieces > max_pieces
    with pytest.raises(NmpcError):
        eng.refs_set_polynomes([1], [wiggly_polynome(rng, [0.9] * 10)])           # 9 s -> 91 checkpoints > 80
    assert not eng.refs_download(1)["valid"]


def test_cold_start_is_per_robot_and_emergency_stop_forgets_the_trajectory():
    """A robot whose odometry / trajectory arrives later is reset (x <- pose, u <- 0) on ITS first solve
    (mpc.cpp:317-320), exactly like a robot that was there from the start; /planner/emergency_stop makes a robot
    idle until the next trajectory (mpc.cpp:279-294)."""
    from alore_legged_manipulator_amd.host import BatchedMpcController
    B, N, dt = 3, 20, 0.01
    m = arc_polynome(1.2, 0.6, 0.2, [0.5, 0.5, 0.5], xv=0.1, t0=0.0)
    od = (0.05, -0.1, 0.25)
    late = BatchedMpcController(B, N, dt)
    alone = BatchedMpcController(1, N, dt)
    for r in (late.robots[0], late.robots[2]):
        r.traj(as_host_msg(m)); r.odom(*od); r.icr(-0.3, 0.3, 0.1)
    for t in (0.10, 0.11, 0.12):                     # robot 1 is silent: nothing published, others run
        cmd = late.tick_full(t)
        assert not cmd[1].wheel_published and cmd[0].wheel_published
    # robot 1 wakes up at t = 0.13; a lone controller started at the same time must give the same command
    late.robots[1].traj(as_host_msg(m)); late.robots[1].odom(*od); late.robots[1].icr(-0.3, 0.3, 0.1)
    alone.robots[0].traj(as_host_msg(m)); alone.robots[0].odom(*od); alone.robots[0].icr(-0.3, 0.3, 0.1)
    for t in (0.13, 0.14, 0.15):
        a, b = late.tick(t), alone.tick(t)
        assert np.array_equal(a[1], b[0]), (t, a[1], b[0])
    # emergency stop
    late.robots[0].emergency_stop()
    assert not late.robots[0].receive_traj
    cmd = late.tick_full(0.16)
    assert not cmd[0].wheel_published and cmd[1].wheel_published and cmd[2].wheel_published
    late.robots[0].traj(as_host_msg(arc_polynome(0.8, -0.3, 0.2, [0.6, 0.6], xv=0.1, t0=0.0)))
    assert late.tick_full(0.17)[0].wheel_published


def test_a_robot_that_was_idle_warm_starts_from_its_last_solve():
    """The reference resets the iterate once (solve_from_scratch_, mpc.cpp:317-320) and warm-starts ever after: a robot that
    reached its goal (or was stopped) and gets a new trajectory continues from the iterate and multipliers its LAST solve left,
    however many ticks the rest of the fleet ran in between (idle robots sit the batch launch out: alore_nmpc_set_problem_mask).
    A fleet of three, of which robot 1 pauses, against a lone controller with robot 1's history and no fleet around it."""
    from alore_legged_manipulator_amd.host import BatchedMpcController
    B, N, dt = 3, 20, 0.01
    m1 = arc_polynome(1.2, 0.6, 0.2, [0.5, 0.5, 0.5], xv=0.1, t0=0.0)
    m2 = arc_polynome(0.8, -0.4, 0.2, [0.6, 0.6], xv=0.1, t0=0.30)
    od = (0.05, -0.1, 0.25)
    fleet, lone = BatchedMpcController(B, N, dt), BatchedMpcController(1, N, dt)
    for r in list(fleet.robots) + [lone.robots[0]]:
        r.traj(as_host_msg(m1)); r.odom(*od); r.icr(-0.3, 0.3, 0.1)
    for t in (0.10, 0.11, 0.12):
        a, b = fleet.tick(t), lone.tick(t)
        assert np.array_equal(a[1], b[0])
    fleet.robots[1].emergency_stop(); lone.robots[0].emergency_stop()
    for t in (0.13, 0.14, 0.15, 0.16, 0.17):            # the fleet keeps ticking, robot 1 idles; the lone controller has nothing to solve
        cmd = fleet.tick_full(t)
        assert not cmd[1].wheel_published and cmd[0].wheel_published and cmd[2].wheel_published
    fleet.robots[1].traj(as_host_msg(m2)); lone.robots[0].traj(as_host_msg(m2))
    for t in (0.31, 0.32, 0.33):
        a, b = fleet.tick(t), lone.tick(t)
        assert np.array_equal(a[1], b[0]), (t, a[1], b[0])
    (sf, uf, stf), (sl, ul, stl) = fleet.prediction(1), lone.prediction(0)
    assert stf == 0 and stl == 0 and np.array_equal(sf, sl) and np.array_equal(uf, ul)


def test_open_loop_replay_without_mpc():
    """if_mpc = false (mpc.cpp:211-234): the planned flat velocities and accelerations at t_cur are republished."""
    from alore_legged_manipulator_amd.host import BatchedMpcController, default_params
    prm = default_params(if_mpc=0, state_seq_res=0.1, Integral_appr_resInt=4)
    for m, ans in golden_messages()[:4]:
        ctl = BatchedMpcController(1, 20, 0.01, params=prm, max_pieces=12, max_checkpoints=80)
        ctl.robots[0].traj(as_host_msg(m)); ctl.robots[0].odom(0, 0, 0)
        for k in (3, 7, 15):
            t = float(ans["times"][k])
            cmd = ctl.tick_full(m.traj_start_time + t)[0]
            v, a = ans["flat"][k][1], ans["flat"][k][2]
            assert cmd.state_published and not cmd.wheel_published
            assert abs(cmd.omega - v[0]) < 1e-8 and abs(cmd.v - v[1]) < 1e-8
            assert abs(cmd.alpha - a[0]) < 1e-7 and abs(cmd.a - a[1]) < 1e-7
        # past the end the robot is flagged at_goal and the next tick publishes the stop
        ctl.tick_full(m.traj_start_time + float(m.t_pts.sum()) + 0.01)
        assert ctl.robots[0].at_goal
        stop = ctl.tick_full(m.traj_start_time + float(m.t_pts.sum()) + 0.02)[0]
        assert stop.state_published and stop.v == 0.0 and stop.omega == 0.0 and not ctl.robots[0].receive_traj


def test_phase_stamps_do_not_disturb_the_trajectory_store(monkeypatch):
    """ALORE_NMPC_STAMPS=1 (diagnostic build path) with device references in use: the stamp buffer growth must
    leave the trajectory store alone (it once freed it)."""
    monkeypatch.setenv("ALORE_NMPC_STAMPS", "1")
    from alore_legged_manipulator_amd.host import BatchedMpcController
    B = 5
    m = arc_polynome(1.0, 0.5, 0.0, [0.5, 0.5], xv=0.1, t0=0.0)
    stamped = BatchedMpcController(B, 20, 0.01)
    for r in stamped.robots:
        r.traj(as_host_msg(m)); r.odom(0.0, 0.0, 0.0); r.icr(-0.3, 0.3, 0.1)
    a = [stamped.tick(0.1 + 0.01 * k) for k in range(3)]
    monkeypatch.delenv("ALORE_NMPC_STAMPS")
    plain = BatchedMpcController(B, 20, 0.01)
    for r in plain.robots:
        r.traj(as_host_msg(m)); r.odom(0.0, 0.0, 0.0); r.icr(-0.3, 0.3, 0.1)
    b = [plain.tick(0.1 + 0.01 * k) for k in range(3)]
    assert np.array_equal(np.array(a), np.array(b)) and np.any(np.array(a) != 0.0)
