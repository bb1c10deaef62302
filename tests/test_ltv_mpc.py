"""LTV-MPC class (the reference's `mpc` node, mpc_controller/src/mpc.cpp).
CPU: the NumPy oracle builds solveMPCV's matrices and certifies its own solution (KKT residuals <= 1e-9).
GPU: the batched working-set Riccati solver against that oracle (the reference's OSQP stops at eps 1e-6; here 1e-6 is
asserted against the exact solution)."""
import math

import numpy as np
import pytest

from oracle.ltv_mpc_oracle import LtvParams, build_qp, get_cmd, predict_motion, ref_points, solve_mpcv, solve_qp


def arc_refs(p, v, w, T):
    ts = (np.arange(T) + 1) * p.dt
    if abs(w) < 1e-9:
        xref = np.stack([v * ts, 0 * ts, 0 * ts])
    else:
        xref = np.stack([v / w * np.sin(w * ts), v / w * (1 - np.cos(w * ts)), w * ts])
    return xref, np.stack([np.full(T, v), np.full(T, w)])


def random_case(rng, p):
    v, w = rng.uniform(0.3, 3.2), rng.uniform(-2.5, 2.5)
    xref, dref = arc_refs(p, v, w, p.T)
    now = [rng.uniform(-.3, .3), rng.uniform(-.3, .3), rng.uniform(-.5, .5), 0.0]
    out = np.zeros((2, p.T)); out[0] = rng.uniform(0, 3); out[1] = rng.uniform(-2, 2)
    buff = [np.array([rng.uniform(0, 3), rng.uniform(-2, 2)]) for _ in range(p.delay_num)]
    for i in range(p.delay_num):
        out[:, i] = buff[i]
    return now, out, buff, xref, dref


def test_oracle_qp_structure_and_certificate():
    p = LtvParams()
    rng = np.random.default_rng(3)
    now, out, buff, xref, dref = random_case(rng, p)
    xbar = predict_motion(now, out, p)
    P, q, A, l, u, K = build_qp(xbar, xref, dref, p)
    assert K == p.T - p.delay_num and P.shape == (5 * K, 5 * K) and A.shape == (2 * K + 3 * K + 2 * K - 2, 5 * K)
    assert np.allclose(P, P.T) and np.all(np.linalg.eigvalsh(P) >= -1e-12)
    # the Hessian's input block is R + Q2 on the diagonal plus the Rd chain (mpc.cpp:345-358)
    iu = 3 * K
    assert P[iu, iu] == 2 * (p.R[0] + p.Rd[0] + p.Q[2]) and P[iu + 2, iu + 2] == 2 * (p.R[0] + 2 * p.Rd[0] + p.Q[2])
    assert P[iu + 2, iu] == -2 * p.Rd[0] and P[iu + 3, iu + 1] == -2 * p.Rd[1]
    z, y, info = solve_qp(P, q, A, l, u)
    assert info["primal"] < 1e-9 and info["dual"] < 1e-9 and info["compl"] < 1e-9
    # the dynamics rows reproduce the linear model about the rollout
    s1 = z[0:3]; u0 = z[iu:iu + 2]
    th, v = xbar[p.delay_num][2], xbar[p.delay_num][3]
    pred = np.array([xbar[p.delay_num][0] + math.cos(th) * p.dt * u0[0] - math.sin(th) * p.dt * v * (xbar[p.delay_num][2] - th),
                     xbar[p.delay_num][1] + math.sin(th) * p.dt * u0[0], th + p.dt * u0[1]])
    assert np.allclose(s1, pred, atol=1e-9)
    # rate limits and boxes hold
    uu = z[iu:].reshape(K, 2)
    assert np.all(np.abs(np.diff(uu[:, 0])) <= p.max_cv + 1e-9) and np.all(np.abs(np.diff(uu[:, 1])) <= p.max_comega + 1e-9)
    assert np.all(np.abs(uu[:, 0]) <= p.max_speed + 1e-9)


def test_oracle_relinearisation_converges():
    p = LtvParams()
    rng = np.random.default_rng(5)
    now, out, buff, xref, dref = random_case(rng, p)
    o3, _, _ = get_cmd(now, out, buff, xref, dref, p, 3)
    o8, _, _ = get_cmd(now, out, buff, xref, dref, p, 8)
    o9, _, _ = get_cmd(now, out, buff, xref, dref, p, 9)
    assert np.max(np.abs(o9 - o8)) < 1e-6 and np.max(np.abs(o8 - o3)) < 0.2


@pytest.mark.gpu
@pytest.mark.parametrize("delay", [0, 1, 3])
def test_gpu_qp_matches_the_exact_solution(delay):
    from alore_legged_manipulator_amd.ltv_mpc import BatchedLtvMpc, default_config
    p = LtvParams(delay_num=delay)
    B = 40
    rng = np.random.default_rng(20 + delay)
    cases = [random_case(rng, p) for _ in range(B)]
    eng = BatchedLtvMpc(B, default_config(delay_num=delay))
    eng.set_refs(np.stack([c[3].T for c in cases]), np.stack([c[4].T for c in cases]))
    eng.set_state(np.stack([c[1].T for c in cases]), np.stack([np.stack(c[2]) for c in cases]) if delay else None)
    got = eng.get_cmd(np.array([c[0][:3] for c in cases]), n_relin=1)
    assert np.all(got["status"] == 0) and got["sweeps"].max() <= 40
    worst = 0.0
    for b, (now, out, buff, xref, dref) in enumerate(cases):
        ref, z, info, xbar = solve_mpcv(now, out, buff, xref, dref, p)
        assert max(info.values()) < 1e-7  # the oracle certifies itself (KKT residuals)
        worst = max(worst, float(np.max(np.abs(got["output"][b] - ref.T))))
    assert worst < 1e-6, worst
    # warm working set: the same QP again needs one confirming sweep
    eng.set_state(np.stack([c[1].T for c in cases]), np.stack([np.stack(c[2]) for c in cases]) if delay else None)
    again = eng.get_cmd(np.array([c[0][:3] for c in cases]), n_relin=1)
    assert np.all(again["sweeps"] <= 2) and np.max(np.abs(again["output"] - got["output"])) < 1e-9


@pytest.mark.gpu
def test_gpu_qp_on_a_long_horizon_uses_four_stages_per_lane():
    """predict_steps = 40 (K = 39 stages > 32): the lanes kernel's four-stages-per-lane instantiation, against the oracle's
    certified dense solve, moderate and saturated cases.  (At 56 steps some cold starts need more than the 64-sweep cap of
    the working-set method -- in both kernels alike; the reference's configuration has 30.)"""
    from alore_legged_manipulator_amd.ltv_mpc import BatchedLtvMpc, default_config
    p = LtvParams(T=40)
    B = 24
    rng = np.random.default_rng(77)
    cases = [random_case(rng, p) for _ in range(B)]
    for c in cases[B // 2:]:   # drive half of them into the boxes and rate limits
        c[1][0] *= 3.0
        c[3][:2] *= 1.5
    eng = BatchedLtvMpc(B, default_config(predict_steps=40))
    eng.set_refs(np.stack([c[3].T for c in cases]), np.stack([c[4].T for c in cases]))
    eng.set_state(np.stack([c[1].T for c in cases]), np.stack([np.stack(c[2]) for c in cases]))
    got = eng.get_cmd(np.array([c[0][:3] for c in cases]), n_relin=1)
    assert np.all(got["status"] == 0)
    worst = 0.0
    for b, (now, out, buff, xref, dref) in enumerate(cases):
        ref, z, info, xbar = solve_mpcv(now, out, buff, xref, dref, p)
        assert max(info.values()) < 1e-7
        worst = max(worst, float(np.max(np.abs(got["output"][b] - ref.T))))
    assert worst < 1e-6, worst


def wide_case(rng):
    """Far outside the envelope: references beyond the input boxes, rollouts that start saturated -- box bounds and
    rate limits active together over long runs of stages (chains of rate-limited stages hanging from a box)."""
    delay = int(rng.integers(0, 4))
    p = LtvParams(delay_num=delay)
    v, w = rng.uniform(-0.5, 3.5), rng.uniform(-3.5, 3.5)
    if abs(w) < 1e-3:
        w = 0.1
    xref, dref = arc_refs(p, v, w, p.T)
    now = [rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(-1.5, 1.5), 0.0]
    out = np.zeros((2, p.T)); out[0] = rng.uniform(-1, 3.5) * rng.uniform(0, 1, p.T); out[1] = rng.uniform(-3, 3)
    buff = [np.array([rng.uniform(0, 3), rng.uniform(-2, 2)]) for _ in range(delay)]
    for k in range(delay):
        out[:, k] = buff[k]
    return p, (now, out, buff, xref, dref)


@pytest.mark.gpu
def test_gpu_qp_on_saturated_problems():
    """400 problems with many active bounds (tools/ltv_ws_stress.py is the same sweep on the NumPy prototype of the
    working-set rules): every one settles, and a sample is compared with the exact dense solve."""
    from alore_legged_manipulator_amd.ltv_mpc import BatchedLtvMpc, default_config
    rng = np.random.default_rng(999)
    groups = {}
    for i in range(400):
        p, case = wide_case(rng)
        groups.setdefault(p.delay_num, []).append((i, p, case))
    worst, sweeps = 0.0, []
    for delay, items in groups.items():
        B = len(items)
        cases = [it[2] for it in items]
        eng = BatchedLtvMpc(B, default_config(delay_num=delay))
        eng.set_refs(np.stack([c[3].T for c in cases]), np.stack([c[4].T for c in cases]))
        eng.set_state(np.stack([c[1].T for c in cases]), np.stack([np.stack(c[2]) for c in cases]) if delay else None)
        got = eng.get_cmd(np.array([c[0][:3] for c in cases]), n_relin=1)
        assert np.all(got["status"] == 0), (delay, np.nonzero(got["status"])[0])
        sweeps += list(got["sweeps"])
        for b in range(0, B, 6):
            now, out, buff, xref, dref = cases[b]
            ref, z, info, xbar = solve_mpcv(now, out, buff, xref, dref, items[b][1])
            worst = max(worst, float(np.max(np.abs(got["output"][b] - ref.T))))
    print(f"saturated QPs: worst |u - exact| {worst:.2e}; sweeps mean {np.mean(sweeps):.1f} max {max(sweeps)}")
    assert worst < 1e-6 and max(sweeps) <= 64


@pytest.mark.gpu
def test_gpu_get_cmd_tracks_the_oracle_over_relinearisations_and_ticks():
    from alore_legged_manipulator_amd.ltv_mpc import BatchedLtvMpc
    p = LtvParams()
    B, n_relin = 12, 4
    rng = np.random.default_rng(77)
    eng = BatchedLtvMpc(B)
    eng_tick = BatchedLtvMpc(B)   # the same ticks through the one-call entry point (alore_ltv_tick)
    specs = [(rng.uniform(0.5, 2.5), rng.uniform(-1.5, 1.5)) for _ in range(B)]
    state = np.array([[rng.uniform(-.1, .1), rng.uniform(-.1, .1), rng.uniform(-.2, .2)] for _ in range(B)])
    out_o = [np.zeros((2, p.T)) for _ in range(B)]
    buff_o = [[np.zeros(2) for _ in range(p.delay_num)] for _ in range(B)]
    for tick in range(4):
        xr, dr = [], []
        for b, (v, w) in enumerate(specs):
            ts = (tick * p.dt) + (np.arange(p.T) + 1) * p.dt
            xr.append(np.stack([v / w * np.sin(w * ts), v / w * (1 - np.cos(w * ts)), w * ts]).T)
            dr.append(np.stack([np.full(p.T, v), np.full(p.T, w)]).T)
        eng.set_refs(np.stack(xr), np.stack(dr))
        got = eng.get_cmd(state, n_relin=n_relin, reset=(tick == 0))
        assert np.all(got["status"] == 0)
        cmd_only = np.zeros((B, 2)); st_only = np.zeros(B, np.int32)
        import ctypes as C
        eng._check(eng.L.alore_ltv_commands(eng.h, B, cmd_only.ctypes.data_as(C.POINTER(C.c_double)), st_only.ctypes.data_as(C.POINTER(C.c_int)), None))
        assert np.array_equal(cmd_only, got["cmd"]) and np.all(st_only == 0)     # the commands-only download
        eng_tick.set_refs(np.stack(xr), np.stack(dr))
        cmd_t, st_t = eng_tick.tick(state, n_relin=n_relin, reset=(tick == 0))
        assert np.array_equal(cmd_t, got["cmd"]) and np.all(st_t == 0)
        for b in range(B):
            out_o[b], buff_o[b], infos = get_cmd(list(state[b]) + [0.0], out_o[b], buff_o[b], xr[b].T, dr[b].T, p, n_relin)
            assert np.max(np.abs(got["output"][b] - out_o[b].T)) < 1e-6, (tick, b)
        # plant: unicycle driven by the command for one period
        for b in range(B):
            v, w = got["cmd"][b]
            state[b] += [v * math.cos(state[b, 2]) * p.dt, v * math.sin(state[b, 2]) * p.dt, w * p.dt]


@pytest.mark.gpu
def test_gpu_refs_from_the_trajectory_store_match_the_oracle_sampler():
    from alore_legged_manipulator_amd.ltv_mpc import BatchedLtvMpc
    from alore_legged_manipulator_amd.nmpc import BatchedNmpc
    from oracle.traj_driver import RefSampler
    from tests.test_traj_oracle import golden_messages
    msgs = [m for m, _ in golden_messages()][:6]
    for m in msgs:
        m.ICR[2] = 0.0  # the `mpc` node's TrajAnal has no ICR term (mpc_controller/include/mpc_controller/traj_anal.hpp:72-73)
    B = len(msgs)
    store = BatchedNmpc(B, 20, 0.01)
    store.refs_init(max_pieces=12, max_checkpoints=80)
    store.refs_set_polynomes(np.arange(B), msgs)
    eng = BatchedLtvMpc(B)
    p = LtvParams()
    est = np.array([[m.start_position[0], m.start_position[1], m.init_pva[0] + 0.1] for m in msgs])
    now = 0.3
    eng.refs_from_store(store, now, est)
    state = est.copy()
    got = eng.get_cmd(state, n_relin=2, reset=True)
    for b, m in enumerate(msgs):
        s = RefSampler(20, 0.01); s.traj(m)
        xref, dref, _ = ref_points(s, now - m.traj_start_time, est[b, 2], p)
        out, buff, infos = get_cmd(list(state[b]) + [0.0], np.zeros((2, p.T)), [np.zeros(2)], xref, dref, p, 2)
        assert np.max(np.abs(got["output"][b] - out.T)) < 1e-5, b


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["lanes", "tick"])
def test_non_finite_state_gets_a_status_and_leaves_its_wavefront_mates_alone(kernel):
    """ltv_mpc.hip is built without NaN / Inf semantics (csrc/Makefile), so a non-finite odometry sample or reference must
    never reach its arithmetic: the robot gets status 2 and a zero command, nothing hangs, its stored output stays what it
    was, and the other robots -- the three that share its wavefront included -- return the bits of a run without it.
    Also: a NaN / Inf configuration is rejected at create."""
    from alore_legged_manipulator_amd.ltv_mpc import BatchedLtvMpc, LtvError, default_config
    p = LtvParams()
    B = 23
    rng = np.random.default_rng(11)
    cases = [random_case(rng, p) for _ in range(B)]
    xref = np.stack([c[3].T for c in cases]); dref = np.stack([c[4].T for c in cases])
    outs = np.stack([c[1].T for c in cases]); buffs = np.stack([np.stack(c[2]) for c in cases])
    now = np.array([c[0][:3] for c in cases])

    def run(now_, xref_, dref_):
        eng = BatchedLtvMpc(B, default_config())
        eng.set_refs(xref_, dref_)
        eng.set_state(outs, buffs)
        if kernel == "tick":
            cmd, st = eng.tick(now_, n_relin=2)
            return {"cmd": cmd, "status": st}, eng
        return eng.get_cmd(now_, n_relin=2), eng

    clean, _ = run(now, xref, dref)
    assert np.all(clean["status"] == 0)
    bad_now = now.copy(); bad_now[5, 2] = np.nan; bad_now[9, 0] = np.inf; bad_now[14, 1] = -np.inf
    bad_x = xref.copy(); bad_x[17, 12, 2] = np.nan
    bad_d = dref.copy(); bad_d[20, 3, 0] = np.inf
    bad = [5, 9, 14, 17, 20]
    good = [b for b in range(B) if b not in bad]
    got, eng = run(bad_now, bad_x, bad_d)
    assert np.all(got["status"][bad] == 2) and np.all(got["status"][good] == 0), got["status"]
    assert np.array_equal(got["cmd"][good], clean["cmd"][good])
    if kernel == "tick":
        assert np.all(got["cmd"][bad] == 0.0)          # what the tick publishes for a robot it could not solve
    if kernel == "lanes":
        assert np.array_equal(got["output"][bad], outs[bad])   # alore_ltv_results: the stored output, untouched
        assert np.array_equal(got["output"][good], clean["output"][good]) and np.array_equal(got["xopt"][good], clean["xopt"][good])
        # the poisoned robots kept their stored output: a finite tick afterwards is the tick of a robot that was never poisoned
        eng.set_refs(xref, dref)
        after = eng.get_cmd(now, n_relin=2)
        fresh = BatchedLtvMpc(B, default_config()); fresh.set_refs(xref, dref); fresh.set_state(outs, buffs)
        want = fresh.get_cmd(now, n_relin=2)
        assert np.all(after["status"][bad] == 0) and np.array_equal(after["output"][bad], want["output"][bad])
    for field, val in (("dt", float("nan")), ("dt", float("inf")), ("max_acc", float("nan"))):
        with pytest.raises(LtvError):
            BatchedLtvMpc(4, default_config(**{field: val}))
