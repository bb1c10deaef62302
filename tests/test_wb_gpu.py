"""Whole-body class on the GPU (include/alore_wb.h) against the float64 spatial-algebra oracle (oracle/wb_oracle.py;
no reference implementation exists: PARITY UNPINNED, the oracle is pinned by physics identities in
tests/test_wb_oracle.py).  Tolerances: the dynamics are float64 on both sides (different formulations: 1e-10);
derivatives are central differences on both sides (1e-6 relative to the block scale); the Riccati sweep runs on the
float32 matrix cores and is compared with a float64 dense KKT solve of the SAME LQ problem."""
import numpy as np
import pytest

from oracle.wb_oracle import Model, linearize, solve_lq, step
from tests.wb_cases import equilibrium_inputs, make_problems, stand_q, weights

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model():
    return Model()


def sample(model, rng, n):
    q = np.stack([np.concatenate([rng.uniform(-1, 1, 3), rng.uniform(-0.8, 0.8, 3), rng.uniform(model.lower, model.upper)]) for _ in range(n)])
    return q, rng.normal(0, 0.5, (n, 24)), rng.normal(0, 1.0, (n, 24)), rng.normal(0, 60.0, (n, 12))


def test_model_table_compiled_into_the_library(model):
    from alore_legged_manipulator_amd.whole_body import model_info
    info = model_info()
    assert np.allclose(info["masses"], model.mass) and np.allclose(info["effort"], model.effort)
    assert np.allclose(info["lower"], model.lower) and np.allclose(info["upper"], model.upper)


def test_rnea_matches_the_oracle(model):
    from alore_legged_manipulator_amd.whole_body import BatchedWholeBody
    eng = BatchedWholeBody(8)
    rng = np.random.default_rng(31)
    q, v, a, f = sample(model, rng, 96)
    tau = eng.rnea(q, v, a, f)
    tau0 = eng.rnea(q, v, a, None, gravity=False)
    for i in range(96):
        ref = model.rnea(q[i], v[i], a[i], f[i].reshape(4, 3))
        assert np.max(np.abs(tau[i] - ref)) < 1e-10 * max(1.0, np.max(np.abs(ref))), i
        assert np.max(np.abs(tau0[i] - model.rnea(q[i], v[i], a[i], None, gravity=False))) < 1e-9


def test_mass_matrix_and_forward_dynamics(model):
    from alore_legged_manipulator_amd.whole_body import BatchedWholeBody
    eng = BatchedWholeBody(8)
    rng = np.random.default_rng(32)
    q, v, _, f = sample(model, rng, 48)
    u = np.concatenate([rng.normal(0, 5, (48, 18)), f], axis=1)
    M, a = eng.forward_dynamics(q, v, u)
    for i in range(48):
        Mo = model.crba(q[i])
        assert np.max(np.abs(M[i] - Mo)) < 1e-10 * np.max(np.abs(Mo))
        ao = model.aba(q[i], v[i], u[i, :18], u[i, 18:].reshape(4, 3))      # the articulated-body algorithm: a third route
        assert np.max(np.abs(a[i] - ao)) < 1e-8 * max(1.0, np.max(np.abs(ao))), i


def test_articulated_body_algorithm_on_the_gpu(model):
    from alore_legged_manipulator_amd.whole_body import BatchedWholeBody
    eng = BatchedWholeBody(8)
    rng = np.random.default_rng(33)
    q, v, _, f = sample(model, rng, 80)
    u = np.concatenate([rng.normal(0, 6, (80, 18)), f], axis=1)
    a = eng.aba(q, v, u)
    _, a2 = eng.forward_dynamics(q, v, u)                      # the M^-1 route of the stage kernel
    for i in range(80):
        ref = model.aba(q[i], v[i], u[i, :18], u[i, 18:].reshape(4, 3))
        assert np.max(np.abs(a[i] - ref)) < 1e-9 * max(1.0, np.max(np.abs(ref))), i
    assert np.max(np.abs(a - a2)) < 1e-7 * max(1.0, np.max(np.abs(a)))
    # RNEA inverts it: base rows vanish, joint rows give the torques back
    back = eng.rnea(q, v, a, f)
    assert np.max(np.abs(back[:, :6])) < 1e-8 and np.max(np.abs(back[:, 6:] - u[:, :18])) < 1e-8


def test_linearisation_of_the_discrete_dynamics(model):
    from alore_legged_manipulator_amd.whole_body import BatchedWholeBody
    B, N, dt = 3, 4, 0.01
    eng = BatchedWholeBody(B, N, dt)
    x0, xref, uref, xi, ui = make_problems(model, B, N, seed=5)
    rng = np.random.default_rng(6)
    xi = xi + 0.02 * rng.normal(size=xi.shape); ui = ui + rng.normal(0, 2.0, ui.shape)
    eng.set_iterate(xi, ui)
    A, Bm, nxt = eng.linearize()
    for b in range(B):
        for k in (0, N - 1):
            Ao, Bo = linearize(model, xi[b, k], ui[b, k], dt)
            assert np.max(np.abs(nxt[b, k] - step(model, xi[b, k], ui[b, k], dt))) < 1e-10
            assert np.max(np.abs(A[b, k] - Ao)) < 2e-6 * max(1.0, np.max(np.abs(Ao))), (b, k, np.max(np.abs(A[b, k] - Ao)))
            assert np.max(np.abs(Bm[b, k] - Bo)) < 2e-6 * max(1.0, np.max(np.abs(Bo))), (b, k, np.max(np.abs(Bm[b, k] - Bo)))


def lq_reference(model, eng, x0, xref, uref, xi, ui, b, N):
    """float64 dense KKT solve of the LQ problem the kernel solved (its own A, B; float64 defects and gradients)"""
    Q, R, QN = weights()
    A, Bm, nxt = eng._lin
    d = [nxt[b, k] - xi[b, k + 1] for k in range(N)]
    gx = [Q * (xi[b, k] - xref[b, k]) for k in range(N)]
    gu = [R * (ui[b, k] - uref[b, k]) for k in range(N)]
    gN = QN * (xi[b, N] - xref[b, N])
    return solve_lq(list(A[b]), list(Bm[b]), d, np.diag(Q), np.diag(R), np.diag(QN), gx, gu, gN, x0[b] - xi[b, 0])


def test_riccati_step_on_the_matrix_cores_matches_a_float64_kkt_solve(model):
    from alore_legged_manipulator_amd.whole_body import BatchedWholeBody
    B, N, dt = 6, 20, 0.01
    eng = BatchedWholeBody(B, N, dt)
    x0, xref, uref, xi, ui = make_problems(model, B, N, seed=9)
    eng.set_weights(*weights())
    eng.set_problem(x0, xref, uref)
    eng.set_iterate(xi, ui)
    eng._lin = eng.linearize()
    eng.rti(1)
    dx, du = eng.last_step()
    worst_x = worst_u = 0.0
    for b in range(B):
        rx, ru = lq_reference(model, eng, x0, xref, uref, xi, ui, b, N)
        worst_x = max(worst_x, np.max(np.abs(dx[b] - rx)) / max(1e-9, np.max(np.abs(rx))))
        worst_u = max(worst_u, np.max(np.abs(du[b] - ru)) / max(1e-9, np.max(np.abs(ru))))
    print(f"float32 MFMA Riccati vs float64 KKT: rel err dx {worst_x:.2e} du {worst_u:.2e}")
    assert worst_x < 1e-4 and worst_u < 1e-4   # measured 7e-6 / 2e-5
    x1, u1 = eng.get_iterate()
    eff = model.effort
    assert np.allclose(x1, xi + dx, atol=1e-12)
    assert np.all(np.abs(u1[:, :, :18]) <= eff + 1e-12)


def test_real_time_iterations_converge_to_the_stance(model):
    from alore_legged_manipulator_amd.whole_body import BatchedWholeBody
    B, N, dt = 16, 20, 0.01
    eng = BatchedWholeBody(B, N, dt)
    x0, xref, uref, xi, ui = make_problems(model, B, N, seed=11, spread=0.5)
    Q, R, QN = weights()
    eng.set_weights(Q, R, QN)
    eng.set_problem(x0, xref, uref)
    eng.set_iterate(xi, ui)

    def cost_and_defect(x, u):
        c = 0.5 * np.sum(Q * (x[:, :-1] - xref[:, :-1]) ** 2) + 0.5 * np.sum(R * (u - uref) ** 2) + 0.5 * np.sum(QN * (x[:, -1] - xref[:, -1]) ** 2)
        dfc = max(np.max(np.abs(step(model, x[b, k], u[b, k], dt) - x[b, k + 1])) for b in (0, B - 1) for k in (0, N // 2, N - 1))
        return c / B, dfc
    steps = []
    for it in range(6):
        eng.rti(1)
        dx, du = eng.last_step()
        steps.append(float(np.max(np.abs(dx))))
    x, u = eng.get_iterate()
    c, dfc = cost_and_defect(x, u)
    print("step sizes", ["%.2e" % s for s in steps], "cost/problem %.3f defect %.2e" % (c, dfc))
    assert steps[-1] < 1e-3 * steps[0] + 1e-6     # Newton-type contraction
    assert dfc < 1e-5                               # the trajectory is dynamically consistent
    assert np.max(np.abs(x[:, 0] - x0)) < 1e-6


def test_receding_horizon_loop_holds_the_stance(model):
    """Closed loop: the plant is the same discrete dynamics driven by the GPU's articulated-body algorithm
    (alore_wb_aba), the controller one real-time iteration per tick with the iterate shifted by one stage.  From a
    perturbed pose the robot settles on the stance and stays there (2 s of simulated time; the OCP class end to end; the oracle only supplies
    the kinematic map q+ = q + dt G(q) v+)."""
    from alore_legged_manipulator_amd.whole_body import BatchedWholeBody
    B, N, dt = 8, 20, 0.01
    eng = BatchedWholeBody(B, N, dt)
    x0, xref, uref, xi, ui = make_problems(model, B, N, seed=21, spread=0.6)
    eng.set_weights(*weights())
    eng.set_iterate(xi, ui)
    x_now = x0.copy()
    err0 = np.max(np.abs(x_now[:, :24] - xref[:, 0, :24]))
    hist = []
    eng.set_problem(x_now, xref, uref)
    for tick in range(200):
        eng.set_x0(x_now)                                   # per tick: 48 doubles per robot in, 30 out
        eng.rti(2 if tick == 0 else 1)
        u0 = eng.first_input()
        if tick == 0:                                       # the tick interface moves the same data as the full one
            x, u = eng.get_iterate()
            assert np.array_equal(u0, u[:, 0])
        a = eng.aba(x_now[:, :24], x_now[:, 24:], u0)
        v_next = x_now[:, 24:] + dt * a
        q_next = np.stack([x_now[b, :24] + dt * model.qdot(x_now[b, :24], v_next[b]) for b in range(B)])
        x_now = np.concatenate([q_next, v_next], 1)
        if tick == 0:                                       # device shift = drop stage 0, repeat the last stage
            eng.shift_iterate()
            xs, us = eng.get_iterate()
            assert np.array_equal(xs, np.concatenate([x[:, 1:], x[:, -1:]], 1)) and np.array_equal(us, np.concatenate([u[:, 1:], u[:, -1:]], 1))
        else:
            eng.shift_iterate()
        assert np.isfinite(x_now).all()
        if tick % 40 == 39:
            hist.append(round(float(np.max(np.abs(x_now[:, :24] - xref[:, 0, :24]))), 4))
    print('pose error every 40 ticks:', hist)
    err = np.max(np.abs(x_now[:, :24] - xref[:, 0, :24]))
    vel = np.max(np.abs(x_now[:, 24:]))
    print(f"closed loop: pose error {err0:.3f} -> {err:.4f}, max |v| {vel:.4f}")
    # the 0.2 s horizon with these weights pulls gently: the error halves in about 1.5 s and never grows past its start
    assert err < 0.5 * err0 and vel < 0.5 and max(hist) <= 1.2 * err0, hist


def test_torque_limits_in_the_riccati_sweep(model):
    """A posture reference far from the start with stiff weights asks for more joint torque than the URDF allows: the
    Riccati kernel clamps the feed-forward step of those inputs stage by stage and re-solves the free ones (one
    projection, control-limited DDP).  Compared with the float64 restatement of the same rule fed with the kernel's own
    A, B; the applied inputs respect the limits."""
    from alore_legged_manipulator_amd.whole_body import BatchedWholeBody
    from oracle.wb_oracle import solve_lq_clamped
    B, N, dt = 4, 20, 0.01
    eng = BatchedWholeBody(B, N, dt)
    x0, xref, uref, xi, ui = make_problems(model, B, N, seed=31, spread=0.3)
    xref = xref.copy()
    xref[:, :, 18:24] += np.array([0.9, -0.8, 0.9, -0.7, 0.8, -0.9])          # arm posture far away
    Q, R, QN = weights()
    Q = Q.copy(); Q[18:24] = 4000.0; QN = 10 * Q
    eng.set_weights(Q, R, QN)
    eng.set_problem(x0, xref, uref)
    eng.set_iterate(xi, ui)
    A, Bm, nxt = eng.linearize()
    eng.rti(1)
    dx, du = eng.last_step()
    x1, u1 = eng.get_iterate()
    total_clamps, worst = 0, 0.0
    for b in range(B):
        d = [nxt[b, k] - xi[b, k + 1] for k in range(N)]
        gx = [Q * (xi[b, k] - xref[b, k]) for k in range(N)]
        gu = [R * (ui[b, k] - uref[b, k]) for k in range(N)]
        gN = QN * (xi[b, N] - xref[b, N])
        rx, ru, nc = solve_lq_clamped(list(A[b]), list(Bm[b]), d, np.diag(Q), np.diag(R), np.diag(QN), gx, gu, gN, x0[b] - xi[b, 0],
                                      list(ui[b]), model.effort)
        total_clamps += sum(nc)
        worst = max(worst, np.max(np.abs(dx[b] - rx)) / np.max(np.abs(rx)), np.max(np.abs(du[b] - ru)) / np.max(np.abs(ru)))
    print(f"clamped inputs over {B} problems x {N} stages: {total_clamps}; worst rel dev from the float64 restatement {worst:.2e}")
    assert total_clamps > 0
    assert worst < 1e-4          # the north star's tolerance (measured 5.3e-5 with 401 clamped inputs)
    assert np.all(np.abs(u1[:, :, :18]) <= model.effort + 1e-9)
    # switched off, the sweep is the unconstrained one again (and differs from the limited one on this problem)
    eng.set_torque_limits(False)
    eng.set_iterate(xi, ui)
    eng.rti(1)
    dx_u, du_u = eng.last_step()
    assert np.max(np.abs(du_u - du)) > 1.0


def test_result_of_a_problem_is_bit_reproducible_and_independent_of_its_batch_mates(model):
    """One workgroup per problem, every reduction in a fixed order (matrix instructions, DPP, v_readlane): the step of a
    problem has the same bits in a batch of 6, alone, and on a second run."""
    from alore_legged_manipulator_amd.whole_body import BatchedWholeBody
    B, N, dt = 6, 10, 0.01
    x0, xref, uref, xi, ui = make_problems(model, B, N, seed=5)
    Q, R, QN = weights()

    def run(sel):
        eng = BatchedWholeBody(len(sel), N, dt)
        eng.set_weights(Q, R, QN)
        eng.set_problem(x0[sel], xref[sel], uref[sel])
        eng.set_iterate(xi[sel], ui[sel])
        eng.rti(1)
        return eng.last_step()

    dx_all, du_all = run(list(range(B)))
    dx_again, du_again = run(list(range(B)))
    assert np.array_equal(dx_all, dx_again) and np.array_equal(du_all, du_again)
    dx_one, du_one = run([3])
    assert np.array_equal(dx_one[0], dx_all[3]) and np.array_equal(du_one[0], du_all[3])


def _lq_inputs(model, eng, xi, ui, x0, xref, uref, Q, R, QN, b, N):
    A, Bm, nxt = eng._lin
    d = [nxt[b, k] - xi[b, k + 1] for k in range(N)]
    gx = [Q * (xi[b, k] - xref[b, k]) for k in range(N)]
    gu = [R * (ui[b, k] - uref[b, k]) for k in range(N)]
    gN = QN * (xi[b, N] - xref[b, N])
    return list(A[b]), list(Bm[b]), d, np.diag(Q), np.diag(R), np.diag(QN), gx, gu, gN, x0[b] - xi[b, 0]


def _inside_contact_constraints(u, mu, stance=None, tol=1e-9):
    mu = float(np.float32(mu))            # the library keeps the coefficient in float32
    f = u[..., 18:30].reshape(u.shape[:-1] + (4, 3))
    ok = np.all(f[..., 2] >= -tol) and np.all(np.abs(f[..., 0]) <= mu * f[..., 2] + tol) and np.all(np.abs(f[..., 1]) <= mu * f[..., 2] + tol)
    if stance is not None:
        ok = ok and np.all(np.abs(f[~np.asarray(stance, bool)]) <= tol)
    return bool(ok)


def test_contact_constraints_leave_the_static_stance_alone(model):
    """Standing still on four feet with the equilibrium inputs is a fixed point of the constrained OCP as well: the
    vertical foot forces are strictly inside their friction pyramids, nothing is clamped, the step is (numerically) zero."""
    from alore_legged_manipulator_amd.whole_body import BatchedWholeBody
    B, N, dt = 2, 20, 0.01
    eng = BatchedWholeBody(B, N, dt)
    qs = stand_q(); ueq = equilibrium_inputs(model, qs)
    x = np.concatenate([qs, np.zeros(24)])
    x0 = np.tile(x, (B, 1)); xref = np.tile(x, (B, N + 1, 1)); uref = np.tile(ueq, (B, N, 1))
    eng.set_weights(*weights())
    eng.set_contact_constraints(True, 0.6)
    eng.set_problem(x0, xref, uref)
    eng.set_iterate(xref, uref)
    eng.rti(1)
    dx, du = eng.last_step()
    x1, u1 = eng.get_iterate()
    assert np.max(np.abs(dx)) < 2e-5 and np.max(np.abs(du[:, :, :18])) < 0.05 and np.max(np.abs(du[:, :, 18:])) < 0.5
    assert _inside_contact_constraints(u1, 0.6)
    assert np.max(np.abs(x1 - xref)) < 2e-5


def test_contact_constraints_keep_the_forces_in_the_friction_pyramid(model):
    """A commanded sideways shift of the base with stiff weights and a slippery floor (mu = 0.15) asks for more tangential
    force than the pyramids allow: the sweep clamps those force components stage by stage (bounds from the projected
    normal force of the stage) and re-solves the free inputs.  Compared with the float64 restatement of the same rule
    fed with the kernel's own A, B; the applied forces are inside the pyramids exactly; with the constraints off they
    are not."""
    from alore_legged_manipulator_amd.whole_body import BatchedWholeBody
    from oracle.wb_oracle import solve_lq_clamped
    B, N, dt, mu = 3, 20, 0.01, 0.15
    eng = BatchedWholeBody(B, N, dt)
    x0, xref, uref, xi, ui = make_problems(model, B, N, seed=5, spread=0.2)
    xref = xref.copy()
    xref[:, :, 1] += 0.25                                    # base 25 cm to the left, now
    Q, R, QN = weights()
    Q = Q.copy(); Q[:3] = 5000.0; QN = 10 * Q
    eng.set_weights(Q, R, QN)
    eng.set_torque_limits(False)
    eng.set_contact_constraints(True, mu)
    eng.set_problem(x0, xref, uref)
    eng.set_iterate(xi, ui)
    eng._lin = eng.linearize()
    eng.rti(1)
    dx, du = eng.last_step()
    x1, u1 = eng.get_iterate()
    total, worst = 0, 0.0
    for b in range(B):
        rx, ru, nc = solve_lq_clamped(*_lq_inputs(model, eng, xi, ui, x0, xref, uref, Q, R, QN, b, N), list(ui[b]), None, mu=mu)
        total += sum(nc)
        worst = max(worst, np.max(np.abs(dx[b] - rx)) / np.max(np.abs(rx)), np.max(np.abs(du[b] - ru)) / np.max(np.abs(ru)))
    print(f"clamped force components over {B} problems x {N} stages: {total}; worst rel dev from the float64 restatement {worst:.2e}")
    assert total > 10 and worst < 1e-4          # measured 8.3e-6 with 380 clamped components
    assert _inside_contact_constraints(u1, mu)
    eng.set_contact_constraints(False)
    eng.set_iterate(xi, ui)
    eng.rti(1)
    _, u_free = eng.get_iterate()
    assert not _inside_contact_constraints(u_free, mu, tol=1e-3)


def test_a_swing_foot_carries_no_force(model):
    """Contact schedule: the front-right foot leaves the ground from stage 6 on.  Its force inputs are exactly zero there in
    the applied iterate, the other three feet take the weight (inside their pyramids), and the step matches the float64
    restatement with the same schedule."""
    from alore_legged_manipulator_amd.whole_body import BatchedWholeBody
    from oracle.wb_oracle import solve_lq_clamped
    B, N, dt, mu = 2, 20, 0.01, 0.7
    eng = BatchedWholeBody(B, N, dt)
    x0, xref, uref, xi, ui = make_problems(model, B, N, seed=9, spread=0.2)
    stance = np.ones((B, N, 4), np.uint8)
    stance[:, 6:, 1] = 0
    Q, R, QN = weights()
    eng.set_weights(Q, R, QN)
    eng.set_torque_limits(False)
    eng.set_contact_constraints(True, mu)
    eng.set_contact_schedule(stance)
    eng.set_problem(x0, xref, uref)
    eng.set_iterate(xi, ui)
    eng._lin = eng.linearize()
    eng.rti(1)
    dx, du = eng.last_step()
    x1, u1 = eng.get_iterate()
    assert _inside_contact_constraints(u1, mu, stance)
    f = u1[:, :, 18:30].reshape(B, N, 4, 3)
    assert np.all(f[:, 6:, 1] == 0.0) and np.all(f[:, 6:, [0, 2, 3], 2].sum(-1) > 300.0)   # three feet carry the robot
    worst = 0.0
    for b in range(B):
        rx, ru, nc = solve_lq_clamped(*_lq_inputs(model, eng, xi, ui, x0, xref, uref, Q, R, QN, b, N), list(ui[b]), None, mu=mu,
                                      stance=stance[b])
        assert sum(nc) >= 3 * (N - 6)
        worst = max(worst, np.max(np.abs(dx[b] - rx)) / np.max(np.abs(rx)), np.max(np.abs(du[b] - ru)) / np.max(np.abs(ru)))
    print(f"swing foot: worst rel dev from the float64 restatement {worst:.2e}")
    assert worst < 1e-4, worst
    eng.set_contact_schedule(None)                     # back to four stance feet
    eng.set_iterate(xi, ui)
    eng.rti(1)
    _, u2 = eng.get_iterate()
    assert np.all(u2[:, 6:, 18 + 3 + 2] > 50.0)


def test_configs2_full_size_batch():
    """BASELINE configs[2] at its full size in a test: B = 4096 problems, N = 20, one real-time iteration with torque
    limits and contact constraints on.  Size-independent checks: every problem reports a finite, applied step; sampled
    problems have the same bits as the same problems solved in a batch of their own (one workgroup per problem, fixed
    reduction order); the applied inputs respect the torque limits and the friction pyramids; a second iteration
    contracts the step."""
    from alore_legged_manipulator_amd.whole_body import BatchedWholeBody, model_info
    from tests.wb_cases import make_problems_fast
    B, N, dt, mu = 4096, 20, 0.01, 0.6
    x0, xref, uref, xi, ui = make_problems_fast(B, N, seed=3)
    Q, R, QN = weights()
    eng = BatchedWholeBody(B, N, dt)
    eng.set_weights(Q, R, QN)
    eng.set_contact_constraints(True, mu)
    eng.set_problem(x0, xref, uref)
    eng.set_iterate(xi, ui)
    eng.rti(1)
    assert np.all(eng.status() == 0)
    dx, du = eng.last_step()
    x1, u1 = eng.get_iterate()
    assert np.isfinite(dx).all() and np.isfinite(du).all() and np.isfinite(x1).all() and np.isfinite(u1).all()
    effort = model_info()["effort"]
    assert np.all(np.abs(u1[:, :, :18]) <= effort + 1e-9)
    assert _inside_contact_constraints(u1, mu)
    idx = [0, 1, 63, 64, 2048, 4095]
    small = BatchedWholeBody(len(idx), N, dt)
    small.set_weights(Q, R, QN)
    small.set_contact_constraints(True, mu)
    small.set_problem(x0[idx], xref[idx], uref[idx])
    small.set_iterate(xi[idx], ui[idx])
    small.rti(1)
    sdx, sdu = small.last_step()
    assert np.array_equal(sdx, dx[idx]) and np.array_equal(sdu, du[idx])
    eng.rti(1)
    dx2, _ = eng.last_step()
    assert np.all(eng.status() == 0)
    assert np.median(np.abs(dx2).max(axis=(1, 2))) < 0.5 * np.median(np.abs(dx).max(axis=(1, 2)))


def _stance_foot_speed(model, x, stance=None):
    """|J_c(q_k) v_k| of the stance feet along one trajectory x [N+1][48], per stage"""
    out = []
    for k in range(x.shape[0]):
        J = model.contact_jacobian(x[k, :24])
        w = J @ x[k, 24:]
        if stance is not None and k < len(stance):
            w = w * np.repeat(np.asarray(stance[k], float), 3)
        out.append(np.linalg.norm(w))
    return np.array(out)


def test_contact_penalty_step_matches_the_float64_restatement(model):
    """alore_wb_set_contact_penalty: the stage cost gains 1/2 rho |J_c(q_k) v_k|^2 over the stance feet (the constraint
    Jacobian of 'stance feet do not move', Gauss-Newton).  The LQ step of the kernels (J_c from the stage kernel's foot-force
    columns, rho J_c' J_c added to the cost-to-go tiles on the matrix cores, float32) against a float64 dense KKT solve of the
    same problem with the oracle's own J_c (a different formulation of the kinematics); one foot swings from stage 6 on and
    must carry no penalty there."""
    from alore_legged_manipulator_amd.whole_body import BatchedWholeBody
    from oracle.wb_oracle import contact_penalty, solve_lq
    B, N, dt = 3, 20, 0.01
    x0, xref, uref, xi, ui = make_problems(model, B, N, seed=21, spread=0.5)
    stance = np.ones((B, N, 4), np.uint8)
    stance[:, 6:, 2] = 0
    Q, R, QN = weights()
    # One float32 sweep loses accuracy with the stiffness of the penalty (rho J_c' J_c of 1e3 against velocity weights of
    # 0.5 - 2: 1.5e-4 at rho = 200, 1.5e-3 at rho = 2000); with one step of iterative refinement (float64 residuals through the
    # same sweep, alore_wb_set_refinement) the step meets the north star's 1e-4 at both (measured 1.7e-6 / 8.2e-6).  At
    # rho = 20000 two steps reach 1.2e-4 and a third changes nothing: that is the float32 storage of sqrt(rho) J_c and of
    # A, B against the oracle's float64 Jacobian, not the sweep.
    for rho, refine, tol in ((200.0, 0, 1e-3), (2000.0, 0, 5e-3), (200.0, 1, 1e-4), (2000.0, 1, 1e-4), (20000.0, 2, 3e-4)):
        eng = BatchedWholeBody(B, N, dt)
        eng.set_weights(Q, R, QN)
        eng.set_torque_limits(False)
        eng.set_refinement(refine)
        eng.set_contact_penalty(rho)
        eng.set_contact_schedule(stance)
        eng.set_problem(x0, xref, uref)
        eng.set_iterate(xi, ui)
        eng._lin = eng.linearize()
        eng.rti(1)
        dx, du = eng.last_step()
        assert (eng.status() == 0).all()
        worst = 0.0
        for b in range(B):
            A, Bm, d, Qd, Rd, QNd, gx, gu, gN, dx0 = _lq_inputs(model, eng, xi, ui, x0, xref, uref, Q, R, QN, b, N)
            Ql, gl = [], []
            for k in range(N):
                Qa, ga = contact_penalty(model, xi[b, k], rho, stance[b, k])
                Ql.append(Qd + Qa); gl.append(gx[k] + ga)
            rx, ru = solve_lq(A, Bm, d, Ql, Rd, QNd, gl, gu, gN, dx0)
            worst = max(worst, np.max(np.abs(dx[b] - rx)) / np.max(np.abs(rx)), np.max(np.abs(du[b] - ru)) / np.max(np.abs(ru)))
            # and the penalty does change the step: without it the restatement is somewhere else
            fx, fu = solve_lq(A, Bm, d, Qd, Rd, QNd, gx, gu, gN, dx0)
            assert np.max(np.abs(fx - rx)) > 50 * np.max(np.abs(dx[b] - rx))
        print(f"contact penalty rho = {rho}, refinement {refine}: worst rel dev of the LQ step from the float64 KKT solve {worst:.2e}")
        assert worst < tol


def test_contact_rows_match_the_float64_kkt_solve_and_hold_the_feet(model):
    """alore_wb_set_contact_rows: J_c(q_k) v_{k+1} = 0 for the stance feet as equality rows, solved for the foot forces (the
    forces leave the dynamics, the sweep runs on (dx, dtau), the forces follow).  The step of the kernels against the float64
    dense KKT system with the same rows built from the ORACLE's contact Jacobian (oracle/wb_oracle.py:
    solve_lq_contact_rows), 1e-4; the linearised foot-point velocities after the step vanish; a swing foot carries no force and
    is free to move; repeated iterations settle with the stance feet at rest."""
    from alore_legged_manipulator_amd.whole_body import BatchedWholeBody
    from oracle.wb_oracle import solve_lq_contact_rows
    B, N, dt = 3, 20, 0.01
    x0, xref, uref, xi, ui = make_problems(model, B, N, seed=41, spread=0.5)
    stance = np.ones((B, N, 4), np.uint8)
    stance[:, 6:, 1] = 0
    Q, R, QN = weights()
    eng = BatchedWholeBody(B, N, dt)
    eng.set_weights(Q, R, QN)
    eng.set_torque_limits(False)
    eng.set_contact_rows(True)
    eng.set_contact_schedule(stance)
    eng.set_problem(x0, xref, uref)
    eng.set_iterate(xi, ui)
    eng._lin = eng.linearize()
    eng.rti(1)
    assert (eng.status() == 0).all()
    dx, du = eng.last_step()
    x1, u1 = eng.get_iterate()
    worst, rest = 0.0, 0.0
    for b in range(B):
        A, Bm, d, Qd, Rd, QNd, gx, gu, gN, dx0 = _lq_inputs(model, eng, xi, ui, x0, xref, uref, Q, R, QN, b, N)
        J = [model.contact_jacobian(xi[b, k, :24]) for k in range(N)]
        vn = [eng._lin[2][b, k, 24:] for k in range(N)]
        rx, ru = solve_lq_contact_rows(A, Bm, d, Qd, Rd, QNd, gx, gu, gN, dx0, J, vn, [ui[b, k, 18:30] for k in range(N)], stance[b])
        worst = max(worst, np.max(np.abs(dx[b] - rx)) / np.max(np.abs(rx)), np.max(np.abs(du[b] - ru)) / np.max(np.abs(ru)))
        for k in range(N):
            w = J[k] @ (vn[k] + (A[k] @ dx[b, k] + Bm[k] @ du[b, k])[24:])
            w = w * np.repeat(stance[b, k].astype(float), 3)
            rest = max(rest, np.max(np.abs(w)))
        # without the rows the step is somewhere else: the rows do bind
        from oracle.wb_oracle import solve_lq
        fx, fu = solve_lq(A, Bm, d, Qd, Rd, QNd, gx, gu, gN, dx0)
        assert np.max(np.abs(fx - rx)) > 100 * np.max(np.abs(dx[b] - rx))
    print(f"contact rows: worst rel dev of the LQ step from the float64 KKT solve {worst:.2e}; linearised stance-foot speed after the step {rest:.2e} m/s")
    assert worst < 1e-4 and rest < 1e-5
    assert np.all(u1[:, 6:, 18 + 3:18 + 6] == 0.0)                 # the swing foot carries no force
    for _ in range(4):
        eng.rti(1)
    assert (eng.status() == 0).all()
    x5, _ = eng.get_iterate()
    dxl, _ = eng.last_step()
    assert np.max(np.abs(dxl)) < 1e-2
    # the rows are J_c(q_k) v_{k+1} = 0: the Jacobian of stage k on the velocity the step ends with
    held = np.array([max(np.max(np.abs((model.contact_jacobian(x5[b, k, :24]) @ x5[b, k + 1, 24:]) * np.repeat(stance[b, k].astype(float), 3)))
                         for k in range(N)) for b in range(B)])
    same_stage = np.array([_stance_foot_speed(model, x5[b], np.vstack([np.ones((1, 4)), stance[b]]))[1:].max() for b in range(B)])
    print(f"settled iterate: |J_c(q_k) v_k+1| of the stance feet (max over stages) {held}; with the Jacobian of the same stage {same_stage}")
    assert np.all(held < 1e-4) and np.all(same_stage < 5e-3)


def test_contact_penalty_keeps_the_stance_feet_still(model):
    """Started with random joint rates, the posture cost alone lets the stance feet slide while the robot settles; with the
    penalty on J_c v the foot-point speeds of the returned trajectory drop by an order of magnitude (soft constraint:
    ~1 / rho), the swing foot stays free to move, and the iteration still contracts."""
    from alore_legged_manipulator_amd.whole_body import BatchedWholeBody
    B, N, dt = 2, 20, 0.01
    x0, xref, uref, xi, ui = make_problems(model, B, N, seed=33, spread=0.6)
    stance = np.ones((B, N, 4), np.uint8)
    stance[:, :, 3] = 0
    Q, R, QN = weights()
    res = {}
    for rho in (0.0, 5000.0):
        eng = BatchedWholeBody(B, N, dt)
        eng.set_weights(Q, R, QN)
        eng.set_torque_limits(False)
        eng.set_contact_penalty(rho)
        eng.set_contact_schedule(stance)
        eng.set_problem(x0, xref, uref)
        eng.set_iterate(xi, ui)
        for _ in range(4):
            eng.rti(1)
        assert (eng.status() == 0).all()
        x1, _ = eng.get_iterate()
        dxl, _ = eng.last_step()
        res[rho] = (np.array([_stance_foot_speed(model, x1[b], stance[b])[2:].mean() for b in range(B)]), np.max(np.abs(dxl)), x1)
    free, held = res[0.0][0], res[5000.0][0]
    print(f"mean stance-foot speed over stages 2..N: free {free}, penalised {held}; last steps {res[0.0][1]:.2e} / {res[5000.0][1]:.2e}")
    assert np.all(held < 0.1 * free) and np.all(free > 0.02)
    assert res[5000.0][1] < 1e-2                       # the Gauss-Newton iteration has settled with the penalty too
    # the swing foot is not held: its point still moves in the penalised solution
    J = model.contact_jacobian(res[5000.0][2][0, 3, :24])
    assert np.linalg.norm((J @ res[5000.0][2][0, 3, 24:])[9:12]) > 5 * held[0]


def _oracle_lq_inputs(model, xi, ui, x0, xref, uref, Q, R, QN, b, N, dt):
    """The LQ problem of one real-time iteration from the ORACLE's own float64 dynamics and linearisation (oracle/wb_oracle.py:
    step, linearize) -- nothing of the kernels in it."""
    from oracle.wb_oracle import linearize, step
    A, Bm, d = [], [], []
    for k in range(N):
        a_, b_ = linearize(model, xi[b, k], ui[b, k], dt)
        A.append(a_); Bm.append(b_)
        d.append(step(model, xi[b, k], ui[b, k], dt) - xi[b, k + 1])
    gx = [Q * (xi[b, k] - xref[b, k]) for k in range(N)]
    gu = [R * (ui[b, k] - uref[b, k]) for k in range(N)]
    gN = QN * (xi[b, N] - xref[b, N])
    return A, Bm, d, np.diag(Q), np.diag(R), np.diag(QN), gx, gu, gN, x0[b] - xi[b, 0]


@pytest.mark.parametrize("case", ["torque_limits", "friction_pyramid", "swing_foot"])
def test_exact_working_set_mode_finds_the_minimiser_of_the_inequality_constrained_lq_problem(model, case):
    """alore_wb_set_constraint_mode(exact): the working-set iteration around the unconstrained MFMA sweep against the float64
    active-set solve of the inequality-constrained QP (oracle/wb_oracle.py: solve_lq_inequality, KKT residuals checked) built from
    the ORACLE's linearisation -- on the three scenarios of the projection tests above: joint torques at their limits, tangential
    forces on the faces of a slippery friction pyramid, a foot leaving the ground.  The step must match to 1e-4 (relative to the
    largest component), the working set must have settled, the applied inputs satisfy the constraints exactly.  Printed beside it:
    how far the default mode (one projection per stage inside the sweep) is from that minimiser on the same problem."""
    from alore_legged_manipulator_amd.whole_body import BatchedWholeBody
    from oracle.wb_oracle import solve_lq_inequality
    B, N, dt = 2, 20, 0.01
    Q, R, QN = weights()
    Q = Q.copy()
    mu, stance, effort = None, None, None
    if case == "torque_limits":
        x0, xref, uref, xi, ui = make_problems(model, B, N, seed=31, spread=0.3)
        xref = xref.copy(); xref[:, :, 18:24] += np.array([0.9, -0.8, 0.9, -0.7, 0.8, -0.9])
        Q[18:24] = 4000.0; effort = model.effort
    elif case == "friction_pyramid":
        x0, xref, uref, xi, ui = make_problems(model, B, N, seed=5, spread=0.2)
        xref = xref.copy(); xref[:, :, 1] += 0.25
        Q[:3] = 5000.0; mu = 0.15
    else:
        x0, xref, uref, xi, ui = make_problems(model, B, N, seed=9, spread=0.2)
        mu = 0.7
        stance = np.ones((B, N, 4), np.uint8); stance[:, 6:, 1] = 0
    QN = 10 * Q if case != "swing_foot" else QN
    if mu is not None:   # a feasible start for the force constraints: the library projects the applied forces, so does the test's iterate
        f = ui[:, :, 18:30].reshape(B, N, 4, 3).copy()
        if stance is not None:
            f[~stance.astype(bool)] = 0.0
        f[..., 2] = np.maximum(f[..., 2], 0.0)
        m32 = float(np.float32(mu))
        f[..., 0] = np.clip(f[..., 0], -m32 * f[..., 2], m32 * f[..., 2]); f[..., 1] = np.clip(f[..., 1], -m32 * f[..., 2], m32 * f[..., 2])
        ui = ui.copy(); ui[:, :, 18:30] = f.reshape(B, N, 12)

    def run(exact):
        eng = BatchedWholeBody(B, N, dt)
        eng.set_weights(Q, R, QN)
        eng.set_torque_limits(effort is not None)
        if mu is not None:
            eng.set_contact_constraints(True, mu)
        if stance is not None:
            eng.set_contact_schedule(stance)
        eng.set_constraint_mode(exact, 1000)
        eng.set_problem(x0, xref, uref)
        eng.set_iterate(xi, ui)
        eng.rti(1)
        dx, du = eng.last_step()
        _, u1 = eng.get_iterate()
        info = eng.working_set_info(B) if exact else None
        assert (eng.status()[:B] == 0).all()
        return dx, du, u1, info

    dx_e, du_e, u_e, (sweeps, changed, ws, gu) = run(True)
    dx_p, du_p, u_p, _ = run(False)
    worst_e, worst_p, active = 0.0, 0.0, 0
    mu_lib = None if mu is None else float(np.float32(mu))
    for b in range(B):
        rx, ru, info = solve_lq_inequality(*_oracle_lq_inputs(model, xi, ui, x0, xref, uref, Q, R, QN, b, N, dt), ui[b], effort, mu_lib,
                                           None if stance is None else stance[b])
        k = info["kkt"]
        assert k["stationarity"] < 1e-8 * max(1.0, k["scale"]) and k["feasibility"] < 1e-9 and k["dual"] < 1e-9, info
        active += info["active"]
        sx, su = np.max(np.abs(rx)), np.max(np.abs(ru))
        worst_e = max(worst_e, np.max(np.abs(dx_e[b] - rx)) / sx, np.max(np.abs(du_e[b] - ru)) / su)
        worst_p = max(worst_p, np.max(np.abs(dx_p[b] - rx)) / sx, np.max(np.abs(du_p[b] - ru)) / su)
    held = int((ws[:, :, :30] != 0).sum())
    print(f"{case}: {active} active rows (inequalities at their bound, zero-force equalities) in the float64 optimum, {held} inputs held by the kernel's working set after {sweeps} sweeps "
          f"(changes in the last: {changed.tolist()}); rel dev from the optimum: exact mode {worst_e:.2e}, one projection per stage {worst_p:.2e}")
    assert active > 0
    assert (changed == 0).all(), (sweeps, changed)
    assert worst_e < 1e-4, worst_e
    if effort is not None:
        assert np.all(np.abs(u_e[:, :, :18]) <= model.effort + 1e-9)
    if mu is not None:
        assert _inside_contact_constraints(u_e, mu, stance)
