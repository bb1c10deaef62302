"""Parity of the HIP path (through the C ABI) with the CPU oracle and with the
fixtures generated from the compiled reference.  Needs a real MI355X.

Tolerances (float32 path, BASELINE.json: trajectories within 1e-4 relative of
the CPU reference): per RTI tick from identical inputs, max-abs error relative
to max(1, largest entry) < 1e-4 for x and u, < 1e-3 for the multipliers; the
linearisation (d, Gx, Gu) < 2e-6 absolute.  The reference's own QP solutions
carry errors of a few 1e-5 (float32 homotopy), which is the floor."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from alore_legged_manipulator_amd.scenarios import make_batch, problem  # noqa: E402
from oracle.drivers import Oracle  # noqa: E402

IN_KEYS = ("x", "u", "od", "y", "yN", "W", "WN", "x0", "lbValues", "ubValues", "dual")


def relerr(a, b):
    """element-wise relative error with an absolute floor of 1: max_i |a_i - b_i| / max(1, |b_i|)  (never smaller than
    the vector-norm figure max|a - b| / max(1, max|b|), which `relerr_vec` reports)"""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))


def relerr_vec(a, b):
    return float(np.max(np.abs(a - b)) / max(1.0, np.max(np.abs(b))))


def exact_box_qp(H, g, lb, ub):
    """float64 minimiser of 0.5 x'Hx + g'x in [lb, ub]: bounded-variable least squares on the Cholesky factor
    (scipy.optimize.lsq_linear, BVLS: an exact active-set method)."""
    from scipy.optimize import lsq_linear
    H = np.asarray(H, np.float64); H = 0.5 * (H + H.T)
    L = np.linalg.cholesky(H)
    rhs = -np.linalg.solve(L, np.asarray(g, np.float64))
    r = lsq_linear(L.T, rhs, bounds=(np.asarray(lb, np.float64), np.asarray(ub, np.float64)), method="bvls", tol=1e-15, max_iter=2000)
    return r.x


def oracle_tick(orc, p):
    orc.reset()
    orc.initialize_solver()
    orc.load(p)
    orc.preparation_step()
    st = orc.feedback_step()
    return st, {k: orc.v[k].copy() for k in ("x", "u", "dual", "dx", "d", "evGx", "evGu")}, orc.get_kkt()


@pytest.fixture(scope="module")
def nmpc_mod():
    from alore_legged_manipulator_amd import nmpc
    return nmpc


@pytest.mark.parametrize("N,L", [(20, 0), (20, 4), (20, 8), (20, 16), (20, 32), (20, 64), (50, 0), (50, 32), (50, 16), (7, 0), (1, 0)])
def test_one_tick_matches_oracle(nmpc_mod, N, L):
    B = 96 if N <= 20 else 40
    batch = make_batch(B, N, seed=1234 + N, fast_tail=0.3)
    eng = nmpc_mod.BatchedNmpc(B, N, lanes_per_problem=L)
    eng.load(batch)
    lin = eng.linearize()
    eng.rti(1)
    out = eng.fetch()
    orc = Oracle(N)
    for b in range(B):
        st, ref, kkt = oracle_tick(orc, problem(batch, b))
        assert st == 0 and out["status"][b] == 0
        assert np.max(np.abs(lin["d"][b].reshape(-1) - ref["d"])) < 2e-6
        assert np.max(np.abs(lin["evGx"][b].reshape(-1) - ref["evGx"])) < 2e-6
        assert np.max(np.abs(lin["evGu"][b].reshape(-1) - ref["evGu"])) < 2e-6
        assert relerr(out["x"][b].reshape(-1), ref["x"]) < 1e-4, (b, "x")
        assert relerr(out["u"][b].reshape(-1), ref["u"]) < 1e-4, (b, "u")
        assert relerr(out["dual"][b].reshape(-1), ref["dual"]) < 1e-3, (b, "dual")
        assert abs(out["kkt"][b] - kkt) <= 2e-3 * max(1.0, kkt), (b, out["kkt"][b], kkt)
        assert 1 <= out["n_iter"][b] <= 16


@pytest.mark.parametrize("N", [20, 50])
def test_sqp_iterations_in_kernel(nmpc_mod, N):
    """K real-time iterations inside one launch == K ticks of the oracle (feasible
    references; the full-step Gauss-Newton iteration is contractive there)."""
    B, K = 32, 15
    batch = make_batch(B, N, seed=99, fast_tail=0.0)
    eng = nmpc_mod.BatchedNmpc(B, N)
    eng.load(batch)
    eng.rti(K)
    out = eng.fetch()
    orc = Oracle(N)
    for b in range(B):
        orc.reset()
        orc.initialize_solver()
        orc.load(problem(batch, b))
        for _ in range(K):
            orc.preparation_step()
            assert orc.feedback_step() == 0
        assert out["status"][b] == 0
        assert relerr(out["x"][b].reshape(-1), orc.v["x"]) < 1e-4
        assert relerr(out["u"][b].reshape(-1), orc.v["u"]) < 1e-4
        obj = orc.get_objective()
        assert abs(out["obj"][b] - obj) <= 1e-4 * max(1.0, abs(obj))


def test_golden_n50_ticks(nmpc_mod, golden_dir):
    """All 10 scenarios x 15 ticks of the compiled reference (tests/golden/nmpc_n50.npz),
    one batch per tick, each tick started from the reference's own iterate."""
    G = np.load(os.path.join(golden_dir, "nmpc_n50.npz"))
    S, K = int(G["n_scen"]), int(G["K"])
    eng = nmpc_mod.BatchedNmpc(S, 50)
    cur = {k: np.stack([G[f"s{s}_in_{k}"] for s in range(S)]) for k in IN_KEYS}
    for it in range(K):
        eng.load(cur)
        eng.rti(1)
        out = eng.fetch()
        for s in range(S):
            assert out["status"][s] == 0
            assert relerr(out["x"][s].reshape(-1), G[f"s{s}_x"][it]) < 1e-4, (s, it)
            assert relerr(out["u"][s].reshape(-1), G[f"s{s}_u"][it]) < 1e-4, (s, it)
            assert relerr(out["dual"][s].reshape(-1), G[f"s{s}_dual"][it]) < 1e-3, (s, it)
            assert abs(out["kkt"][s] - G[f"s{s}_kkt"][it]) <= 2e-3 * max(1.0, G[f"s{s}_kkt"][it])
        cur = dict(cur, x=np.stack([G[f"s{s}_x"][it] for s in range(S)]),
                   u=np.stack([G[f"s{s}_u"][it] for s in range(S)]),
                   dual=np.stack([G[f"s{s}_dual"][it] for s in range(S)]))


def test_golden_n20_embedded(nmpc_mod, golden_dir):
    G = np.load(os.path.join(golden_dir, "nmpc_n20_embedded.npz"))
    S, K, N = int(G["n_scen"]), int(G["K"]), int(G["N"])
    eng = nmpc_mod.BatchedNmpc(S, N)
    cur = {k: np.stack([G[f"s{s}_in_{k}"] for s in range(S)]) for k in IN_KEYS}
    for it in range(K):
        eng.load(cur)
        eng.rti(1)
        out = eng.fetch()
        for s in range(S):
            assert out["status"][s] == G[f"s{s}_status"][it] == 0
            assert relerr(out["x"][s].reshape(-1), G[f"s{s}_x"][it]) < 1e-4
            assert relerr(out["u"][s].reshape(-1), G[f"s{s}_u"][it]) < 1e-4
        cur = dict(cur, x=np.stack([G[f"s{s}_x"][it] for s in range(S)]),
                   u=np.stack([G[f"s{s}_u"][it] for s in range(S)]),
                   dual=np.stack([G[f"s{s}_dual"][it] for s in range(S)]))


def test_ragged_batch_and_full_size_properties(nmpc_mod):
    """B = 4096 (BASELINE config 2) and a batch size that is not a multiple of the
    problems per workgroup.  Size-independent properties: every problem of the
    big batch equals the same problem solved in a small batch (bitwise, the
    kernel is deterministic and problems are independent); the returned
    controls respect the bounds; KKT values are finite."""
    N = 20
    big = make_batch(4096, N, fast_tail=0.05)
    eng = nmpc_mod.BatchedNmpc(4096, N)
    eng.load(big)
    eng.rti(1)
    out = eng.fetch()
    assert (out["status"] == 0).all()
    assert np.isfinite(out["kkt"]).all() and np.isfinite(out["obj"]).all()
    assert (out["u"] <= big["ubValues"] + 1e-6).all() and (out["u"] >= big["lbValues"] - 1e-6).all()
    idx = np.array([0, 1, 63, 64, 777, 2048, 4094, 4095, 3001, 17, 18])  # 11 problems: ragged last block
    small = {k: v[idx] for k, v in big.items()}
    eng2 = nmpc_mod.BatchedNmpc(len(idx), N, lanes_per_problem=eng.launch_info()["lanes_per_problem"])
    eng2.load(small)
    eng2.rti(1)
    out2 = eng2.fetch()
    for k in ("x", "u", "dual"):
        np.testing.assert_array_equal(out[k][idx], out2[k])
    # and against the oracle on a sample
    orc = Oracle(N)
    for b in idx:
        st, ref, _ = oracle_tick(orc, problem(big, int(b)))
        assert st == 0
        assert relerr(out["u"][b].reshape(-1), ref["u"]) < 1e-4


def test_idempotent_at_fixed_point(nmpc_mod):
    """At a converged iterate with x0 = x[0] a further tick must not move (property)."""
    N, B = 20, 64
    batch = make_batch(B, N, seed=5, fast_tail=0.0)
    eng = nmpc_mod.BatchedNmpc(B, N)
    eng.load(batch)
    eng.rti(30)
    a = eng.fetch()
    eng.load({"x0": a["x"][:, 0, :]})
    eng.rti(1)
    b = eng.fetch()
    assert np.max(np.abs(a["u"] - b["u"])) < 5e-5
    assert np.max(np.abs(a["x"] - b["x"])) < 5e-5


def test_edge_bounds(nmpc_mod):
    """Equality bounds (lb == ub) pin a control; lb > ub is reported as infeasible (33)."""
    N, B = 20, 8
    batch = make_batch(B, N, seed=8)
    batch["lbValues"][0, 3, 0] = 0.5
    batch["ubValues"][0, 3, 0] = 0.5
    batch["lbValues"][1, 2, 1] = 1.0
    batch["ubValues"][1, 2, 1] = -1.0
    eng = nmpc_mod.BatchedNmpc(B, N)
    eng.load(batch)
    eng.rti(1)
    out = eng.fetch()
    assert out["status"][0] == 0 and abs(out["u"][0, 3, 0] - 0.5) < 1e-6
    assert out["status"][1] == 33
    assert (out["status"][2:] == 0).all()
    orc = Oracle(N)
    st, ref, _ = oracle_tick(orc, problem(batch, 0))
    assert st == 0 and relerr(out["u"][0].reshape(-1), ref["u"]) < 1e-4


def test_shift_and_forward_sim_match_oracle(nmpc_mod):
    N, B = 20, 16
    batch = make_batch(B, N, seed=21)
    rng = np.random.default_rng(2)
    batch["u"] = rng.uniform(-2, 2, batch["u"].shape).astype(np.float32)
    eng = nmpc_mod.BatchedNmpc(B, N)
    eng.load(batch)
    eng.forward_simulate()
    uend = rng.uniform(-1, 1, (B, 2)).astype(np.float32)
    eng.shift(2, None, uend)
    out = eng.fetch(("x", "u"))
    orc = Oracle(N)
    for b in range(B):
        orc.reset(); orc.initialize_solver(); orc.load(problem(batch, b))
        orc.initialize_nodes_by_forward_simulation()
        orc.shift_states(2, None, uend[b])
        orc.shift_controls(uend[b])
        assert np.max(np.abs(out["x"][b].reshape(-1) - orc.v["x"])) < 5e-6
        np.testing.assert_array_equal(out["u"][b].reshape(-1), orc.v["u"])


def test_mpc_wrapper_mirror(nmpc_mod):
    """BatchedMpcWrapper (mirror of Tracked_nmpc::MpcWrapper): setCosts /
    setICRParameters / setTrajectory / solve, against the oracle driven the way
    the reference wrapper drives ACADO."""
    N, B = 20, 12
    batch = make_batch(B, N, seed=314)
    w = nmpc_mod.BatchedMpcWrapper(B, N, Q=np.diag([10, 10, 0.5]), R=np.diag([0.1, 0.1]))
    w.setICRParameters(batch["od"][:, 0, :])
    states = np.concatenate([batch["y"][:, :, :3], batch["yN"][:, None, :]], 1).transpose(0, 2, 1)
    inputs = np.concatenate([batch["y"][:, :, 3:], batch["y"][:, -1:, 3:]], 1).transpose(0, 2, 1)
    w.setTrajectory(states, inputs)
    assert w.solve(batch["x0"])
    U = w.getInputs()
    X = w.getStates()
    assert U.shape == (B, 2, N) and X.shape == (B, 3, N + 1) and U.dtype == np.float64
    orc = Oracle(N)
    for b in range(B):
        st, ref, _ = oracle_tick(orc, problem(batch, b))
        assert relerr(U[b].T.reshape(-1).astype(np.float32), ref["u"]) < 1e-4
        assert relerr(X[b].T.reshape(-1).astype(np.float32), ref["x"]) < 1e-4


@pytest.mark.gpu
def test_status_codes_indefinite_and_iteration_cap(nmpc_mod):
    """qpOASES return values on the failure paths the reference can take (acado_feedbackStep's return value,
    ignored at mpc_wrapper.cpp:298 but kept in batch.status here): 31 when the Hessian is not positive
    definite (QProblemB::setupCholeskyDecomposition), 58 when the working set is still changing at the
    iteration cap (nWSR exhausted); healthy neighbours in the same wavefront are unaffected."""
    N, B = 20, 8
    batch = make_batch(B, N, seed=77, fast_tail=1.0)        # every problem hits its bounds
    # problem 3: negative control weight and no state weight -> indefinite QP
    batch["W"][3] = 0.0
    batch["W"][3, :, 3, 3] = -1.0
    batch["W"][3, :, 4, 4] = -1.0
    batch["WN"][3] = 0.0
    eng = nmpc_mod.BatchedNmpc(B, N)
    eng.load(batch)
    eng.rti(1)
    out = eng.fetch()
    assert out["status"][3] == 31
    ok = np.arange(B) != 3
    assert (out["status"][ok] == 0).all()
    orc = Oracle(N)
    st, _, _ = oracle_tick(orc, problem(batch, 3))
    assert st == 31                                          # the reference's own code says the same
    for b in (2, 4):                                         # neighbours of the bad problem, same wavefront
        st, ref, _ = oracle_tick(orc, problem(batch, b))
        assert st == 0 and relerr(out["u"][b].reshape(-1), ref["u"]) < 1e-4
    # iteration cap: no prediction, one sweep allowed -> bound-hitting problems cannot finish
    eng1 = nmpc_mod.BatchedNmpc(B, N, max_as_iter=1, warm_start_steps=0)
    eng1.load(batch)
    eng1.rti(1)
    o1 = eng1.fetch()
    assert (o1["status"][ok] == 58).sum() >= 1 and set(np.unique(o1["status"][ok])) <= {0, 58}
    assert np.isfinite(o1["x"]).all() and np.isfinite(o1["u"]).all()


@pytest.mark.gpu
def test_config3_whole_batch_on_one_gpu(nmpc_mod):
    """BASELINE configs[3] (262144 problems, the 8-GPU batch) as ONE launch: every problem solved, sampled
    problems within tolerance of the oracle, and the answer of a problem does not depend on the batch it is in
    (bit-identical to a 4096-problem run of the same problems)."""
    N, B = 20, 262144
    hb = make_batch(B, N)
    # the prediction length is set by hand: left to the library it follows the size of the launch (4 steps for a grid of several
    # residencies, 6 for a launch on its own), which changes the number of sweeps a problem is reported with, not its solution
    eng = nmpc_mod.BatchedNmpc(B, N, warm_start_steps=4)
    eng.load(hb)
    eng.rti(1)
    out = eng.fetch()
    assert (out["status"] == 0).all() and np.isfinite(out["x"]).all() and np.isfinite(out["u"]).all()
    orc = Oracle(N)
    for b in (0, 1, B // 3, B // 2, B - 2, B - 1):
        st, ref, _ = oracle_tick(orc, problem(hb, b))
        assert st == 0 and relerr(out["u"][b].reshape(-1), ref["u"]) < 1e-4 and relerr(out["x"][b].reshape(-1), ref["x"]) < 1e-4
    lo = B - 4096
    # same lane mapping as the big launch: the library picks the mapping from the batch size, and different
    # mappings agree to rounding only
    small = nmpc_mod.BatchedNmpc(4096, N, lanes_per_problem=eng.launch_info()["lanes_per_problem"], warm_start_steps=4)
    small.load({k: v[lo:] for k, v in hb.items()})
    small.rti(1)
    so = small.fetch()
    for k in ("x", "u", "dual", "kkt", "obj", "n_iter"):
        assert np.array_equal(so[k], out[k][lo:]), k


@pytest.mark.gpu
def test_wide_problems_three_ticks_against_oracle(nmpc_mod):
    """Stress distribution (scenarios.make_wide_batch): 6000 problems x 3 consecutive ticks (ticks 1, 2 start from
    stale duals).  Every problem must be solved -- three of them make the primal-dual working-set iteration cycle
    and are finished by the active-set safeguard -- and agree with the oracle.
    Tolerance: 1e-4 (BASELINE.json) wherever the reference's own float32 answer is that close to the float64
    solution of its QP; where it is not (cond(H) ~ 2e3: the reference's homotopy is up to 3.5e-4 off), the kernel
    must be CLOSER to the float64 solution than the reference is -- proven here per problem, not assumed."""
    from alore_legged_manipulator_amd.scenarios import make_wide_batch
    N, B = 20, 6000
    batch = make_wide_batch(B, N, 99)
    eng = nmpc_mod.BatchedNmpc(B, N)
    eng.load(batch)
    orc = Oracle(N)
    prev = None
    long_runs = 0
    err_k, err_r, loose = [], [], 0
    for k in range(3):
        u_in = eng.fetch(names=("u",))["u"].copy()
        eng.rti(1)
        out = eng.fetch()
        assert (out["status"] == 0).all(), (k, np.nonzero(out["status"])[0][:8], out["status"][out["status"] != 0][:8])
        check = set(range(0, B, 13)) | set(np.nonzero(out["n_iter"] >= 10)[0].tolist()) | {95, 2734, 3996}
        for b in sorted(check):
            p = dict(problem(batch, b))
            if prev is not None:
                p["x"] = prev["x"][b].reshape(-1); p["u"] = prev["u"][b].reshape(-1); p["dual"] = prev["dual"][b].reshape(-1)
            orc.reset(); orc.initialize_solver(); orc.load(p); orc.preparation_step()
            assert orc.feedback_step() == 0
            eu, ex = relerr(out["u"][b].reshape(-1), orc.v["u"]), relerr(out["x"][b].reshape(-1), orc.v["x"])
            assert eu < 5e-4 and ex < 5e-4, (k, b, eu, ex)   # hard cap; the real criterion follows
            # the float64 solution of the reference's own condensed QP (its float32 H, g, bounds)
            n = 2 * N
            du_true = exact_box_qp(orc.v["H"].reshape(n, n), orc.v["g"], orc.v["lb"], orc.v["ub"])
            scale = max(1.0, float(np.max(np.abs(orc.v["u"]))))
            ek = float(np.max(np.abs((out["u"][b].reshape(-1).astype(np.float64) - u_in[b].reshape(-1)) - du_true))) / scale
            er = float(np.max(np.abs(orc.v["dx"].astype(np.float64) - du_true))) / scale
            err_k.append(ek); err_r.append(er)
            if max(eu, ex) >= 1e-4:          # beyond BASELINE's tolerance against the reference: then the reference is the
                loose += 1                   # one that is off, and the kernel is the closer of the two to the true minimiser
                assert ek < er and ek < 1e-4, (k, b, eu, ex, ek, er)
        long_runs += int((out["n_iter"] > 16).sum())
        prev = out
    err_k, err_r = np.array(err_k), np.array(err_r)
    print(f"wide: {len(err_k)} QPs; vs float64 truth: kernel max {err_k.max():.2e} median {np.median(err_k):.2e}; "
          f"reference max {err_r.max():.2e} median {np.median(err_r):.2e}; beyond 1e-4 of the reference: {loose}")
    assert err_k.max() < 1e-4 and err_k.max() <= err_r.max()
    assert loose <= len(err_k) // 200, (loose, len(err_k))   # measured: 2 of 1398 (0.14 %); bound 0.5 %
    assert long_runs >= 3        # the safeguard ran


@pytest.mark.gpu
@pytest.mark.parametrize("B,N,L", [(1, 20, 0), (5, 20, 0), (33, 20, 8), (5, 7, 0), (3, 50, 0), (7, 50, 16), (2, 1, 0), (2051, 20, 0), (4101, 7, 0)])
def test_no_out_of_bounds_writes(nmpc_mod, B, N, L):
    """Ragged batches (padding lane groups shadow the last problem) must not write outside their members: the
    members of slot 1 sit between those of slots 0 and 2 in memory, which hold a canary pattern."""
    import torch
    eng = nmpc_mod.BatchedNmpc(B, N, lanes_per_problem=L, slots=3)
    batch = make_batch(B, N, seed=5, fast_tail=0.5)
    eng.load(batch, slot=1)
    for k, t in eng.ts.items():
        for s in (0, 2):
            t[s].fill_(777 if t.dtype == torch.int32 else 12345.0)
    eng.rti(1, slot=1)
    eng.rti(2, slot=1)
    if N >= 7:
        eng.refs_init(max_pieces=4, max_checkpoints=32)
        eng.refs_sample(0.1, np.zeros((B, 3)), np.tile([0.1, -0.3, 0.3], (B, 1)), slot=1)   # empty store: must leave slot 1 alone too
    torch.cuda.synchronize()
    for k, t in eng.ts.items():
        want = 777 if t.dtype == torch.int32 else 12345.0
        for s in (0, 2):
            assert bool((t[s] == want).all()), (k, s)
    out = eng.fetch(slot=1)
    assert (out["status"] == 0).all() and np.isfinite(out["x"]).all()


@pytest.mark.gpu
def test_results_do_not_depend_on_wavefront_mates(nmpc_mod):
    """A problem's answer must be the same bits whatever problems share its wavefront (prediction length and sweep
    restarts are decided per problem / parity-preserving): permute a batch that needs many working-set sweeps."""
    from alore_legged_manipulator_amd.scenarios import make_wide_batch
    B, N = 4096, 20
    batch = make_wide_batch(B, N, 3)
    perm = np.random.default_rng(0).permutation(B)
    e1 = nmpc_mod.BatchedNmpc(B, N); e1.load(batch); e1.rti(1); o1 = e1.fetch()
    e2 = nmpc_mod.BatchedNmpc(B, N); e2.load({k: v[perm] for k, v in batch.items()}); e2.rti(1); o2 = e2.fetch()
    assert (o1["n_iter"] > 1).sum() > 100                      # the batch does exercise restarts
    for k in ("x", "u", "dual", "kkt", "obj", "status", "n_iter"):
        assert np.array_equal(o1[k][perm], o2[k]), k


@pytest.mark.gpu
def test_shared_members_give_the_replicated_results(nmpc_mod):
    """alore_nmpc_set_shared_members: W/WN, bounds and od read from ONE copy (row 0) must give exactly what B
    identical copies give; rows > 0 of the shared members are then never read (poisoned here)."""
    N, B = 20, 1024
    batch = make_batch(B, N, seed=12, fast_tail=0.2)
    batch["od"][:] = batch["od"][0]                      # one object class for the whole batch
    ref = nmpc_mod.BatchedNmpc(B, N); ref.load(batch); ref.rti(2); want = ref.fetch()
    eng = nmpc_mod.BatchedNmpc(B, N)
    poisoned = {k: v.copy() for k, v in batch.items()}
    for k in ("W", "WN", "lbValues", "ubValues", "od"):
        poisoned[k][1:] = np.nan
    eng.load(poisoned)
    eng.set_shared_members(W=True, bounds=True, od=True)
    eng.rti(2)
    got = eng.fetch()
    assert (got["status"] == 0).all()
    for k in ("x", "u", "dual", "kkt", "obj", "n_iter"):
        assert np.array_equal(got[k], want[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("N,B", [(20, 257), (50, 70), (20, 2100), (7, 33)])
def test_without_diagnostics_same_solution(nmpc_mod, N, B):
    """kkt / obj NULL in the batch: the diagnostics are skipped, everything else is bit-identical (all kernel
    variants: one / four wavefronts per workgroup, register-resident weights at N = 50)."""
    batch = make_batch(B, N, seed=41, fast_tail=0.3)
    a = nmpc_mod.BatchedNmpc(B, N); a.load(batch); a.rti(2); oa = a.fetch()
    b = nmpc_mod.BatchedNmpc(B, N, diagnostics=False); b.load(batch)
    b.t["kkt"].fill_(-7.0); b.t["obj"].fill_(-7.0)
    b.rti(2); ob = b.fetch()
    for k in ("x", "u", "dual", "status", "n_iter"):
        assert np.array_equal(oa[k], ob[k]), k
    assert (ob["kkt"] == -7.0).all() and (ob["obj"] == -7.0).all()      # untouched


@pytest.mark.gpu
def test_c_abi_result_exchange_over_rccl_single_rank():
    """alore_nmpc_comm_* (the C / C++ caller's multi-GPU path): with one rank the all-gather must reproduce the local
    results on the gathered buffers -- exercises the lazy RCCL load, communicator creation and the grouped call on a
    real device.  (More ranks need more GPUs; the Python twin of this path is covered by tests/test_shard_gloo.py.)"""
    import ctypes as C
    import torch
    from alore_legged_manipulator_amd import _lib
    from alore_legged_manipulator_amd.nmpc import BatchedNmpc
    from alore_legged_manipulator_amd.scenarios import make_batch
    lib = _lib.load()
    B, N = 64, 20
    eng = BatchedNmpc(B, N)
    eng.load(make_batch(B, N, seed=5))
    eng.rti(1)
    uid = C.create_string_buffer(128)
    assert lib.alore_nmpc_comm_unique_id(uid) == 0, lib.alore_nmpc_comm_last_error()
    comm = C.c_void_p()
    assert lib.alore_nmpc_comm_create(1, 0, uid, 0, C.byref(comm)) == 0, lib.alore_nmpc_comm_last_error()
    local, allb = _lib.Batch(), _lib.Batch()
    out = {"x": torch.zeros(B * 3 * (N + 1), dtype=torch.float32, device="cuda"), "u": torch.zeros(B * 2 * N, dtype=torch.float32, device="cuda"),
           "status": torch.full((B,), -7, dtype=torch.int32, device="cuda"), "kkt": torch.zeros(B, dtype=torch.float32, device="cuda")}
    for k in ("x", "u", "status", "kkt"):
        setattr(local, k, eng.t[k].data_ptr())
        setattr(allb, k, out[k].data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.alore_nmpc_comm_all_gather(comm, C.byref(local), B, N, C.byref(allb), stream) == 0, lib.alore_nmpc_comm_last_error()
    torch.cuda.synchronize()
    for k in ("x", "u", "status", "kkt"):
        assert torch.equal(out[k].reshape(-1), eng.t[k].reshape(-1)), k
    assert lib.alore_nmpc_comm_destroy(comm) == 0


BLOCK = 0x100  # ALORE_NMPC_BLOCK_LANES(L) = 0x100 | L (include/alore_nmpc.h)


@pytest.mark.gpu
@pytest.mark.parametrize("N,lanes", [(20, BLOCK | 4), (20, BLOCK | 8), (20, BLOCK | 16), (7, BLOCK | 4), (24, BLOCK | 8),
                                     (31, BLOCK | 16), (50, BLOCK | 16), (1, BLOCK | 4), (20, BLOCK | 32), (32, BLOCK | 32),
                                     (9, BLOCK | 32), (20, 32), (50, 64), (32, BLOCK | 16), (64, BLOCK | 16)])
def test_every_lane_mapping_against_the_oracle(nmpc_mod, N, lanes):
    """Both kernels, every instantiated lane mapping (stage-block L = 4 / 8 / 16 / 32 with 5 / 3 / 2 or 4 / 1 stages per lane, the
    wavefront mapping): one tick from the cold start and one warm tick (the dual seeds the working set) of seeded
    problems, a ragged batch, against the oracle: x, u <= 1e-4 relative (BASELINE.json), multipliers 1e-3."""
    B = 37
    batch = make_batch(B, N, seed=31, fast_tail=0.4)
    eng = nmpc_mod.BatchedNmpc(B, N, lanes_per_problem=lanes)
    eng.load(batch)
    eng.rti(1)
    a = eng.fetch()
    assert eng.launch_info()["lanes_per_problem"] == lanes
    eng.rti(1)
    b = eng.fetch()
    orc = Oracle(N)
    for p in range(0, B, 3):
        orc.reset(); orc.initialize_solver(); orc.load(problem(batch, p))
        for out in (a, b):
            orc.preparation_step()
            assert orc.feedback_step() == 0 and out["status"][p] == 0
            for k, tol in (("x", 1e-4), ("u", 1e-4), ("dual", 1e-3)):
                assert relerr(out[k][p].reshape(-1), orc.v[k]) < tol, (p, k)
            assert abs(out["obj"][p] - orc.get_objective()) <= 1e-4 * max(1.0, abs(orc.get_objective()))
            assert abs(out["kkt"][p] - orc.get_kkt()) <= 2e-3 * max(1.0, orc.get_kkt())


def check_against_oracle_and_float64(batch, out, u_in, picks, N, tag, max_loose_share=0.01):
    """The rule of test_wide_problems_three_ticks_against_oracle for one cold tick (distances are element-wise relative errors
    with an absolute floor of 1, `relerr`):
      * never more than 5e-4 from the oracle -- except where the ORACLE misses the float64 minimiser of its own condensed QP by more
        than 1e-4 (the reference's float32 homotopy does, on ill-conditioned problems): there 1e-3, and at most one such problem in 400;
      * wherever the kernel is more than 1e-4 (BASELINE.json) from the oracle it must be closer than the oracle to that float64
        minimiser and within 1e-4 of it ("loose" problems), and their share is bounded: <= max_loose_share of the picks.
    Measured in round 5 (MI355X): stress batch, every mapping: 2 of 457 picks loose (0.44 %; the picks are every 9th problem plus
    every one with >= 10 working-set iterations, i.e. the hard ones), one of them 6.1e-4 from the oracle (oracle 1.7e-4 from the
    float64 minimiser, kernel 5.6e-5); slots of the timed configuration: 0 of 180 loose, worst 6.1e-5; wide batch, 3 ticks: 2 of 1398."""
    orc = Oracle(N)
    err_k, err_r, loose, worst, oracle_off = [], [], 0, 0.0, 0
    for b in picks:
        orc.reset(); orc.initialize_solver(); orc.load(problem(batch, b)); orc.preparation_step()
        assert orc.feedback_step() == 0 and out["status"][b] == 0, (tag, b)
        eu, ex = relerr(out["u"][b].reshape(-1), orc.v["u"]), relerr(out["x"][b].reshape(-1), orc.v["x"])
        worst = max(worst, eu, ex)
        n = 2 * N
        du_true = exact_box_qp(orc.v["H"].reshape(n, n), orc.v["g"], orc.v["lb"], orc.v["ub"])
        scale = max(1.0, float(np.max(np.abs(orc.v["u"]))))
        ek = float(np.max(np.abs((out["u"][b].reshape(-1).astype(np.float64) - u_in[b].reshape(-1)) - du_true))) / scale
        er = float(np.max(np.abs(orc.v["dx"].astype(np.float64) - du_true))) / scale
        err_k.append(ek); err_r.append(er)
        if max(eu, ex) >= 5e-4:   # hard cap, lifted only where the oracle is demonstrably the one that is off
            oracle_off += 1
            assert er > 1e-4 and max(eu, ex) < 1e-3, (tag, b, eu, ex, ek, er)
        if max(eu, ex) >= 1e-4:
            loose += 1
            print(f"  {tag}: problem {b} beyond 1e-4 of the oracle: u {eu:.2e} x {ex:.2e}; from the float64 minimiser: kernel {ek:.2e}, oracle {er:.2e}")
            assert ek < er and ek < 1e-4, (tag, b, eu, ex, ek, er)
    err_k, err_r = np.array(err_k), np.array(err_r)
    print(f"{tag}: {len(err_k)} QPs; vs float64 truth: kernel max {err_k.max():.2e}, oracle max {err_r.max():.2e}; beyond 1e-4 of the oracle: {loose} "
          f"({100.0 * loose / len(err_k):.2f} %), beyond 5e-4 (oracle off): {oracle_off}, worst distance from the oracle {worst:.2e}")
    assert err_k.max() < 1e-4
    assert loose <= max(1, int(max_loose_share * len(err_k))), (tag, loose, len(err_k))
    assert oracle_off <= max(1, len(err_k) // 400), (tag, oracle_off, len(err_k))
    return loose, len(err_k), worst


@pytest.mark.gpu
@pytest.mark.parametrize("lanes", [BLOCK | 4, BLOCK | 8, BLOCK | 16, BLOCK | 32])
def test_stage_block_kernel_on_the_stress_distribution(nmpc_mod, lanes):
    """Every lane mapping of the stage-block kernel on the wide distribution (long working-set iterations, the primal
    active-set safeguard), pinned directly: >= 400 problems per mapping -- every 9th and every one that needed 10 or more
    working-set iterations -- against the ORACLE and the float64 minimiser of its condensed QP by the rule of
    test_wide_problems_three_ticks_against_oracle, and the answer of a problem is the same bits wherever it sits in the
    batch (permutation)."""
    from alore_legged_manipulator_amd.scenarios import make_wide_batch
    B, N = 4099, 20
    batch = make_wide_batch(B, N, 3)
    eng = nmpc_mod.BatchedNmpc(B, N, lanes_per_problem=lanes); eng.load(batch); eng.rti(1); got = eng.fetch()
    assert eng.launch_info()["lanes_per_problem"] == lanes
    assert (got["status"] == 0).all()
    assert (got["n_iter"] > 3).sum() > 50 and got["n_iter"].max() > 16     # restarts and the safeguard are exercised
    picks = sorted(set(range(0, B, 9)) | set(np.nonzero(got["n_iter"] >= 10)[0].tolist()))
    assert len(picks) >= 400
    check_against_oracle_and_float64(batch, got, batch["u"], picks, N, f"stress L={lanes & 0xff}")
    perm = np.random.default_rng(1).permutation(B)
    e2 = nmpc_mod.BatchedNmpc(B, N, lanes_per_problem=lanes); e2.load({k: v[perm] for k, v in batch.items()}); e2.rti(1)
    o2 = e2.fetch()
    for k in ("x", "u", "dual", "kkt", "obj", "status", "n_iter"):
        assert np.array_equal(got[k][perm], o2[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["groups", "streams"])
def test_what_the_bench_times_is_pinned(nmpc_mod, mode):
    """The configuration bench.py times -- automatic lane mapping, one iteration per launch (n_sqp = 1), slots of B = 4096 kept
    in flight by ONE alore_nmpc_rti_many call -- on slots with DIFFERENT problems: (a) the automatic choice is the (4, 5)
    mapping; (b) every slot is bit-equal to the same slot solved alone with the mapping forced to what launch_info()
    reports, eagerly and replayed from a hipGraph; (c) 600 problems over three slots, half of them from the stress
    distribution, against the oracle and float64."""
    import torch
    from alore_legged_manipulator_amd.scenarios import make_wide_batch
    B, N, slots = 4096, 20, 9
    batches = [make_batch(B, N, seed=100 + s, fast_tail=0.3) if s % 2 == 0 else make_wide_batch(B, N, 40 + s) for s in range(slots)]
    eng = nmpc_mod.BatchedNmpc(B, N, slots=slots)           # lanes_per_problem = 0: automatic
    eng.set_many_mode(mode)
    for s in range(slots):
        eng.load(batches[s], slot=s)
    eng.rti_range(0, slots)
    torch.cuda.synchronize()
    info = eng.launch_info()
    assert info["lanes_per_problem"] == (BLOCK | 4), info
    got = {k: eng.ts[k].clone() for k in ("x", "u", "dual", "status", "n_iter", "kkt", "obj")}
    assert (got["status"] == 0).all()
    # inside a grid that fills the chip the library runs the working-set prediction with 4 steps instead of 6 (the solution does
    # not depend on it, the iteration counts do): the slot-by-slot reference is given the same number
    ref = nmpc_mod.BatchedNmpc(B, N, slots=slots, lanes_per_problem=info["lanes_per_problem"], warm_start_steps=4 if mode == "groups" else -1)
    for s in range(slots):
        ref.load(batches[s], slot=s)
    for s in range(slots):
        ref.rti(1, slot=s)
    torch.cuda.synchronize()
    for k, v in got.items():
        assert torch.equal(ref.ts[k], v), k
    # replayed from a graph
    for s in range(slots):
        eng.load(batches[s], slot=s)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            eng.rti_range(0, slots)
    g.replay()
    torch.cuda.synchronize()
    for k, v in got.items():
        assert torch.equal(eng.ts[k], v), k
    for s in (0, 3, 8):
        out = {k: v[s].cpu().numpy() for k, v in got.items()}
        picks = sorted(set(range(0, B, 23)) | set(np.nonzero(out["n_iter"] >= 10)[0].tolist()[:40]))
        check_against_oracle_and_float64(batches[s], out, batches[s]["u"], picks, N, f"in flight ({mode}) slot {s}")


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["strided", "scattered"])
def test_groups_of_batches_ragged_and_more_than_one_grid(nmpc_mod, layout):
    """alore_nmpc_rti_many in groups mode, 53 slots of a ragged batch (B not a multiple of the problems per wavefront),
    automatic mapping.  strided: the slots of one arena in order -> ONE grid over all 53 (constant member strides).
    scattered: the same slots in a shuffled order -> descriptor tables of at most 24, three grids.  Either way the bits of
    slot-by-slot launches with the mapping pinned; canary slots around them stay intact."""
    import ctypes as C
    import torch
    from alore_legged_manipulator_amd._lib import Batch
    B, N, slots = 212, 20, 55      # 212 = 13.25 wavefronts of 16 problems; a multiple of 4 keeps every slot 16-byte aligned
    eng = nmpc_mod.BatchedNmpc(B, N, slots=slots)
    rng = np.random.default_rng(0)
    seeds = rng.integers(0, 1 << 30, slots)
    for s in range(slots):
        eng.load(make_batch(B, N, seed=int(seeds[s]), fast_tail=0.3), slot=s)
    for k, t in eng.ts.items():
        for s in (0, slots - 1):
            t[s].fill_(777 if t.dtype == torch.int32 else 12345.0)
    snap = {k: eng.ts[k].clone() for k in eng.ts}
    if layout == "strided":
        eng.rti_range(1, slots - 2)
        n_last = slots - 2
    else:
        order = rng.permutation(np.arange(1, slots - 1))
        arr = (Batch * (slots - 2))(*[eng._batches[int(i)] for i in order])
        eng._check(eng.lib.alore_nmpc_rti_many(eng.h, arr, slots - 2, B, 1, eng._stream()))
        n_last = 17                              # 53 = 18 + 18 + 17: launch_info describes the last grid
    torch.cuda.synchronize()
    info = eng.launch_info()
    # the mapping packs for the problems of ONE grid: all 53 x 212 (4 lanes per problem) or 18 x 212 (16 lanes)
    assert info["lanes_per_problem"] == ((BLOCK | 4) if layout == "strided" else (BLOCK | 16)), info
    per_wave = 64 // (info["lanes_per_problem"] & 0xff)
    assert info["grid"] == n_last * ((B + per_wave - 1) // per_wave), info
    for k, t in eng.ts.items():
        want = 777 if t.dtype == torch.int32 else 12345.0
        for s in (0, slots - 1):
            assert bool((t[s] == want).all()), (k, s)
    ref = nmpc_mod.BatchedNmpc(B, N, slots=slots, lanes_per_problem=info["lanes_per_problem"])
    for k in ref.ts:
        ref.ts[k].copy_(snap[k])
    for s in range(1, slots - 1):
        ref.rti(1, slot=s)
    torch.cuda.synchronize()
    assert (ref.ts["status"][1:slots - 1] == 0).all()
    for k in ("x", "u", "dual", "status", "n_iter", "kkt", "obj"):
        assert torch.equal(eng.ts[k][1:slots - 1], ref.ts[k][1:slots - 1]), k
    with pytest.raises(ValueError):
        eng.rti_range(slots - 1, 2)


@pytest.mark.gpu
def test_overlapping_slices_keep_the_order(nmpc_mod):
    """alore_nmpc_rti_many's independence check works on address ranges: two batches whose x arrays overlap without being
    equal pointers (a slice shifted by one problem) are not independent and run in order, one launch each."""
    import ctypes as C
    import torch
    from alore_legged_manipulator_amd._lib import Batch
    B, N = 64, 20
    eng = nmpc_mod.BatchedNmpc(B + 4, N, slots=2)
    batch = make_batch(B + 4, N, seed=3, fast_tail=0.3)
    eng.load(batch, slot=None)
    arr = (Batch * 2)(shifted_of(eng, 0, 0), shifted_of(eng, 0, 4))     # the second batch starts four problems into the first (16-byte aligned members)
    eng._check(eng.lib.alore_nmpc_rti_many(eng.h, arr, 2, B, 1, eng._stream()))
    torch.cuda.synchronize()
    ref = nmpc_mod.BatchedNmpc(B + 4, N, slots=2, lanes_per_problem=eng.launch_info()["lanes_per_problem"])
    ref.load(batch, slot=None)
    ref._check(ref.lib.alore_nmpc_rti(ref.h, C.byref(shifted_of(ref, 0, 0)), B, 1, ref._stream()))
    ref._check(ref.lib.alore_nmpc_rti(ref.h, C.byref(shifted_of(ref, 0, 4)), B, 1, ref._stream()))
    torch.cuda.synchronize()
    for k in ("x", "u", "dual", "status"):
        assert torch.equal(eng.ts[k][0], ref.ts[k][0]), k


def shifted_of(eng, slot, off):
    from alore_legged_manipulator_amd._lib import Batch
    b = Batch()
    for name, _ in Batch._fields_:
        setattr(b, name, eng.ts[name][slot][off:].data_ptr())
    return b


@pytest.mark.gpu
@pytest.mark.parametrize("ways,mode", [(1, "groups"), (2, "streams"), (3, "groups"), (3, "streams"), (8, "streams"), (32, "groups"), (32, "streams")])
def test_overlapped_launches_of_independent_slots_give_the_in_order_bits(nmpc_mod, ways, mode):
    """alore_nmpc_rti_many keeps `ways` independent batches in flight on streams forked from / joined into the caller's:
    every slot's results are the bits of the same slot solved alone, eagerly and replayed from a hipGraph; a batch listed
    twice (two successive iterations of the same problems) keeps the call in order."""
    import ctypes as C
    import torch
    B, N, slots = 517, 20, 7
    batch = make_batch(B, N, seed=5, fast_tail=0.3)
    ref = nmpc_mod.BatchedNmpc(B, N, slots=slots, lanes_per_problem=BLOCK | 16)  # pinned: the automatic mapping packs for the launches in flight
    ref.load(batch, slot=None)
    for s in range(slots):
        ref.rti(1, slot=s)
    torch.cuda.synchronize()
    want = {k: ref.ts[k].clone() for k in ("x", "u", "dual", "status", "n_iter", "kkt", "obj")}
    eng = nmpc_mod.BatchedNmpc(B, N, slots=slots, lanes_per_problem=BLOCK | 16)
    eng.set_launch_overlap(ways)
    eng.set_many_mode(mode)
    eng.load(batch, slot=None)
    eng.rti_range(0, slots)
    torch.cuda.synchronize()
    for k, v in want.items():
        assert torch.equal(eng.ts[k], v), k
    # the same from a captured graph on a side stream
    eng.load(batch, slot=None)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        eng.rti_range(0, 1)          # warm the launch path outside the capture
        side.synchronize()
        eng.load(batch, slot=None)
        side.synchronize()
        with torch.cuda.graph(g, stream=side):
            eng.rti_range(0, slots)
    g.replay()
    torch.cuda.synchronize()
    for k, v in want.items():
        assert torch.equal(eng.ts[k], v), k
    # one batch twice = two iterations in order
    two = nmpc_mod.BatchedNmpc(B, N, lanes_per_problem=BLOCK | 16)
    two.load(batch); two.rti(1); two.rti(1)
    torch.cuda.synchronize()
    rep = nmpc_mod.BatchedNmpc(B, N, lanes_per_problem=BLOCK | 16)
    rep.set_launch_overlap(ways)
    rep.set_many_mode(mode)
    rep.load(batch)
    from alore_legged_manipulator_amd._lib import Batch
    arr = (Batch * 2)(rep._batches[0], rep._batches[0])
    rep._check(rep.lib.alore_nmpc_rti_many(rep.h, arr, 2, B, 1, rep._stream()))
    torch.cuda.synchronize()
    for k in ("x", "u", "dual", "status"):
        assert torch.equal(rep.ts[k], two.ts[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("n_sqp", [2, 15])
@pytest.mark.parametrize("mode", ["groups", "streams"])
@pytest.mark.parametrize("lanes", [BLOCK | 4, BLOCK | 16])
def test_launches_in_flight_with_two_iterations_and_shared_members(nmpc_mod, lanes, mode, n_sqp):
    """alore_nmpc_rti_many on the builds that keep the iteration loop (n_sqp = 2; 15 = the converged solves that bench.py's
    converged_all_gather.in_flight keeps in one grid), with W / bounds / od read from one shared copy, five slots in flight: the
    bits of slot-by-slot launches."""
    import torch
    B, N, slots = 300, 20, 5
    batch = make_batch(B, N, seed=8, fast_tail=0.3)
    batch["od"][:] = batch["od"][0]
    ref = nmpc_mod.BatchedNmpc(B, N, slots=slots, lanes_per_problem=lanes)
    ref.load(batch, slot=None)
    ref.set_shared_members(W=True, bounds=True, od=True)
    for s in range(slots):
        ref.rti(n_sqp, slot=s)
    torch.cuda.synchronize()
    eng = nmpc_mod.BatchedNmpc(B, N, slots=slots, lanes_per_problem=lanes)
    eng.load(batch, slot=None)
    eng.set_shared_members(W=True, bounds=True, od=True)
    eng.set_launch_overlap(5)
    eng.set_many_mode(mode)
    eng.rti_range(0, slots, n_sqp=n_sqp)
    torch.cuda.synchronize()
    assert (eng.ts["status"] == 0).all()
    for k in ("x", "u", "dual", "status", "n_iter", "kkt", "obj"):
        assert torch.equal(eng.ts[k], ref.ts[k]), k


@pytest.mark.gpu
def test_input_column_is_the_column_of_the_downloaded_inputs(nmpc_mod):
    """alore_nmpc_input_column: the 12 bytes per problem a control tick publishes (inputs of one node + status), packed on the
    device into pinned memory, against the full download."""
    B, N = 777, 20
    eng = nmpc_mod.BatchedNmpc(B, N)
    eng.load(make_batch(B, N, seed=2, fast_tail=0.3))
    eng.rti(1)
    full = eng.fetch()
    for node in (0, 1, N - 1):
        cmd, st = eng.input_column(node)
        assert np.array_equal(cmd, full["u"][:, node, :]) and np.array_equal(st, full["status"])
    with pytest.raises(Exception):
        eng.input_column(N)


@pytest.mark.gpu
def test_diagonal_weight_path_agrees_with_the_general_path(nmpc_mod):
    """The N = 20 build of the (4, 5) mapping runs a wavefront whose W, WN are diagonal (the reference's controller sets
    W = diag(Q, R)) through an instantiation with literal zeros off the diagonal.  A denormal-sized off-diagonal entry in ONE
    problem sends that problem's wavefront (16 problems) down the general path: the results of all of them must agree with
    the diagonal path to float32 rounding, and the other wavefronts must not notice (bit-equal)."""
    B, N = 512, 20
    batch = make_batch(B, N, seed=12, fast_tail=0.3)
    eng = nmpc_mod.BatchedNmpc(B, N, lanes_per_problem=BLOCK | 4)
    eng.load(batch); eng.rti(1); a = eng.fetch()
    tweaked = {k: v.copy() for k, v in batch.items()}
    tweaked["W"][37, 3, 0, 1] = 1e-38; tweaked["W"][37, 3, 1, 0] = 1e-38
    e2 = nmpc_mod.BatchedNmpc(B, N, lanes_per_problem=BLOCK | 4)
    e2.load(tweaked); e2.rti(1); b = e2.fetch()
    assert (a["status"] == 0).all() and (b["status"] == 0).all()
    wave = np.arange(32, 48)                      # the wavefront of problem 37
    rest = np.setdiff1d(np.arange(B), wave)
    for k in ("x", "u", "dual", "kkt", "obj"):
        assert np.array_equal(a[k][rest], b[k][rest]), k
        d = np.max(np.abs(a[k][wave].astype(np.float64) - b[k][wave])) / max(1.0, float(np.max(np.abs(a[k][wave]))))
        assert d < 2e-6, (k, d)
    assert np.array_equal(a["n_iter"], b["n_iter"])


@pytest.mark.gpu
@pytest.mark.parametrize("N,lanes,n_sqp", [(20, 0, 1), (20, BLOCK | 4, 1), (20, BLOCK | 16, 2), (31, BLOCK | 16, 1), (7, BLOCK | 8, 1),
                                           (20, 32, 1), (50, 64, 2), (20, 16, 1)])
def test_masked_problems_are_left_exactly_as_they_are(nmpc_mod, N, lanes, n_sqp):
    """alore_nmpc_set_problem_mask: the problems with mask 0 (idle robots of a fleet) keep every member bit for bit -- x, u, dual,
    status, n_iter, kkt, obj -- even when their references are NaN, and the others return the bits of a launch without a mask."""
    B = 131
    batch = make_batch(B, N, seed=21, fast_tail=0.4)
    ref = nmpc_mod.BatchedNmpc(B, N, lanes_per_problem=lanes)
    ref.load(batch); ref.rti(n_sqp); want = ref.fetch()
    mask = np.ones(B, np.uint8); mask[[0, 5, 17, 64, 65, 130]] = 0
    broken = {k: v.copy() for k, v in batch.items()}
    broken["y"][5] = np.nan; broken["x0"][64] = np.inf            # stale / non-finite references of idle robots
    eng = nmpc_mod.BatchedNmpc(B, N, lanes_per_problem=lanes)
    eng.load(broken)
    for k in ("status", "n_iter"):
        eng.ts[k][0].fill_(-7)
    for k in ("kkt", "obj"):
        eng.ts[k][0].fill_(123.5)
    before = eng.fetch()
    eng.set_problem_mask(mask)
    eng.rti(n_sqp)
    got = eng.fetch()
    on, off = mask == 1, mask == 0
    for k in ("x", "u", "dual", "status", "n_iter", "kkt", "obj"):
        assert np.array_equal(got[k][off], before[k][off], equal_nan=True), k
        assert np.array_equal(got[k][on], want[k][on]), k
    eng.set_problem_mask(None)
    eng.load(batch); eng.rti(n_sqp)
    again = eng.fetch()
    for k in ("x", "u", "dual", "status"):
        assert np.array_equal(again[k], want[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("mode,overlap", [("groups", 16), ("streams", 4), ("groups", 1)])
def test_the_problem_mask_applies_to_every_batch_of_rti_many(nmpc_mod, mode, overlap):
    """alore_nmpc_rti_many (rti_range) with a problem mask, in every mode -- one grid, forked streams, in order: problem b of EVERY
    batch of the call sits out when mask[b] = 0 (bit-untouched members, NaN references harmless), the others return the bits
    of a call without a mask."""
    B, N, slots = 300, 20, 5
    rng = np.random.default_rng(5)
    mask = (rng.uniform(size=B) > 0.2).astype(np.uint8); mask[[0, 16, 299]] = 0
    engs = []
    for masked in (False, True):
        eng = nmpc_mod.BatchedNmpc(B, N, slots=slots)
        eng.set_launch_overlap(overlap); eng.set_many_mode(mode)
        for s in range(slots):
            b = make_batch(B, N, seed=40 + s, fast_tail=0.3)
            if masked:
                b["y"][16] = np.nan; b["yN"][299] = np.inf
            eng.load(b, slot=s)
        for k in ("status", "n_iter"):
            eng.ts[k].fill_(-3)
        for k in ("kkt", "obj"):
            eng.ts[k].fill_(-1.25)
        engs.append(eng)
    ref, eng = engs
    before = {k: eng.ts[k].cpu().numpy().copy() for k in ("x", "u", "dual", "status", "n_iter", "kkt", "obj")}
    ref.rti_range(0, slots)
    eng.set_problem_mask(mask)
    eng.rti_range(0, slots)
    on, off = mask == 1, mask == 0
    for k in before:
        want, got = ref.ts[k].cpu().numpy(), eng.ts[k].cpu().numpy()
        assert np.array_equal(got[:, off], before[k][:, off], equal_nan=True), (mode, k)
        assert np.array_equal(got[:, on], want[:, on]), (mode, k)
    assert (ref.ts["status"].cpu().numpy() == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("N", [20, 50, 7])
def test_condensed_qp_matches_the_oracle_and_its_dense_solve_matches_the_stage_wise_one(nmpc_mod, N):
    """alore_nmpc_condense: H, g, lb, ub of the condensed QP (what acado_condensePrep / condenseFdb leave in acadoWorkspace)
    against the oracle's (bit-exact with the reference at N = 50) on a non-trivial iterate; alore_nmpc_dense_qp (the role of
    acado_solve) on those arrays against the float64 minimiser and against the step of the stage-wise real-time iteration."""
    import torch
    B = 24
    batch = make_batch(B, N, seed=4, fast_tail=0.5)
    eng = nmpc_mod.BatchedNmpc(B, N)
    eng.load(batch)
    eng.rti(1)                                   # a non-trivial iterate and dual
    mid = eng.fetch()
    qp = eng.condense()
    Hd, gd, lbd, ubd = (qp[k].cpu().numpy() for k in ("H", "g", "lb", "ub"))
    y0 = eng.ts["dual"][0].clone()
    x, y, st, ni = eng.dense_qp(qp["H"], qp["g"], qp["lb"], qp["ub"], y0=y0)
    eng.rti(1)
    out = eng.fetch()
    n = 2 * N
    x, y, st = x.cpu().numpy().reshape(B, n), y.cpu().numpy().reshape(B, n), st.cpu().numpy()
    assert (st == 0).all() and (out["status"] == 0).all()
    orc = Oracle(N)
    for b in range(0, B, 3):
        p = dict(problem(batch, b))
        p["x"] = mid["x"][b].reshape(-1); p["u"] = mid["u"][b].reshape(-1); p["dual"] = mid["dual"][b].reshape(-1)
        orc.reset(); orc.initialize_solver(); orc.load(p); orc.preparation_step()
        assert orc.feedback_step() == 0
        H, g = orc.v["H"].reshape(n, n), orc.v["g"]
        assert np.max(np.abs(Hd[b] - H)) < 1e-5 * np.max(np.abs(H)), (b, "H")
        assert np.max(np.abs(gd[b] - g)) < 1e-5 * max(1.0, np.max(np.abs(g))), (b, "g")
        assert np.array_equal(lbd[b], orc.v["lb"]) and np.array_equal(ubd[b], orc.v["ub"])
        truth = exact_box_qp(H, g, orc.v["lb"], orc.v["ub"])
        scale = max(1.0, float(np.max(np.abs(orc.v["u"]))))
        assert np.max(np.abs(x[b] - truth)) / scale < 1e-4, (b, np.max(np.abs(x[b] - truth)))
        # the stage-wise iteration took the same step (clipped into the box, like the expansion does)
        du = out["u"][b].reshape(-1) - mid["u"][b].reshape(-1)
        assert np.max(np.abs(du - np.clip(x[b], lbd[b], ubd[b]))) / scale < 1e-4, b
        act = np.abs(y[b]) > 1e-6
        assert np.all(np.where(y[b][act] > 0, np.abs(x[b][act] - lbd[b][act]), np.abs(x[b][act] - ubd[b][act])) < 1e-5)


@pytest.mark.gpu
def test_dense_box_qp_edge_cases(nmpc_mod):
    """alore_nmpc_dense_qp (the role of acado_solve / QProblemB, QPO/SRC/QProblemB.cpp:315-506) at the edges: orders 1 and 128,
    most variables at a bound, a warm start from the true dual (one factorisation), a wrong warm start (recovers), and the
    reference's failure codes: inconsistent bounds 33 (RET_INIT_FAILED_INFEASIBILITY), a Hessian that is not positive
    definite 31 (RET_INIT_FAILED_CHOLESKY), an iteration cap that is too small 58 (RET_MAX_NWSR_REACHED).  Problems of a launch do not
    see each other: the failing ones sit between healthy ones."""
    import torch
    rng = np.random.default_rng(5)
    dev = torch.device("cuda:0")

    def spd(n, cond=50.0):
        q, _ = np.linalg.qr(rng.normal(size=(n, n)))
        return (q * np.geomspace(1.0, cond, n)) @ q.T

    eng = nmpc_mod.BatchedNmpc(4, 20)
    for n in (1, 2, 40, 128):
        B = 6
        H = np.stack([spd(n) for _ in range(B)]).astype(np.float32)
        g = rng.normal(size=(B, n)).astype(np.float32) * 3.0
        lb = -np.abs(rng.normal(size=(B, n))).astype(np.float32) * 0.3
        ub = np.abs(rng.normal(size=(B, n))).astype(np.float32) * 0.3
        g[1] *= 100.0                      # most variables of problem 1 end at a bound
        lb[3, 0], ub[3, 0] = 1.0, -1.0     # problem 3: inconsistent bounds
        H[4] = -H[4]                       # problem 4: not positive definite
        t = [torch.from_numpy(a).to(dev) for a in (H, g, lb, ub)]
        x, y, st, ni = eng.dense_qp(*t)
        x, y, st, ni = x.cpu().numpy(), y.cpu().numpy(), st.cpu().numpy(), ni.cpu().numpy()
        assert st[3] == 33 and st[4] == 31, (n, st)
        for b in (0, 1, 2, 5):
            assert st[b] == 0, (n, b, st)
            truth = exact_box_qp(H[b], g[b], lb[b], ub[b])
            assert np.max(np.abs(x[b] - truth)) < 1e-4 * max(1.0, np.max(np.abs(truth))), (n, b)
            assert np.all(x[b] >= lb[b] - 1e-6) and np.all(x[b] <= ub[b] + 1e-6)
            grad = H[b].astype(np.float64) @ x[b] + g[b]
            free = y[b] == 0.0
            assert np.max(np.abs(grad[free]), initial=0.0) < 2e-3 * max(1.0, np.max(np.abs(g[b]))), (n, b)   # stationarity on the free set
            assert np.all(y[b][~free] * (x[b][~free] - 0.5 * (lb[b][~free] + ub[b][~free])) <= 0.0)          # > 0 at lower, < 0 at upper
        assert np.mean((x[1] == lb[1]) | (x[1] == ub[1])) > 0.5            # problem 1 sits mostly on its bounds
        # warm start from the dual just returned: the working set reproduces itself in one factorisation
        x2, y2, st2, ni2 = eng.dense_qp(*t, y0=torch.from_numpy(y).to(dev))
        ok = [0, 1, 2, 5]
        assert (st2.cpu().numpy()[ok] == 0).all() and (ni2.cpu().numpy()[ok] == 1).all(), (n, ni2)
        assert np.max(np.abs(x2.cpu().numpy()[ok] - x[ok])) < 1e-5
        # warm start from the opposite working set: still the same minimiser
        x3, _, st3, _ = eng.dense_qp(*t, y0=torch.from_numpy(-y - 1.0).to(dev))
        assert (st3.cpu().numpy()[ok] == 0).all()
        assert np.max(np.abs(x3.cpu().numpy()[ok] - x[ok])) < 1e-4 * max(1.0, np.max(np.abs(x[ok])))
    # iteration cap: a cold start with active bounds needs more than one factorisation
    capped = nmpc_mod.BatchedNmpc(4, 20, max_as_iter=1)
    n = 40
    H = np.stack([spd(n) for _ in range(3)]).astype(np.float32)
    g = (rng.normal(size=(3, n)) * 5.0).astype(np.float32)
    lb = np.full((3, n), -0.1, np.float32); ub = np.full((3, n), 0.1, np.float32)
    g[1] = 0.0                             # problem 1: the unconstrained minimiser 0 is inside the box, one factorisation
    _, _, st, ni = capped.dense_qp(*[torch.from_numpy(a).to(dev) for a in (H, g, lb, ub)])
    st = st.cpu().numpy()
    assert st[0] == 58 and st[2] == 58 and st[1] == 0, st


@pytest.mark.gpu
def test_prepared_descriptor_sets_and_overlapping_batches(nmpc_mod):
    """alore_nmpc_rti_many_prepare: an independent set passes and every contiguous run of it then goes straight to the launch (the
    cache compares the descriptors themselves); a set with a batch listed twice is refused; rti_range on overlapping batches still
    runs them in order (two iterations of the same slot = rti(2))."""
    import ctypes as C
    from alore_legged_manipulator_amd._lib import Batch
    B, N, slots = 260, 20, 6
    eng = nmpc_mod.BatchedNmpc(B, N, slots=slots)
    for s in range(slots):
        eng.load(make_batch(B, N, seed=70 + s, fast_tail=0.3), slot=s)
    eng.prepare_range(0, slots)
    eng.prepare_range(2, 3)                         # a run of the remembered set
    eng.rti_range(2, 3)
    ref = nmpc_mod.BatchedNmpc(B, N, slots=slots, lanes_per_problem=eng.launch_info()["lanes_per_problem"])
    for s in range(slots):
        ref.load(make_batch(B, N, seed=70 + s, fast_tail=0.3), slot=s)
    for s in (2, 3, 4):
        ref.rti(1, slot=s)
    for k in ("x", "u", "dual", "status", "n_iter"):
        assert np.array_equal(eng.ts[k][2:5].cpu().numpy(), ref.ts[k][2:5].cpu().numpy()), k
    twice = (Batch * 2)(eng._batches[0], eng._batches[0])
    rc = eng.lib.alore_nmpc_rti_many_prepare(eng.h, twice, 2, B)
    assert rc != 0 and b"overlap" in eng.lib.alore_nmpc_last_error(eng.h)
    # the same slot twice through rti_many: in order, i.e. two real-time iterations
    one = nmpc_mod.BatchedNmpc(B, N, lanes_per_problem=32); one.load(make_batch(B, N, seed=70, fast_tail=0.3)); one.rti(1); one.rti(1)
    two = nmpc_mod.BatchedNmpc(B, N, lanes_per_problem=32); two.load(make_batch(B, N, seed=70, fast_tail=0.3))
    pair = (Batch * 2)(two._batches[0], two._batches[0])
    assert two.lib.alore_nmpc_rti_many(two.h, pair, 2, B, 1, two._stream()) == 0
    for k in ("x", "u", "dual"):
        assert np.array_equal(one.fetch()[k], two.fetch()[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("env", ["ALORE_NMPC_PERSIST=1", "ALORE_NMPC_TRACE=%TRACE%",
                                 "ALORE_NMPC_XCD_SPEEDS=1.3,0.7,1.0,1.15,0.85,1.0,0.5,1.5 ALORE_NMPC_XCD_MIN_RES=2",   # shares apply from 12 residencies on by default
                                 "ALORE_NMPC_XCD_SHARES=0 ALORE_NMPC_XCD_MIN_RES=2"])
def test_the_persistent_grid_and_the_traced_twin_return_the_bits_of_the_plain_grid(nmpc_mod, env, tmp_path):
    """The opt-in persistent grid (blocks of problems by ticket), the instrumented twin of the grid build (per-SIMD trace), strongly
    uneven XCD shares (preset speeds: every XCD works on a different number of blocks, surplus workgroups leave at once) and equal
    shares run the same arithmetic as the default grid: a 24-batch rti_range in a child process with the switch on returns the bits of
    the default; the trace file parses (tools/trace_timeline.py) and accounts for every workgroup."""
    import subprocess
    import sys
    B, N, slots = 4096, 20, 24
    code = f"""
import sys, numpy as np
sys.path.insert(0, {ROOT!r})
from alore_legged_manipulator_amd.nmpc import BatchedNmpc
from alore_legged_manipulator_amd.scenarios import make_batch, make_wide_batch
eng = BatchedNmpc({B}, {N}, slots={slots})
for s in range({slots}):
    eng.load(make_batch({B}, {N}, seed=300 + s, fast_tail=0.3) if s % 3 else make_wide_batch({B}, {N}, 300 + s), slot=s)
eng.rti_range(0, {slots})
out = {{k: eng.ts[k].cpu().numpy() for k in ("x", "u", "dual", "status", "n_iter", "kkt", "obj")}}
np.savez(sys.argv[1], **out)
"""
    trace = str(tmp_path / "trace")
    switched = dict(kv.split("=") for kv in env.replace("%TRACE%", trace).split())
    res = {}
    for tag, extra in (("plain", {"ALORE_NMPC_XCD_MIN_RES": "2"}), ("switched", switched)):
        path = str(tmp_path / f"{tag}.npz")
        r = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[tag] = np.load(path)
    assert (res["plain"]["status"] == 0).all()
    for name in res["plain"].files:
        assert np.array_equal(res["plain"][name], res["switched"][name]), (env, name)
    if "TRACE" in env:
        files = sorted(tmp_path.glob("trace.*"))
        assert files
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "trace_timeline.py"), str(files[-1])], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and f"{slots * B // 16} workgroups" in r.stdout and "SIMDs used: 1024" in r.stdout, r.stdout[:500] + r.stderr[-500:]


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["bench+stress", "ragged+mask", "scattered", "tail of one workgroup", "graph"])
def test_two_phase_grids_return_the_bits_of_the_one_pass_grid(nmpc_mod, case):
    """alore_nmpc_set_two_phase(h, 1): a grid whose batches are solved in two passes (first pass without prediction, problems whose working
    set moves queued for tail workgroups later in the SAME grid) returns x, u, dual, status, kkt, obj of the one-pass grid -- every
    problem is solved by exactly one pass from its unmodified inputs, with the same arithmetic -- bit for bit on these cold starts, and
    n_iter never larger; on a WARM tick (the second tick of the same slots: generic float32 iterates, the dual names the working set) the
    same status and x, u, dual, kkt, obj within 2e-6 on the bench distribution (the kernel carries its own copies of the body, which the
    compiler contracts independently: measured 7e-7) and within BASELINE's 1e-4 on the stress batches, whose conditioning amplifies
    the last bit (measured 1.4e-5).  Cases: cold Monte-Carlo
    batches mixed with stress batches (general weights: workgroups that take one pass inside the first pass's range; 40 % queued);
    B not a multiple of 16 with masked-out problems; batches as a descriptor table (scattered order); a tail of ONE workgroup per batch
    (every tail workgroup takes turn after turn: the overflow path); the call replayed from a hipGraph."""
    import torch
    from alore_legged_manipulator_amd.scenarios import make_wide_batch
    N = 20
    B, slots = (4096, 12) if case != "ragged+mask" else (4092, 12)   # a multiple of 4 keeps every slot 16-byte aligned
    batches = [make_batch(B, N, seed=500 + s, fast_tail=0.3) if s % 3 else make_wide_batch(B, N, 500 + s) for s in range(slots)]
    order = list(range(slots))
    if case == "scattered":
        order = [7, 2, 11, 0, 5, 9, 1, 10, 3, 8, 6, 4]

    def run(two_phase, env=None):
        eng = nmpc_mod.BatchedNmpc(B, N, slots=slots)
        eng.set_two_phase(two_phase)
        for s in range(slots):
            eng.load(batches[s], slot=s)
        mask = None
        if case == "ragged+mask":
            mask = np.ones(B, dtype=np.uint8)
            mask[::7] = 0
            eng.set_problem_mask(mask)
        torch.cuda.synchronize()
        if case == "scattered":
            import ctypes as C
            from alore_legged_manipulator_amd._lib import Batch
            arr = (Batch * slots)(*[eng._batches[s] for s in order])
            eng._check(eng.lib.alore_nmpc_rti_many(eng.h, arr, slots, B, 1, eng._stream()))
        elif case == "graph":
            side = torch.cuda.Stream()
            eng.rti_range(0, slots)   # eager once: the queue region exists before the capture
            torch.cuda.synchronize()
            for s in range(slots):
                eng.load(batches[s], slot=s)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(g, stream=side):
                    eng.rti_range(0, slots)
            g.replay()
        else:
            eng.rti_range(0, slots)
        torch.cuda.synchronize()
        return {k: eng.ts[k].clone() for k in ("x", "u", "dual", "status", "n_iter", "kkt", "obj")}, eng.two_phase_info(), eng

    if case == "tail of one workgroup":
        os.environ["ALORE_NMPC_TP_TAIL"] = "1"   # read once per process by the library: this case runs in a child below
    try:
        if case == "tail of one workgroup":
            import subprocess
            import sys
            code = f"""
import sys, os, numpy as np, torch
sys.path.insert(0, {ROOT!r})
from alore_legged_manipulator_amd.nmpc import BatchedNmpc
from alore_legged_manipulator_amd.scenarios import make_batch, make_wide_batch
B, N, slots = {B}, {N}, {slots}
res = {{}}
for tp in (0, 1):
    eng = BatchedNmpc(B, N, slots=slots)
    eng.set_two_phase(tp)
    for s in range(slots):
        eng.load(make_batch(B, N, seed=500 + s, fast_tail=0.3) if s % 3 else make_wide_batch(B, N, 500 + s), slot=s)
    eng.rti_range(0, slots)
    torch.cuda.synchronize()
    res[tp] = {{k: eng.ts[k].cpu().numpy() for k in ("x", "u", "dual", "status", "n_iter", "kkt", "obj")}}
    info = eng.two_phase_info()
assert info["last_grid_two_phase"] == 1 and info["tail_workgroups_per_batch"] == 1, info
for k in ("x", "u", "dual", "status", "kkt", "obj"):
    assert np.array_equal(res[0][k], res[1][k]), k
assert (res[1]["n_iter"] <= res[0]["n_iter"]).all()
print("ok")
"""
            r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ), capture_output=True, text=True, timeout=600)
            assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
            return
    finally:
        os.environ.pop("ALORE_NMPC_TP_TAIL", None)
    one, info1, _ = run(0)
    two, info2, eng2 = run(1)
    assert info1["last_grid_two_phase"] == 0 and info2["last_grid_two_phase"] == 1 and info2["two_phase_batches"] >= 6, (info1, info2)
    assert 0.05 < info2["tail_share"] < 0.6 or case == "graph", info2
    for k in ("x", "u", "dual", "status", "kkt", "obj"):
        assert torch.equal(one[k], two[k]), (case, k)
    assert (two["n_iter"] <= one["n_iter"]).all()
    if case != "ragged+mask":
        assert (two["status"] == 0).all()
    # a warm tick on top (not for the scattered order: rti_range walks the slots in order)
    if case == "bench+stress":
        _, _, eng1 = run(0)
        eng1.rti_range(0, slots)
        eng2.rti_range(0, slots)
        torch.cuda.synchronize()
        assert torch.equal(eng1.ts["status"], eng2.ts["status"])
        for k in ("x", "u", "dual", "kkt", "obj"):
            d = (eng1.ts[k].double() - eng2.ts[k].double()).abs()
            rel = (d / torch.clamp(eng1.ts[k].double().abs(), min=1.0)).reshape(slots, -1).amax(dim=1)
            for sl in range(slots):   # slots of the bench distribution: rounding; stress slots (cond(H) ~ 2e3 amplifies it): BASELINE's tolerance
                tol = 2e-6 if sl % 3 else (1e-4 if k in ("x", "u") else 1e-3)
                assert float(rel[sl]) < tol, (case, "warm tick", k, sl, float(rel[sl]))
    # the queue is back in order: a second two-phase grid on the same handle gives the same bits
    for s in range(slots):
        eng2.load(batches[s], slot=s)
    torch.cuda.synchronize()
    eng2.rti_range(0, slots)
    torch.cuda.synchronize()
    if case != "scattered":
        for k in ("x", "u", "dual", "status", "kkt", "obj"):
            assert torch.equal(eng2.ts[k], one[k]), (case, "second grid", k)
