#!/usr/bin/env python3
"""bench.py -- NMPC solves/s of the batched real-time-iteration kernel on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One STEP = one pass of the hot path over one batch: `alore_nmpc_rti(n_sqp=1)` on
B = 4096 planar ICR-DDR NMPC problems per GPU (BASELINE.json configs[1]: 3
states, 2 controls, horizon N = 20), i.e. for every problem exactly what one
control tick of the reference does (acado_preparationStep + acado_feedbackStep),
started from the cold start of MpcWrapper::solve (x <- x0 replicated, u <- 0).
Inputs are synthetic (alore_legged_manipulator_amd/scenarios.py),
resident in HBM before the timed region; every timed step works on its own
fresh copy of the batch ("slot") so that all steps do identical work and read
their inputs from HBM rather than from cache.  The steps are therefore
independent batches, and alore_nmpc_rti_many keeps --overlap of them in flight
(forked streams); `in_order` in the line is the same run with
one launch at a time.

Multi-GPU (weak scaling): every rank owns its own B problems (block partition
of the global index).  The path has no exchange step, so the headline keeps the
results sharded -- no data-path collective; passes in which every rank also
receives every trajectory (x, u, status, kkt by RCCL all-gather: of the last
batch, or of every batch in asynchronous buckets) are reported beside it under
`result_exchange`.

Prints ONE JSON line on rank 0 (see README/DESIGN for the fields).
"""
from __future__ import annotations

import argparse
import json
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP32_PEAK_TFLOPS = 157.3  # vector FP32 spec peak


def algorithmic_bytes_per_solve(N: int) -> int:
    """Full reference I/O contract per RTI-solve, float32:
    read 44N+21, write 7N+7 floats."""
    return 4 * (51 * N + 28)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=4096, help="problems per GPU")
    ap.add_argument("--horizon", type=int, default=20)
    ap.add_argument("--lanes", type=int, default=0, help="lanes per problem (0 = auto)")
    ap.add_argument("--warm-start-steps", type=int, default=-1, help="working-set prediction steps (-1 = library default)")
    ap.add_argument("--gather", choices=("last", "full", "none", "both"), default="both",
                    help="multi-GPU: the headline keeps results sharded (the path has no exchange step); beside it, passes that "
                         "all-gather the trajectories of the last timed step (last), of every step in overlapped buckets with "
                         "eager launches (full), both (default) or none")
    ap.add_argument("--gather-every", type=int, default=16, help="steps per all-gather bucket")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--graph", choices=("auto", "always"), default="auto",
                    help="auto: a timed region that is ONE grid (alore_nmpc_rti_many, groups) is launched eagerly -- a one-node hipGraph "
                         "only adds its launch latency -- and regions of K launches replay as a graph; always: graphs everywhere")
    ap.add_argument("--overlap", type=int, default=16, choices=tuple(range(1, 33)),
                    help="independent batches (steps) kept in flight at once by alore_nmpc_rti_many; 1 = strictly in order")
    ap.add_argument("--many-mode", choices=("groups", "streams"), default="groups",
                    help="how alore_nmpc_rti_many keeps the steps in flight: one grid over all of them (default) or one launch per "
                         "batch on forked streams")
    ap.add_argument("--no-converged", action="store_true", help="skip the converged-solve + all-gather pass")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="seconds every host thread works in the baseline leg")
    ap.add_argument("--no-extras", action="store_true", help="skip latency / converged-solve extras")
    ap.add_argument("--steady-order", choices=("first", "after"), default="first",
                    help="a short run's 200-step steady-state pass before (default) or after the contract region and its repeats")
    ap.add_argument("--contract-repeats", type=int, default=5, help="further timed regions that bracket the contract figure (0 = none)")
    ap.add_argument("--no-stress-check", action="store_true",
                    help="skip the parity check on the stress distribution (counter passes: its grids run the general-weights path and would be averaged in)")
    ap.add_argument("--bench-events", action="store_true",
                    help="time the single-grid passes with events recorded by bench.py around the Python call (the placement of round 4) instead of "
                         "the library's own events around the hipLaunchKernel")
    ap.add_argument("--headline", choices=("rti", "converged_all_gather"), default="rti",
                    help="which pass the line's `value` is: rti = sharded real-time iterations (default, BASELINE's metric on configs[1]); "
                         "converged_all_gather = the north star's unit -- every step solves its batch to convergence (15 real-time iterations in one "
                         "launch) and all-gathers the converged trajectories to every rank (RCCL), K = --steps, W = --warmup, so that a scaling run "
                         "plots a number that contains the collective; the other pass stays in the line beside it")
    ap.add_argument("--workload", choices=("planar", "whole_body"), default="planar",
                    help="planar = BASELINE configs[1] (the headline); whole_body = configs[2] (B2+Z1, one line with an MFMA roofline block)")
    return ap.parse_args()


def parity_spot_check(eng, batch, N, warmup, K, hooks, per_slot=32, tol=1e-4):
    """Outside every timed region: reload the cold-start iterate, solve the K timed slots exactly as the timed pass does
    (one alore_nmpc_rti_many call), fetch the first, middle and last of them and compare `per_slot` seeded problems of each
    with the CPU oracle (oracle/nmpc_oracle.c, bit-exact with the compiled reference at N = 50) on x and u, relative to
    max(1, |ref|_inf) like the parity tests.  The oracle is the checker here, never the thing measured."""
    from alore_legged_manipulator_amd.scenarios import problem
    from oracle.drivers import Oracle
    eng.load(batch, slot=None)
    hooks.sync()
    eng.rti_range(warmup, K)
    hooks.sync()
    orc = Oracle(N)
    B = batch["x"].shape[0]
    slots = sorted({warmup, warmup + K // 2, warmup + K - 1})
    worst, n = 0.0, 0
    for si, s in enumerate(slots):
        out = eng.fetch(names=("x", "u", "status"), slot=s)
        picks = np.random.default_rng([20260206, si]).choice(B, size=min(per_slot, B), replace=False)
        for b in picks:
            orc.reset(); orc.initialize_solver(); orc.load(problem(batch, int(b))); orc.preparation_step()
            ok = orc.feedback_step() == 0 and int(out["status"][b]) == 0
            for k in ("x", "u"):
                ref = orc.v[k]
                err = float(np.max(np.abs(out[k][b].reshape(-1) - ref)) / max(1.0, float(np.max(np.abs(ref)))))
                worst = max(worst, err if ok else float("inf"))
            n += 1
    return {"worst_rel": worst, "problems": n, "slots": slots, "tolerance": tol, "ok": bool(worst < tol),
            "against": "oracle/nmpc_oracle.c (float32 restatement, bit-exact with the compiled reference at N = 50)"}


def parity_stress_check(eng, N, warmup, K, hooks, per_slot=32, random_per_slot=700):
    """The same check on the STRESS distribution (scenarios.make_wide_batch: far-off poses, references beyond the bounds, full and
    log-uniform weights, random bounds -- long working-set iterations, the general-weights path of the kernel): the slots are
    loaded with it and solved by the same one-call pass the timed region uses (timed here too, reported, never the headline).
    The rule of tests/test_gpu_parity.py::check_against_oracle_and_float64, shares included:
      * per problem: within 1e-4 of the oracle, or else ("loose") closer than the oracle to the float64 minimiser of the oracle's own
        condensed QP and within 1e-4 of it; never beyond 5e-4 of the oracle unless the oracle itself misses that minimiser by more
        than 1e-4 (then 1e-3: "oracle off");
      * shares: loose <= 1 % and oracle off <= 1 in 400, both over picks DRAWN AT RANDOM (`random_per_slot` problems of each of three
        slots: an unbiased sample, large enough that a true share of 0.5 % does not trip the 1 % bound by chance);
      * on top, the `per_slot // 2` problems with the most working-set iterations of each slot (hard-picked, so not part of a share)
        under the per-problem rule, and `per_slot // 2` of the random ones against the float64 minimiser whatever their distance."""
    import time
    from scipy.optimize import lsq_linear
    from alore_legged_manipulator_amd.scenarios import make_wide_batch, problem
    from oracle.drivers import Oracle
    B = eng.B
    wide = make_wide_batch(B, N, 20261004)
    eng.load(wide, slot=None)
    hooks.sync()
    eng.rti_range(0, warmup)
    hooks.sync()
    t0 = time.perf_counter()
    eng.rti_range(warmup, K)
    hooks.sync()
    el = time.perf_counter() - t0
    orc = Oracle(N)
    slots = sorted({warmup, warmup + K // 2, warmup + K - 1})
    n_var = 2 * N
    worst_o, worst_t, ok, unsolved = 0.0, 0.0, True, 0
    n_rnd = loose_rnd = off_rnd = n_hard = loose_hard = off_hard = n_f64 = 0

    def float64_errors(b, out):
        H = orc.v["H"].reshape(n_var, n_var).astype(np.float64); H = 0.5 * (H + H.T)
        Lc = np.linalg.cholesky(H)
        r = lsq_linear(Lc.T, -np.linalg.solve(Lc, orc.v["g"].astype(np.float64)),
                       bounds=(orc.v["lb"].astype(np.float64), orc.v["ub"].astype(np.float64)), method="bvls", tol=1e-15, max_iter=2000)
        scale = max(1.0, float(np.max(np.abs(orc.v["u"]))))
        ek = float(np.max(np.abs((out["u"][b].reshape(-1).astype(np.float64) - wide["u"][b].reshape(-1)) - r.x))) / scale
        er = float(np.max(np.abs(orc.v["dx"].astype(np.float64) - r.x))) / scale
        return ek, er

    for si, s in enumerate(slots):
        out = eng.fetch(names=("x", "u", "status", "n_iter"), slot=s)
        unsolved += int((out["status"] != 0).sum())
        rnd = np.random.default_rng([20261004, si]).choice(B, size=min(random_per_slot, B), replace=False).tolist()
        always64 = set(rnd[:per_slot // 2])
        hard = [int(b) for b in np.argsort(-out["n_iter"], kind="stable")[:per_slot - per_slot // 2] if int(b) not in set(rnd)]
        for b, is_hard in [(b, False) for b in rnd] + [(b, True) for b in hard]:
            orc.reset(); orc.initialize_solver(); orc.load(problem(wide, int(b))); orc.preparation_step()
            if orc.feedback_step() != 0 or int(out["status"][b]) != 0:
                ok = False
                continue
            e = 0.0
            for k in ("x", "u"):
                ref = orc.v[k].astype(np.float64)
                e = max(e, float(np.max(np.abs(out[k][b].reshape(-1) - ref) / np.maximum(1.0, np.abs(ref)))))
            worst_o = max(worst_o, e)
            n_rnd, n_hard = n_rnd + (0 if is_hard else 1), n_hard + (1 if is_hard else 0)
            if e >= 1e-4 or is_hard or b in always64:
                ek, er = float64_errors(b, out)
                worst_t = max(worst_t, ek)
                n_f64 += 1
                if e >= 1e-4:
                    loose_rnd, loose_hard = loose_rnd + (0 if is_hard else 1), loose_hard + (1 if is_hard else 0)
                    ok = ok and (ek < er and ek < 1e-4)
                if e >= 5e-4:
                    off_rnd, off_hard = off_rnd + (0 if is_hard else 1), off_hard + (1 if is_hard else 0)
                    ok = ok and (er > 1e-4 and e < 1e-3)
    loose_bound, off_bound = max(1, int(0.01 * n_rnd)), max(1, n_rnd // 400)
    ok = ok and worst_t < 1e-4 and unsolved == 0 and loose_rnd <= loose_bound and off_rnd <= off_bound and off_hard <= 1
    return {"ok": bool(ok), "problems": n_rnd + n_hard, "slots": slots, "worst_rel_vs_oracle": worst_o,
            "random_picks": n_rnd, "beyond_1e-4_of_the_oracle_in_the_random_picks": loose_rnd, "bound_1_percent": loose_bound,
            "beyond_5e-4_oracle_off_in_the_random_picks": off_rnd, "bound_1_in_400": off_bound,
            "hard_picks": n_hard, "beyond_1e-4_of_the_oracle_in_the_hard_picks": loose_hard, "beyond_5e-4_in_the_hard_picks": off_hard,
            "checked_against_the_float64_minimiser": n_f64,
            "worst_rel_vs_float64_minimiser": worst_t, "unsolved_in_the_checked_slots": unsolved, "ms_per_step_on_this_distribution": el / K * 1e3,
            "distribution": "scenarios.make_wide_batch (stress: long working-set iterations, full weights)",
            "rule": "per problem: <= 1e-4 from the oracle, else closer than the oracle to the float64 minimiser of its condensed QP and <= 1e-4 "
                    "from it; <= 5e-4 from the oracle unless the oracle misses that minimiser by > 1e-4 (then <= 1e-3).  Shares over the "
                    "random picks: beyond 1e-4 <= 1 %, beyond 5e-4 <= 1 in 400 (tests/test_gpu_parity.py::check_against_oracle_and_float64)"}


def usable_cores() -> int:
    """Host cores this process may actually run on: affinity mask, capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def cpu_baseline(N: int, batch: dict, seconds: float) -> dict:
    """The oracle (our C restatement of the reference, validated bit-exact against the
    compiled reference at N=50) timed on the host cores: one solver instance per
    thread, distinct problems of the bench batch, repeated RTI ticks from the
    same cold start.  Baseline only -- not the target."""
    import threading

    from alore_legged_manipulator_amd.scenarios import problem
    from oracle.drivers import Oracle, RefAcado, ref_available

    cores = usable_cores()

    def all_cores(Nh, bt, secs):
        """every usable core runs its own solver instance on its own problem for `secs` of wall clock"""
        nb = bt["x"].shape[0]
        probs = [problem(bt, b % nb) for b in range(cores)]  # one distinct problem of the batch per thread
        o = Oracle(Nh)
        o.reset(); o.initialize_solver(); o.load(probs[0])
        t = o.time_rti(200)  # calibrate on one core
        per_tick = t / 200
        chunk = max(50, int(0.05 / max(per_tick, 1e-7)))  # ~50 ms of ticks between looks at the clock
        counts = [0] * cores
        deadline = [0.0]
        def work(tid):
            orc = Oracle(Nh)
            orc.reset(); orc.initialize_solver(); orc.load(probs[tid])
            while time.perf_counter() < deadline[0]:  # bounded by wall-clock, whatever the host gives each thread
                orc.time_rti(chunk)
                counts[tid] += chunk
        t0 = time.perf_counter()
        deadline[0] = t0 + secs
        th = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
        [x.start() for x in th]
        [x.join() for x in th]
        wall = time.perf_counter() - t0
        return sum(counts) / wall, sum(counts) // max(cores, 1), wall, per_tick

    rate, iters, wall, per_tick = all_cores(N, batch, seconds)
    out = {"value": rate, "unit": "solves/s", "cores": cores, "kind": "port",
           "sample": f"{cores} problems of the bench batch (one per thread) x ~{iters} RTI ticks each (N={N}, "
                     f"MpcWrapper::solve cold start), oracle/nmpc_oracle.c -O3, {cores} threads (usable cores: affinity "
                     f"mask and cgroup quota), {wall:.1f} s wall",
           "single_core_us_per_solve": per_tick * 1e6}
    if N != 50:  # the reference's own horizon: the same restatement on all cores at N = 50
        try:
            from alore_legged_manipulator_amd.scenarios import make_batch
            r50, it50, w50, pt50 = all_cores(50, make_batch(cores, 50), min(4.0, seconds))
            out["n50_all_cores"] = {"value": r50, "unit": "solves/s", "cores": cores, "single_core_us_per_solve": pt50 * 1e6,
                                    "sample": f"N=50, ~{it50} ticks per thread, {w50:.1f} s wall"}
        except Exception as e:  # pragma: no cover
            out["n50_all_cores"] = {"error": f"{type(e).__name__}: {e}"}
    if ref_available():  # the reference's own code, N = 50 only: reported next to it, for scale
        try:
            from alore_legged_manipulator_amd.scenarios import make_batch
            r = RefAcado()
            r.reset(); r.initialize_solver(); r.load(problem(make_batch(1, r.N), 0))
            tr = r.time_rti(300) / 300
            out["reference_n50_single_core_us_per_solve"] = tr * 1e6
        except Exception as e:  # pragma: no cover
            out["reference_n50_error"] = str(e)
    return out


def backend_cpu_baseline(fts, seconds: float = 8.0) -> dict:
    """oracle/backend_oracle.c `minco_plan` (float64 C restatement of MSPlanner::minco_plan, kind "port": Eigen is not
    in the image) on the usable host cores, one planner instance per thread, distinct problems of the bench set, for a
    bounded wall time.  The reference's own budget is 0.05 s per plan (planner_sim.launch:65)."""
    import threading
    from oracle.backend_driver import BackendOracle, EsdfGrid
    cores = usable_cores()
    grid = EsdfGrid.free(half=20.0)
    counts = [0] * cores
    t0 = time.perf_counter()
    deadline = t0 + seconds
    def work(tid):
        o = BackendOracle()
        i = tid
        while time.perf_counter() < deadline and i < len(fts):
            o.minco_plan(grid, fts[i])   # ctypes call: releases the GIL
            counts[tid] += 1
            i += cores
    th = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
    [x.start() for x in th]
    [x.join() for x in th]
    wall = time.perf_counter() - t0
    n = sum(counts)
    return {"value": n / wall, "unit": "plans/s", "cores": cores, "kind": "port",
            "sample": f"{n} plans of the bench problem set, oracle/backend_oracle.c minco_plan (-O3), {cores} threads, {wall:.1f} s wall",
            "ms_per_plan_per_core": wall * cores / max(n, 1) * 1e3, "reference_budget_ms_per_plan": 50.0}


def ltv_cpu_baseline(lt_cfg, st0, xr, dr, n_relin: int, robots: int = 3) -> dict:
    """oracle/ltv_mpc_oracle.py `get_cmd` (the reference's QP matrices line by line + a dense active-set solve, NumPy
    float64, kind "port": OSQP is not vendored) on `robots` robots of the bench set, one thread.  The reference's own
    budget is a 9.7 ms wall clock per tick (mpc_controller/config/mpc3ms.yaml:4,11)."""
    from oracle import ltv_mpc_oracle as lo
    prm = lo.LtvParams()
    t0 = time.perf_counter()
    for b in range(robots):
        out = np.zeros((2, prm.T)); buff = [np.zeros(2) for _ in range(max(prm.delay_num, 0))]
        lo.get_cmd(list(st0[b]) + [0.0], out, buff, xr[b].T.copy(), dr[b].T.copy(), prm, n_relin)
    per = (time.perf_counter() - t0) / robots
    return {"value": 1.0 / per, "unit": "robot-ticks/s", "cores": 1, "kind": "port",
            "sample": f"{robots} robots of the bench set x {n_relin} relinearisations, oracle/ltv_mpc_oracle.py (NumPy float64), one thread",
            "ms_per_tick": per * 1e3, "reference_budget_ms_per_tick": 9.7}


def _wb_lin_stage(args):
    from oracle.wb_oracle import Model, linearize
    x, u = args
    return linearize(Model(), x, u, 0.01)


def wb_cpu_baseline(N, xi, ui, x0, xref, uref):
    """The float64 NumPy oracle (kind "port": there is no reference implementation of this class) on ONE whole problem:
    the linearisation of all N stages (central differences of the spatial-algebra step), the stages spread over the
    usable host cores, then the dense KKT solve of its LQ problem on one core."""
    import multiprocessing as mp
    from oracle.wb_oracle import solve_lq
    from wb_cases import weights as wb_weights
    Q, R, QN = wb_weights()
    cores = min(usable_cores(), N)
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        AB = pool.map(_wb_lin_stage, [(xi[0, k], ui[0, k]) for k in range(N)])
    t_lin = time.perf_counter() - t0
    A = [ab[0] for ab in AB]; Bm = [ab[1] for ab in AB]
    d = [np.zeros(48) for _ in range(N)]
    gx = [Q * (xi[0, k] - xref[0, k]) for k in range(N)]; gu = [R * (ui[0, k] - uref[0, k]) for k in range(N)]
    t0 = time.perf_counter()
    solve_lq(A, Bm, d, np.diag(Q), np.diag(R), np.diag(QN), gx, gu, QN * (xi[0, N] - xref[0, N]), x0[0] - xi[0, 0])
    t_qp = time.perf_counter() - t0
    return {"value": 1.0 / (t_lin + t_qp), "unit": "solves/s", "cores": cores, "kind": "port",
            "sample": f"one whole problem: oracle/wb_oracle.py linearize on all {N} stages over {cores} processes ({t_lin:.1f} s) + dense "
                      f"KKT solve on one core ({t_qp:.2f} s); NumPy float64"}


def stage_roofline(n_items: int, ms: float) -> dict:
    """wb::stage_kernel is bound by float64 vector issue: one wavefront per (problem, stage) executes a fixed number of
    vector instructions (SQ_INSTS_VALU per wavefront from the committed counter pass, profiles/wb_stage_valu.json), a
    double-precision wave-instruction occupies its SIMD for 4 cycles (FP64 vector peak 78.6 TFLOP/s = half the FP32
    rate): peak = 1024 SIMDs x 2.4 GHz / 4."""
    per_wave, src = 12000.0, None
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "wb_stage_valu.json")))
        per_wave, src = float(rec["valu_instructions_per_wavefront"]), rec.get("source")
    except Exception:
        pass
    peak = 1024 * 2.4e9 / 4 / 1e9
    ach = n_items * per_wave / (ms * 1e-3) / 1e9
    return {"bound": "valu_f64", "achieved": ach, "peak": peak, "unit": "G wave-instructions/s", "frac": ach / peak,
            "kernel": "wb::stage_kernel", "kernel_ms_avg": ms, "valu_instructions_per_wavefront": per_wave, "counter_source": src}


def whole_body_main(a, rank, world, local_rank, torch, dist):
    """BASELINE configs[2]: B = 4096 B2 + Z1 whole-body problems per GPU, N = 20; a step = one real-time iteration
    (linearisation kernel + Riccati kernel) of every problem from the same iterate.  Independent problems: ranks are
    replicas of the same shard size (weak scaling), no collective."""
    from alore_legged_manipulator_amd.whole_body import BatchedWholeBody
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from wb_cases import make_problems_fast, weights as wb_weights
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)
    B, N = a.batch, a.horizon
    eng = BatchedWholeBody(B, N, 0.01, device=local_rank)
    x0, xref, uref, xi, ui = make_problems_fast(B, N, seed=3 + rank)
    eng.set_weights(*wb_weights())
    eng.set_problem(x0, xref, uref)
    steps, warm = min(a.steps, 50), min(a.warmup, 5)
    lin = ric = 0.0
    for i in range(warm):
        eng.set_iterate(xi, ui); eng.rti(1)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    t_host = 0.0
    for i in range(steps):
        th = time.perf_counter()
        eng.set_iterate(xi, ui)          # every step starts from the same cold iterate (host upload, not part of the step)
        torch.cuda.synchronize(dev)
        t_host += time.perf_counter() - th
        eng.rti(1)
        l, r_ = eng.last_times()         # HIP events around the two kernels; synchronises
        lin += l; ric += r_
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0 - t_host
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt[0].item())
    dx, du = eng.last_step()
    if rank == 0:
        lin /= steps; ric /= steps
        mfma_flops = N * ((15 + 15) * 12 + (6 * 3 + 6) * 8) * 2048.0 * B
        tf = mfma_flops / (ric * 1e-3) / 1e12
        print(json.dumps({
            "metric": "whole_body_rti_solves_per_s", "value": B * world * steps / elapsed, "unit": "solves/s", "n_gpus": world,
            "steps": steps, "warmup": warm, "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64 dynamics / f32 MFMA Riccati", "data": "synthetic",
            "config": {"workload": f"B={B} per GPU B2+Z1 whole-body NMPC (floating base + 18 joints, nx 48, nu 30), N={N}, one "
                                   "real-time iteration (linearise + Riccati + step) per problem", "batch_per_gpu": B, "horizon": N,
                       "parallelism": f"independent replicas x{world}"},
            "roofline": {"bound": "mfma", "achieved": tf, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP32_PEAK_TFLOPS,
                         "traffic": None, "kernel": "wb::riccati_kernel", "kernel_ms_avg": ric,
                         "mfma_instructions_per_problem_stage": 552},
            "roofline_stage_kernel": stage_roofline(B * N, lin),
            "kernels_ms": {"wb::stage_kernel": lin, "wb::riccati_kernel": ric},
            "finite": bool(np.isfinite(dx).all() and np.isfinite(du).all()),
            "cpu_baseline": None if a.no_cpu_baseline else wb_cpu_baseline(N, xi, ui, x0, xref, uref)}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        a.gpus = world

    import torch
    import torch.distributed as dist

    if a.workload == "whole_body":
        return whole_body_main(a, rank, world, local_rank, torch, dist)

    from alore_legged_manipulator_amd.nmpc import BatchedNmpc
    from alore_legged_manipulator_amd.scenarios import make_batch
    from alore_legged_manipulator_amd.shard import ResultGatherer

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)

    B, N = a.batch, a.horizon
    # a short run (the driver's --steps 20 is 0.4 ms of GPU work) also times LONG_STEPS back-to-back steps and reports
    # them beside the contract figure (`steady_state`)
    LONG_STEPS = 200
    long_steps = LONG_STEPS if (world == 1 and a.steps < LONG_STEPS and a.steps > 0) else 0
    slots = max(a.steps, long_steps) + a.warmup
    # rank r owns global problems [r*B, (r+1)*B): generated locally from the seeded stream
    batch = make_batch(B, N, offset=rank * B)
    eng = BatchedNmpc(B, N, device=local_rank, lanes_per_problem=a.lanes, slots=slots,
                      warm_start_steps=a.warm_start_steps)
    eng.set_launch_overlap(a.overlap)
    eng.set_many_mode(a.many_mode)
    gatherer = ResultGatherer(dist, world) if (world > 1 and a.gather != "none") else None
    ge = max(1, a.gather_every)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    class GpuHooks:
        """device side of shard.timed_pass: HIP stream synchronisation, hipGraph capture, HIP events"""
        single_grid = False  # the next timed region is one grid of the stage-block kernel (set by main before each pass)

        def sync(self):
            torch.cuda.synchronize(dev)

        def barrier(self):
            barrier()

        def capture(self, fn):
            # the K timed steps captured once into a hipGraph (K kernel nodes, no host launch overhead inside the timed
            # region).  Capture is thread-local so that the RCCL watchdog thread of a multi-rank run cannot invalidate
            # it; any capture failure falls back to eager launches.
            if a.no_graph or (self.single_grid and a.graph == "auto"):
                return None
            try:
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                        fn()
                torch.cuda.current_stream(dev).wait_stream(side)
                return g.replay
            except Exception as e:
                print(f"[bench] graph capture unavailable ({type(e).__name__}: {e}); launching eagerly", file=sys.stderr)
                torch.cuda.synchronize(dev)
                return None

        lib_events = False   # the next timed region is timed by the library's own HIP events around its one launch (set by main)

        def device_timer(self):
            if self.lib_events:
                class N:  # nothing to record here: two more event records would only stand between the barrier and the launch
                    def start(self_inner): pass
                    def stop(self_inner): pass
                    def elapsed_ms(self_inner): return 0.0
                return N()

            class T:  # HIP events on the launch stream; read after the barrier that follows the timed region
                def __init__(self_inner):
                    # torch creates the HIP event at its first record(): done here, before the barrier that opens the timed region
                    # (creating the two events took 50 - 75 us of the region's host clock)
                    self_inner.e0 = torch.cuda.Event(enable_timing=True)
                    self_inner.e1 = torch.cuda.Event(enable_timing=True)
                    self_inner.e0.record(); self_inner.e1.record()

                def start(self_inner):
                    self_inner.e0.record()

                def stop(self_inner):
                    self_inner.e1.record()

                def elapsed_ms(self_inner):
                    return self_inner.e0.elapsed_time(self_inner.e1)
            return T()

        def max_over_ranks(self, values):
            if world == 1:
                return list(values)
            tt = torch.tensor(list(values), dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return [float(v) for v in tt]

    from alore_legged_manipulator_amd import shard as shard_mod
    hooks = GpuHooks()

    def timed_pass(mode, K):
        return shard_mod.timed_pass(eng, batch, mode, K, a.warmup, ge, gatherer, hooks, world)

    # primary figure: the shards are independent (robots do not interact inside the solve), so there is no data-path
    # collective -- every rank steps its own problems, the barriers and the max over the ranks of the contract remain.
    # Passes WITH a result exchange (every rank receives every trajectory) are reported beside it (`result_exchange`)
    primary = "none"
    one_grid = a.overlap > 1 and a.many_mode == "groups" and a.lanes in (0, 0x104)
    hooks.single_grid = one_grid
    # a short run's LONG_STEPS pass goes first: the contract pass (W warm-up steps, barrier, K timed steps, barrier) then starts on a
    # GPU that has been busy for a millisecond instead of one that idled through the set-up of this process
    # Passes that are ONE grid are timed by the library itself (alore_nmpc_set_timing): HIP events on the launch stream recorded by the
    # same C call directly before and after the hipLaunchKernel of the grid.  Events recorded by the bench around the Python call would
    # also contain the host's time inside the call -- 8 .. 10 us of an idle GPU waiting for its dispatch, which is not the kernel -- and
    # cost the region's host clock two more event records.  `--bench-events` keeps the round-4 placement.
    lib_events = one_grid and world == 1 and a.graph == "auto" and not a.bench_events   # eager launches only: events inside a stream capture cannot be read

    def grid_event_ms(region_ms):
        if not lib_events:
            return region_ms, None
        kms = float(eng.launch_info()["last_kernel_ms"])
        if not kms > 0.0:
            raise RuntimeError("the library's launch events returned no duration")
        return kms, None
    if lib_events:
        eng.set_timing(True)
        hooks.lib_events = True
    steady = None

    def steady_pass():
        el_l, dms_l, g_l = timed_pass("none", long_steps)
        dms_l, reg_l = grid_event_ms(dms_l)
        return {"steps": long_steps, "value": float(B) * long_steps / el_l, "ms_per_step": el_l / long_steps * 1e3,
                "kernel_ms_avg": dms_l / long_steps, "hip_graph": g_l,
                "hbm_frac": algorithmic_bytes_per_solve(N) * B / (dms_l / long_steps * 1e-3) / 1e9 / HBM_PEAK_GBS}
    if long_steps and a.steady_order == "first":
        steady = steady_pass()
    if os.environ.get("ALORE_BENCH_PRE_SLEEP_MS"):   # diagnostic (profiles/r06_contract_first_region.txt): idle time in front of the contract pass
        time.sleep(float(os.environ["ALORE_BENCH_PRE_SLEEP_MS"]) * 1e-3)
    elapsed, dev_ms, used_graph = timed_pass(primary, a.steps)
    dev_ms, region_events_ms = grid_event_ms(dev_ms)
    # The contract figure is ONE region of K steps measured once (one grid of ~0.13 ms).  `value` stays that first region; five further
    # regions -- slots reloaded, the same W warm-up steps, the same barriers, the same K -- bracket it.
    repeats = None
    if world == 1 and a.steps > 0 and a.contract_repeats > 0:
        r_ms, r_k = [], []
        for _ in range(a.contract_repeats):
            el_r, dms_r, _g = timed_pass(primary, a.steps)
            dms_r, _ = grid_event_ms(dms_r)
            r_ms.append(el_r / a.steps * 1e3)
            r_k.append(dms_r)
        med = lambda v: float(np.median(np.asarray(v)))
        bps = algorithmic_bytes_per_solve(N) * B
        repeats = {"regions": a.contract_repeats, "steps": a.steps, "warmup": a.warmup,
                   "ms_per_step": {"min": min(r_ms), "median": med(r_ms), "max": max(r_ms), "first_region_the_value_is_from": elapsed / a.steps * 1e3},
                   "kernel_ms_per_region": {"min": min(r_k), "median": med(r_k), "max": max(r_k), "first_region": dev_ms},
                   "hbm_frac_by_kernel_events": {"best": bps * a.steps / (min(r_k) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                 "median": bps * a.steps / (med(r_k) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                 "worst": bps * a.steps / (max(r_k) * 1e-3) / 1e9 / HBM_PEAK_GBS},
                   "what": "further timed regions after the one `value` is from: slots reloaded, W warm-up steps, barrier, K steps, barrier; "
                           "kernel time by the library's HIP events around the grid" if lib_events else
                           "further timed regions after the one `value` is from; device time by HIP events around the region"}
    if long_steps and a.steady_order == "after":
        steady = steady_pass()
    eng.set_timing(False)
    hooks.lib_events = False
    host_breakdown = dict(getattr(hooks, "last_host_breakdown_us", {}))
    info = eng.launch_info()
    # the same K steps strictly one after the other (one launch in flight): what a single launch costs, and the figure the
    # rounds before the overlap quoted
    in_order = None
    if world == 1 and a.overlap > 1 and a.steps > 0:
        eng.set_launch_overlap(1)
        hooks.single_grid = False
        el_o, dms_o, g_o = timed_pass("none", a.steps)
        eng.set_launch_overlap(a.overlap)
        hooks.single_grid = one_grid
        in_order = {"value": float(B) * a.steps / el_o, "ms_per_step": el_o / a.steps * 1e3, "kernel_ms_avg": dms_o / a.steps,
                    "hip_graph": g_o, "lanes_per_problem": eng.launch_info()["lanes_per_problem"],
                    "hbm_frac": algorithmic_bytes_per_solve(N) * B / (dms_o / a.steps * 1e-3) / 1e9 / HBM_PEAK_GBS}
    # The north star's unit beside the headline: every step solves its batch to CONVERGENCE (15 real-time iterations in
    # one launch) and all-gathers the converged trajectories (x, u, status, kkt) to every rank with RCCL, the collective of
    # step i running under the solve of step i + 1.  Same pass at every world size (world = 1: the gather is a local copy),
    # so a scaling run compares like with like; `value` of the line keeps the sharded real-time-iteration unit at every N.
    conv = None
    if a.steps > 0 and not a.no_converged:
        try:
            head_conv = a.headline == "converged_all_gather"
            Kc = a.steps if head_conv else min(a.steps, 50)   # as the headline: exactly K steps after W warm-up steps
            Wc = a.warmup if head_conv else min(a.warmup, 5)
            cg = ResultGatherer(dist, world, depth=2)
            el_c, dms_c, _ = shard_mod.timed_pass(eng, batch, "converged", Kc, Wc, ge, cg, hooks, world, conv_iters=15)
            last = cg.wait()
            slot_last = Wc + Kc - 1
            mine_ok = bool(torch.equal(last["x"][rank], eng.ts["x"][slot_last]) and torch.equal(last["status"][rank], eng.ts["status"][slot_last]))
            flags = hooks.max_over_ranks([0.0 if mine_ok else 1.0])
            per_rank_bytes = 4 * B * ((N + 1) * 3 + N * 2 + 2)
            conv = {"metric": "converged NMPC solves/s with every rank holding every trajectory", "value": float(B) * world * Kc / el_c,
                    "unit": "solves/s", "steps": Kc, "warmup": Wc, "ms_per_step": el_c / Kc * 1e3, "kernel_ms_avg": dms_c / Kc,
                    "real_time_iterations_per_solve": 15,
                    "rccl_ranks": int(dist.get_world_size()) if world > 1 else 1,
                    "gathered_status_sum": int(last["status"].sum().item()), "gathered_problems": int(last["status"].numel()),
                    "every_rank_holds_its_own_slab_in_the_gather": flags[0] == 0.0,
                    "all_gather_bytes_in_per_rank_per_step": per_rank_bytes * (world - 1),
                    "what": "per step: one launch of 15 real-time iterations (MpcWrapper::solve cold start to convergence), then x, u, status, kkt "
                            "all-gathered to every rank (RCCL, asynchronous, overlapped with the next step's launch); all collectives complete "
                            "inside the timed region"}
            # the same unit with the steps of a bucket (`--gather-every` independent batches) solved by ONE grid and gathered by ONE
            # collective per tensor: what the headline's pass does for single iterations (the packed lane mapping fills the chip)
            try:
                cg2 = ResultGatherer(dist, world, depth=2)
                el_f, dms_f, _ = shard_mod.timed_pass(eng, batch, "converged_in_flight", Kc, Wc, ge, cg2, hooks, world, conv_iters=15)
                last2 = cg2.wait()
                ok2 = bool(torch.equal(last2["x"][rank][-1], eng.ts["x"][slot_last]) and torch.equal(last2["status"][rank][-1], eng.ts["status"][slot_last]))
                flags2 = hooks.max_over_ranks([0.0 if ok2 else 1.0])
                conv["in_flight"] = {"value": float(B) * world * Kc / el_f, "ms_per_step": el_f / Kc * 1e3, "kernel_ms_avg": dms_f / Kc,
                                     "steps_per_grid": min(ge, Kc), "gathered_status_sum": int(last2["status"].sum().item()),
                                     "every_rank_holds_its_own_slab_in_the_gather": flags2[0] == 0.0,
                                     "what": "the steps of a bucket are independent batches: ONE grid of nmpc::rti_block_kernel runs the 15 real-time "
                                             "iterations of all of them (alore_nmpc_rti_many), ONE all-gather per tensor and bucket follows and runs under "
                                             "the next bucket's grid"}
            except Exception as e:  # pragma: no cover
                conv["in_flight"] = {"error": f"{type(e).__name__}: {e}"}
        except Exception as e:  # pragma: no cover
            conv = {"error": f"{type(e).__name__}: {e}"}

    exchange = None
    if world > 1 and a.gather != "none":
        # secondary figures: they must never cost the primary one (a rank that fails here would hang the others in the
        # collective, so the decision to run them is taken before, not inside, and errors are reported in the line)
        exchange = {}
        for mode, label in (("last", "last_batch"), ("full", "every_batch")):
            if a.gather not in (mode, "both"):
                continue
            try:
                el2, dms2, g2 = timed_pass(mode, a.steps)
                exchange[label] = {"value": float(B) * world * a.steps / el2, "ms_per_step": el2 / a.steps * 1e3,
                                   "kernel_ms_avg": dms2 / a.steps, "hip_graph": g2,
                                   "what": ("x, u, status, kkt of the last timed batch all-gathered to every rank inside the timed region"
                                            if mode == "last" else
                                            f"x, u, status, kkt of EVERY timed batch all-gathered to every rank in buckets of {ge} steps, "
                                            "eager launches in order")}
            except Exception as e:  # pragma: no cover
                exchange[label] = {"error": f"{type(e).__name__}: {e}"}

    # parity of what was timed: the headline pass is run once more (untimed; the in_order / steady_state passes reloaded the
    # slots) and 32 seeded problems of each of 3 timed slots are compared with the CPU oracle at BASELINE.json's 1e-4
    spot = None
    if rank == 0 and a.steps > 0:
        spot = parity_spot_check(eng, batch, N, a.warmup, a.steps, hooks)

    # every timed step must have solved every problem
    st = eng.ts["status"][a.warmup:a.warmup + a.steps]
    n_bad = int((st != 0).sum().item())
    n_iter_mean = float(eng.ts["n_iter"][a.warmup:a.warmup + a.steps].float().mean().item())

    # ... and the same timed configuration on the stress distribution (outside every timed region; the slots are reloaded after)
    stress = None
    if rank == 0 and a.steps > 0 and world == 1 and not a.no_stress_check:
        try:
            stress = parity_stress_check(eng, N, a.warmup, a.steps, hooks)
        except Exception as e:  # pragma: no cover
            stress = {"ok": False, "error": f"{type(e).__name__}: {e}"}
        eng.load(batch, slot=None)
        hooks.sync()

    if world > 1:
        nb = torch.tensor([n_bad], dtype=torch.int64, device=dev)
        dist.all_reduce(nb)
        n_bad = int(nb.item())

    result = None
    if rank == 0:
        total_solves = float(B) * world * a.steps
        value = total_solves / elapsed
        ms_per_step = elapsed / a.steps * 1e3
        # average launch duration of the dominant kernel: HIP events on the launch stream around the K back-to-back
        # launches (multi-rank: of the pass without collectives between the launches)
        kern_ms = dev_ms / a.steps
        if a.overlap > 1 and a.many_mode == "groups" and (info["lanes_per_problem"] & 0x100):
            n_launches, per_launch = 1, a.steps   # the slots sit at constant strides: one grid for the K batches
        else:
            n_launches, per_launch = a.steps, 1
        bytes_per_launch = algorithmic_bytes_per_solve(N) * B
        achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9
        # HBM bytes per launch from the PMC counters: collected by rocprofv3 in separate --pmc passes over this same
        # command (tools/profile_round.sh, corrected as MI355X_MICROARCH.md prescribes) and kept in profiles/: a
        # profiler cannot run inside the timed process, so the line names the record it quotes
        traffic, traffic_src = None, None
        tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tf):
            try:
                rec = json.load(open(tf)).get(f"B{B}_N{N}", {})
                kname = "rti_block_kernel" if (info["lanes_per_problem"] & 0x100) else "rti_kernel"
                if rec.get("kernel", kname).find(kname) >= 0:   # only a record of the kernel that ran
                    traffic = rec.get("bytes_per_launch")
                    traffic_src = rec.get("source")
            except Exception:
                traffic = None
        flops_per_solve = 350.0 * N + n_iter_mean * 160.0 * N  # stage-wise algorithm, see DESIGN.md
        result = {
            "metric": "nmpc_rti_solves_per_s", "value": value, "unit": "solves/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"B={B} per GPU planar ICR-DDR NMPC (3 states, 2 controls), horizon N={N}, "
                                   "one real-time iteration (prepare+feedback) per problem from the "
                                   "MpcWrapper::solve cold start, full reference I/O contract",
                       "batch_per_gpu": B, "global_batch": B * world, "horizon": N, "dt": 0.01,
                       "parallelism": f"independent shards x{world}, results stay sharded",
                       "lanes_per_problem": info["lanes_per_problem"], "threads_per_block": info["threads_per_block"],
                       "lds_bytes_per_block": info["lds_bytes_per_block"], "hip_graph": used_graph,
                       "launches_in_flight": a.overlap, "many_mode": a.many_mode if a.overlap > 1 else "in order",
                       "resident_problems": B * (min(a.overlap, a.steps) if a.many_mode == "streams" else a.steps) if a.overlap > 1 else B,
                       "headline_is": ("throughput of independent B-problem batches solved together (alore_nmpc_rti_many); "
                                       "the one-batch-at-a-time figure is in_order") if a.overlap > 1 else "one batch at a time"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "kernel": "nmpc::rti_block_kernel" if (info["lanes_per_problem"] & 0x100) else "nmpc::rti_kernel",
                         "kernel_ms_avg": kern_ms,
                         "launches_in_flight": a.overlap,
                         "batches_per_launch": per_launch, "kernel_launches": n_launches,
                         "kernel_ms_per_launch": dev_ms / n_launches,
                         "events": ("HIP events on the launch stream, recorded by the library directly before and after the hipLaunchKernel of the grid "
                                    "(alore_nmpc_set_timing; --bench-events: recorded by bench.py around the Python call instead)" if lib_events
                                    else "HIP events on the launch stream, recorded by bench.py around the timed region"),
                         "note": ("achieved = algorithmic bytes of one B-problem batch / kernel_ms_avg, kernel_ms_avg = HIP-event time of "
                                  "the timed region / K batches.  many_mode groups: ONE grid of nmpc::rti_block_kernel serves the K "
                                  "batches of the timed region (alore_nmpc_rti_many; the slots sit at constant strides), so a profiler's "
                                  "duration of that launch is kernel_ms_per_launch = K x kernel_ms_avg and the ratio bytes / duration is "
                                  "the same.  many_mode streams: one launch per batch on "
                                  "forked streams, per-kernel durations overlap.  in_order has the one-batch-at-a-time figures"
                                  if a.overlap > 1 else
                                  "achieved = algorithmic bytes of one launch / (HIP-event time of the K launches / K)"),
                         "algorithmic_bytes_per_solve": algorithmic_bytes_per_solve(N),
                         "fp32_frac": value / world * flops_per_solve / (FP32_PEAK_TFLOPS * 1e12)},
            "unsolved_problems": n_bad, "working_set_iters_mean": n_iter_mean,
            "host_clock_breakdown_us": host_breakdown,
        }
        if repeats is not None:
            result["contract_repeats"] = repeats
        if spot is not None:
            result["parity_spot_check"] = spot
            if not spot["ok"]:  # a fast kernel whose results differ from the reference's is not measured
                result["value"] = None
                result["error"] = f"parity spot check failed: worst relative error {spot['worst_rel']:.3e} > {spot['tolerance']}"
        if stress is not None:
            result["parity_stress_check"] = stress
            if not stress["ok"]:
                result["value"] = None
                result["error"] = f"parity check on the stress distribution failed: {stress}"
        if conv is not None:
            result["converged_all_gather"] = conv
        if a.headline == "converged_all_gather":
            # the north star's unit as the line's value: the sharded real-time-iteration figures move into `rti_pass`
            if conv is None or "error" in conv:
                result["value"] = None
                result["error"] = f"--headline converged_all_gather: the pass did not run ({conv})"
            else:
                result["rti_pass"] = {k: result[k] for k in ("metric", "value", "unit", "ms_per_step", "roofline")}
                fl = conv.get("in_flight")
                if fl and "error" not in fl and fl["every_rank_holds_its_own_slab_in_the_gather"]:  # the buckets-in-flight figure is the line's
                    conv = dict(conv, value=fl["value"], ms_per_step=fl["ms_per_step"], kernel_ms_avg=fl["kernel_ms_avg"])
                kms = conv["kernel_ms_avg"]
                ach = algorithmic_bytes_per_solve(N) * B / (kms * 1e-3) / 1e9
                result.update({"metric": "nmpc_converged_solves_per_s_all_gathered", "value": conv["value"], "ms_per_step": conv["ms_per_step"],
                               "steps": conv["steps"], "warmup": conv["warmup"]})
                result["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                                      "kernel": "nmpc::rti_block_kernel (15 real-time iterations per launch)", "kernel_ms_avg": kms,
                                      "algorithmic_bytes_per_solve": algorithmic_bytes_per_solve(N),
                                      "note": "a converged solve reads and writes the I/O contract ONCE and does the arithmetic 15 times: this "
                                              "pass is bound by the instruction issue of the sweeps, the HBM fraction is what it leaves of the roofline"}
                result["config"]["headline_is"] = ("converged solves (15 real-time iterations per launch) with the converged trajectories all-gathered to "
                                                   "every rank inside the timed region (RCCL over xGMI for N > 1), buckets of --gather-every steps per grid and "
                                                   "per collective (converged_all_gather.in_flight; the block's own figures are one step per launch); rti_pass has "
                                                   "the sharded real-time-iteration figure")
        if exchange:
            result["result_exchange"] = exchange
        if steady is not None:
            result["steady_state"] = steady
        if in_order is not None:
            result["in_order"] = in_order

    # ---- extras on rank 0 of a single-GPU run: latency, converged solves, large batch, CPU baseline
    if rank == 0 and world == 1 and not a.no_extras:
        extras = {}
        # p50/p99 of one synchronous launch (what a caller of the C ABI sees): 1000 launches after 50 discarded,
        # B = 4096 and small batches
        lat = {}
        for b_lat in (B, 64, 1):
            hb = batch if b_lat == B else make_batch(b_lat, N)
            e2 = eng if b_lat == B else BatchedNmpc(b_lat, N, device=local_rank)
            if b_lat != B:
                e2.load(hb, slot=None)
            xs0, us0, ds0 = (torch.from_numpy(hb[k]).to(dev) for k in ("x", "u", "dual"))
            ts = []
            for i in range(1050):
                e2.ts["x"][0].copy_(xs0); e2.ts["u"][0].copy_(us0); e2.ts["dual"][0].copy_(ds0)
                torch.cuda.synchronize(dev)
                t_a = time.perf_counter()
                e2.rti_sync(1, slot=0)   # alore_nmpc_rti + alore_nmpc_synchronize (the launch stream; torch.cuda.synchronize waits for the whole device)
                ts.append(time.perf_counter() - t_a)
            ts = np.array(ts[50:]) * 1e3
            lat[f"B{b_lat}"] = {"launches": int(ts.size), "p50_ms": float(np.percentile(ts, 50)), "p99_ms": float(np.percentile(ts, 99)),
                                "max_ms": float(ts.max())}
        result["latency"] = {"what": "one synchronous real-time iteration from the MpcWrapper::solve cold start (alore_nmpc_rti + alore_nmpc_synchronize: enqueue + kernel + "
                                     "stream synchronisation), host clock of this Python process through ctypes", **lat}
        # the same pair from a C++ host (the reference's controller is C++): tools/micro/rti_latency, built by csrc/Makefile
        try:
            import subprocess
            exe = os.path.join(ROOT, "tools", "micro", "rti_latency")
            cpp = {}
            for b_lat in (1, 64, B):
                out = subprocess.run([exe, str(b_lat)], capture_output=True, text=True, timeout=120).stdout
                m = re.search(r"enqueue p50 ([0-9.]+) us; .* p50 ([0-9.]+)  p99 ([0-9.]+) us; status (\d+)", out)
                cpp[f"B{b_lat}"] = {"enqueue_p50_ms": float(m.group(1)) * 1e-3, "p50_ms": float(m.group(2)) * 1e-3, "p99_ms": float(m.group(3)) * 1e-3,
                                    "status": int(m.group(4))}
            result["latency"]["cpp_host"] = {"what": "the same pair called from C++ through the C ABI (tools/micro/rti_latency.cpp), 2000 cold-start solves of a "
                                                     "bound-hitting problem", **cpp}
        except Exception as e:  # pragma: no cover
            result["latency"]["cpp_host"] = {"error": f"{type(e).__name__}: {e}"}
        # converged solve: K = 15 real-time iterations inside one launch (MpcWrapper::solve + 14 update() calls of the
        # reference collapsed into one kernel: the I/O contract is read and written once, the arithmetic 15 times)
        e3 = BatchedNmpc(B, N, device=local_rank, slots=52)
        e3.load(batch, slot=None)
        e3.rti(15, slot=0); e3.rti(15, slot=1)
        torch.cuda.synchronize(dev)
        ev_a = torch.cuda.Event(enable_timing=True); ev_b = torch.cuda.Event(enable_timing=True)
        ev_a.record()
        for i in range(2, 52):
            e3.rti(15, slot=i)
        ev_b.record(); torch.cuda.synchronize(dev)
        ms15 = ev_a.elapsed_time(ev_b) / 50
        it15 = float(e3.ts["n_iter"][2:52].float().mean().item())
        gb15 = algorithmic_bytes_per_solve(N) * B / (ms15 * 1e-3) / 1e9
        fl15 = 15 * 350.0 * N + it15 * 160.0 * N
        extras["converged_solve_k15"] = {"ms_per_launch": ms15, "solves_per_s": B / (ms15 * 1e-3)}
        result["roofline_converged_k15"] = {
            "bound": "hbm", "achieved": gb15, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gb15 / HBM_PEAK_GBS,
            "traffic": None, "kernel": f"nmpc::rti_block_kernel, {e3.launch_info()['lanes_per_problem'] & 0xff} lanes per problem, multi-iteration build (n_sqp = 15)",
            "kernel_ms_avg": ms15, "launches": 50,
            "algorithmic_bytes_per_solve": algorithmic_bytes_per_solve(N),
            "working_set_iters_last_iteration_mean": it15,
            "fp32_frac": B / (ms15 * 1e-3) * fl15 / (FP32_PEAK_TFLOPS * 1e12),
            "unsolved": int((e3.ts["status"][2:52] != 0).sum().item()),
            "note": "same I/O bytes as one real-time iteration, 15 x the arithmetic: latency/issue bound, not HBM bound"}
        del e3
        # large batch (BASELINE configs[3] per-GPU share: 262144 / 8)
        Bl = 32768
        try:
            e4 = BatchedNmpc(Bl, N, device=local_rank, slots=24)
            e4.load(make_batch(Bl, N), slot=None)
            e4.rti(1, slot=0); e4.rti(1, slot=1)
            torch.cuda.synchronize(dev)
            ev_a.record()
            for i in range(2, 24):
                e4.rti(1, slot=i)
            ev_b.record(); torch.cuda.synchronize(dev)
            msl = ev_a.elapsed_time(ev_b) / 22
            gbs = algorithmic_bytes_per_solve(N) * Bl / (msl * 1e-3) / 1e9
            extras["large_batch"] = {"batch": Bl, "ms_per_launch": msl, "solves_per_s": Bl / (msl * 1e-3),
                                     "hbm_GBs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS,
                                     "lanes_per_problem": e4.launch_info()["lanes_per_problem"]}
            del e4
        except Exception as e:  # pragma: no cover
            extras["large_batch"] = {"error": str(e)}
        # warm tick (closed-loop regime): converge first (K = 15), disturb the measured state by a millimetre /
        # milliradian, then time ONE real-time iteration from the carried iterate and duals -- no working-set
        # prediction, normally a single Riccati sweep
        try:
            e5 = BatchedNmpc(B, N, device=local_rank, slots=24)
            e5.load(batch, slot=None)
            for i in range(24):
                e5.rti(15, slot=i)
            e5.ts["x0"].add_(1e-3)
            torch.cuda.synchronize(dev)
            keep = {k: e5.ts[k].clone() for k in ("x", "u", "dual")}   # the converged, disturbed state every timed leg starts from
            e5.rti(1, slot=0); e5.rti(1, slot=1)
            torch.cuda.synchronize(dev)
            ev_a.record()
            for i in range(2, 24):
                e5.rti(1, slot=i)
            ev_b.record(); torch.cuda.synchronize(dev)
            msw = ev_a.elapsed_time(ev_b) / 22
            nit_eager = float(e5.ts["n_iter"][2:24].float().mean().item())
            uns_eager = int((e5.ts["status"][2:24] != 0).sum().item())
            # the same 22 launches replayed from a hipGraph (the eager loop is bound by the host's launch calls): the slots are
            # put back to the converged, disturbed state before the capture, before the untimed replay and before the timed one
            msw_graph, nit_graph, uns_graph = None, None, None
            try:
                def restore():
                    for k in keep:
                        e5.ts[k].copy_(keep[k])
                    torch.cuda.synchronize(dev)
                restore()
                side5 = torch.cuda.Stream(device=dev)
                g5 = torch.cuda.CUDAGraph()
                side5.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side5):
                    with torch.cuda.graph(g5, stream=side5):
                        for i in range(2, 24):
                            e5.rti(1, slot=i)
                torch.cuda.current_stream(dev).wait_stream(side5)
                torch.cuda.synchronize(dev)
                restore(); g5.replay(); torch.cuda.synchronize(dev)
                restore()
                ev_a.record(); g5.replay(); ev_b.record(); torch.cuda.synchronize(dev)
                msw_graph = ev_a.elapsed_time(ev_b) / 22
                nit_graph = float(e5.ts["n_iter"][2:24].float().mean().item())
                uns_graph = int((e5.ts["status"][2:24] != 0).sum().item())
            except Exception as e:  # pragma: no cover
                msw_graph = None
            extras["warm_tick"] = {"ms_per_launch": msw, "ms_per_launch_graph": msw_graph, "solves_per_s": B / (msw * 1e-3),
                                   "working_set_iters_mean": nit_eager, "unsolved": uns_eager,
                                   "working_set_iters_mean_graph": nit_graph, "unsolved_graph": uns_graph}
            del e5
        except Exception as e:  # pragma: no cover
            extras["warm_tick"] = {"error": str(e)}
        # the whole control tick of the reference node for B robots (host controller -> C ABI -> kernels ->
        # wheel-speed commands; mpc.cpp CmdCallback); trajectories live in the device store, references are
        # sampled there
        try:
            from alore_legged_manipulator_amd.host import BatchedMpcController, Polynome
            rng = np.random.default_rng(7)
            vw = rng.uniform([0.5, -1.0], [1.8, 1.0], (B, 2))
            ctl = BatchedMpcController(B, N, 0.01, device=local_rank, max_pieces=8, max_checkpoints=64)
            t_a = time.perf_counter()
            for b in range(B):
                v, w = vw[b]
                T = np.array([0.5, 0.5, 0.5, 0.5]); Tc = np.cumsum(T)
                ctl.robots[b].traj(Polynome(np.stack([w * Tc[:-1], v * Tc[:-1]], 1), T, [0, 0, w, v, 0, 0],
                                            [w * Tc[-1], v * Tc[-1], w, v, 0, 0], [0, 0, 0], [-0.3, 0.3, 0.1], 0.0))
                ctl.robots[b].odom(0.01, -0.01, 0.02)
                ctl.robots[b].icr(-0.3, 0.3, 0.1)
            ctl.tick(0.05)
            t_first = time.perf_counter() - t_a   # includes the B trajectory messages -> device store
            ctl.tick(0.06)
            t_a = time.perf_counter()
            nt = 30
            for i in range(nt):
                ctl.tick_full(0.07 + 0.01 * i)   # the C entry point (alore_host_controller_tick); no Python unpacking of the B commands
            t_b = time.perf_counter()
            # a tick in which EVERY robot has a new trajectory message (the messages are handed over before the clock
            # starts: the Python loop that builds them is not the library's time): B spline solves + checkpoint tables on
            # the device inside the tick
            for b in range(B):
                v, w = vw[b]
                T = np.array([0.5, 0.5, 0.5, 0.5]); Tc = np.cumsum(T)
                ctl.robots[b].traj(Polynome(np.stack([w * Tc[:-1], v * Tc[:-1]], 1), T, [0, 0, w, v, 0, 0],
                                            [w * Tc[-1], v * Tc[-1], w, v, 0, 0], [0, 0, 0], [-0.3, 0.3, 0.1], 0.07 + 0.01 * nt))
            t_c = time.perf_counter()
            ctl.tick_full(0.08 + 0.01 * nt)
            t_replan = time.perf_counter() - t_c
            extras["controller_tick"] = {"robots": B, "ms_per_tick": (t_b - t_a) / nt * 1e3,
                                         "robot_ticks_per_s": B * nt / (t_b - t_a),
                                         "tick_with_a_new_trajectory_for_every_robot_ms": t_replan * 1e3,
                                         "first_tick_with_trajectory_messages_ms": t_first * 1e3}
            del ctl
        except Exception as e:  # pragma: no cover
            extras["controller_tick"] = {"error": f"{type(e).__name__}: {e}"}
        # back_end: B_be Monte-Carlo goals planned in one launch (MSPlanner::minco_plan per problem), then handed to
        # the NMPC store device-to-device and tracked for 100 closed-loop ticks (BASELINE configs[4], one GPU's share)
        try:
            from alore_legged_manipulator_amd.backend import BatchedMSPlanner
            from alore_legged_manipulator_amd.flat_traj import monte_carlo_goals
            Bb = 2048
            fts = monte_carlo_goals(Bb, seed=44)
            pl = BatchedMSPlanner(Bb, 16, device=local_rank)
            pl.set_free_map(half=20.0)
            pl.set_problems(fts)
            pl.plan(); torch.cuda.synchronize(dev)
            t_a = time.perf_counter()
            pl.plan()
            res = pl.results()
            t_b = time.perf_counter()
            pl_ms = pl.last_plan_ms()
            be = {"problems": Bb, "ms_per_launch_device": pl_ms, "ms_wall_incl_results": (t_b - t_a) * 1e3,
                  "plans_per_s": Bb / (pl.last_plan_ms() * 1e-3), "ok": int(res["ok"].sum()),
                  "cost_evaluations_mean": float(res["evals"].mean()), "pieces_mean": float(res["n_pieces"].mean())}
            e10 = BatchedNmpc(Bb, N, device=local_rank, diagnostics=False)
            e10.load({k: batch[k][:Bb] for k in ("W", "WN", "lbValues", "ubValues")})
            e10.refs_init(max_pieces=16, max_checkpoints=128)
            t_a = time.perf_counter()
            e10.refs_set_from_backend(pl)
            torch.cuda.synchronize(dev)
            be["handover_to_nmpc_store_ms"] = (time.perf_counter() - t_a) * 1e3
            icr = np.array([[(0.0, -0.3, 0.3), (0.2, -0.3, 0.3), (0.1, -0.25, 0.25), (0.3, -0.35, 0.35)][b % 4] for b in range(Bb)])
            e10.plant_init()
            e10.plant_set_state(np.array([ft.start_xytheta for ft in fts]), icr)
            e10.closed_loop_reset()
            e10.closed_loop_tick(0.05)
            torch.cuda.synchronize(dev)
            t_a = time.perf_counter()
            e10.closed_loop_run(0.06, 0.01, 100)
            torch.cuda.synchronize(dev)
            be["closed_loop_100_ticks_ms"] = (time.perf_counter() - t_a) * 1e3
            be["unsolved_last_tick"] = int((e10.t["status"] != 0).sum().item())
            # float64 vector issue model of backend::backend_kernel (one wavefront per plan): instructions per wavefront from
            # the committed counter pass, a double-precision wave-instruction holds its SIMD for 4 cycles
            try:
                rec = json.load(open(os.path.join(ROOT, "profiles", "backend_valu.json")))
                ach = Bb * float(rec["valu_instructions_per_wavefront"]) / (pl_ms * 1e-3) / 1e9
                peak = 1024 * 2.4e9 / 4 / 1e9
                be["roofline"] = {"bound": "valu_f64", "achieved": ach, "peak": peak, "unit": "G wave-instructions/s", "frac": ach / peak,
                                  "kernel": "backend::backend_kernel", "kernel_ms_avg": pl_ms,
                                  "valu_instructions_per_wavefront": float(rec["valu_instructions_per_wavefront"]),
                                  "counter_source": rec.get("source"),
                                  "note": "HBM is idle (2.5 KB in / out per plan); a launch lasts as long as its slowest plan"}
            except Exception:
                pass
            if not a.no_cpu_baseline:
                try:
                    be["cpu_baseline"] = backend_cpu_baseline(fts)
                except Exception as e:  # pragma: no cover
                    be["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
            extras["backend_to_nmpc_pipeline"] = be
            del e10, pl
            # the same pipeline at the size ONE GPU of BASELINE configs[4] carries: 4 object classes x 16384 poses / 8 GPUs = 8192 plans
            # (about four residencies of the planner kernel), planned in one launch, handed over device to device, 100 closed-loop ticks
            try:
                B8 = 8192
                fts8 = monte_carlo_goals(B8, seed=44)
                pl8 = BatchedMSPlanner(B8, 16, device=local_rank)
                pl8.set_free_map(half=20.0)
                pl8.set_problems(fts8)
                pl8.plan(); torch.cuda.synchronize(dev)
                t_a = time.perf_counter()
                pl8.plan()
                res8 = pl8.results()
                t_b = time.perf_counter()
                be8 = {"problems": B8, "ms_per_launch_device": pl8.last_plan_ms(), "ms_wall_incl_results": (t_b - t_a) * 1e3,
                       "plans_per_s": B8 / (pl8.last_plan_ms() * 1e-3), "ok": int(res8["ok"].sum()),
                       "cost_evaluations_mean": float(res8["evals"].mean()), "cost_evaluations_max": int(res8["evals"].max())}
                e11 = BatchedNmpc(B8, N, device=local_rank, diagnostics=False)
                W8 = np.tile(np.diag([10, 10, 0.5, 0.1, 0.1]).astype(np.float32), (B8, N, 1, 1))
                WN8 = np.tile(np.diag([10, 10, 0.5]).astype(np.float32), (B8, 1, 1))
                e11.load({"W": W8, "WN": WN8})
                e11.refs_init(max_pieces=16, max_checkpoints=128)
                t_a = time.perf_counter()
                e11.refs_set_from_backend(pl8)
                torch.cuda.synchronize(dev)
                be8["handover_to_nmpc_store_ms"] = (time.perf_counter() - t_a) * 1e3
                icr8 = np.array([[(0.0, -0.3, 0.3), (0.2, -0.3, 0.3), (0.1, -0.25, 0.25), (0.3, -0.35, 0.35)][b % 4] for b in range(B8)])
                e11.plant_init()
                e11.plant_set_state(np.array([ft.start_xytheta for ft in fts8]), icr8)
                e11.closed_loop_reset()
                e11.closed_loop_tick(0.05)
                torch.cuda.synchronize(dev)
                t_a = time.perf_counter()
                e11.closed_loop_run(0.06, 0.01, 100)
                torch.cuda.synchronize(dev)
                be8["closed_loop_100_ticks_ms"] = (time.perf_counter() - t_a) * 1e3
                be8["unsolved_last_tick"] = int((e11.t["status"] != 0).sum().item())
                try:
                    rec = json.load(open(os.path.join(ROOT, "profiles", "backend_valu.json")))
                    ach = B8 * float(rec["valu_instructions_per_wavefront"]) / (pl8.last_plan_ms() * 1e-3) / 1e9
                    be8["valu_f64_issue_frac"] = ach / (1024 * 2.4e9 / 4 / 1e9)
                except Exception:
                    pass
                extras["backend_to_nmpc_pipeline_8192"] = be8
                del e11, pl8
            except Exception as e:  # pragma: no cover
                extras["backend_to_nmpc_pipeline_8192"] = {"error": f"{type(e).__name__}: {e}"}
        except Exception as e:  # pragma: no cover
            extras["backend_to_nmpc_pipeline"] = {"error": f"{type(e).__name__}: {e}"}
        # whole-body class (BASELINE configs[2]): B = 4096 B2 + Z1 problems, N = 20, one real-time iteration =
        # linearisation kernel (RNEA-based, float64) + Riccati kernel (float32 MFMA 16x16x4)
        try:
            from alore_legged_manipulator_amd.whole_body import BatchedWholeBody
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from wb_cases import make_problems_fast, weights as wb_weights
            Bw = B
            from alore_legged_manipulator_amd.whole_body import model_info
            wbe_effort = model_info()["effort"]
            wbe = BatchedWholeBody(Bw, N, 0.01, device=local_rank)
            x0w, xrefw, urefw, xiw, uiw = make_problems_fast(Bw, N, seed=3)
            wbe.set_weights(*wb_weights())
            wbe.set_problem(x0w, xrefw, urefw)
            wbe.set_iterate(xiw, uiw)
            wbe.rti(1); torch.cuda.synchronize(dev)
            wbe.set_iterate(xiw, uiw)
            t_a = time.perf_counter()
            wbe.rti(1); torch.cuda.synchronize(dev)
            t_b = time.perf_counter()
            lin_ms, ric_ms = wbe.last_times()
            dxw, duw = wbe.last_step()
            x1w, u1w = wbe.get_iterate()
            sat = float(np.mean(np.any(np.abs(u1w[:, :, :18]) >= wbe_effort - 1e-9, axis=2)))   # stages with a torque on its limit
            wbe.rti(1); torch.cuda.synchronize(dev)     # the next iteration from the updated iterate (closed-loop regime)
            lin2_ms, ric2_ms = wbe.last_times()
            # matrix-core flops of the Riccati sweep per problem: per stage 30 tiles with K = 48 (12 MFMAs each; the
            # symmetric products only their upper-triangular tiles) and 24 tiles with K = 32 (8 each) = 552
            # v_mfma_f32_16x16x4_f32 of 2048 flops (SQ_INSTS_MFMA agrees: profiles/)
            mfma_flops = N * ((15 + 15) * 12 + (6 * 3 + 6) * 8) * 2048.0
            # the same iteration with the constraint terms on: friction pyramids + the contact-consistency penalty on J_c v
            wbe.set_contact_constraints(True, 0.7)
            wbe.set_contact_penalty(1000.0)
            wbe.set_iterate(xiw, uiw)
            wbe.rti(1); torch.cuda.synchronize(dev)
            lin3_ms, ric3_ms = wbe.last_times()
            con_ok = bool((wbe.status() == 0).all())
            # hard contact rows (the forces eliminated by the rows, then the sweep) and the refined penalty step (round 4)
            wbe.set_contact_constraints(False)
            wbe.set_contact_penalty(0.0)
            wbe.set_contact_rows(True)
            wbe.set_iterate(xiw, uiw)
            wbe.rti(1); torch.cuda.synchronize(dev)
            lin4_ms, ric4_ms = wbe.last_times()
            rows_ok = bool((wbe.status() == 0).all())
            wbe.set_contact_rows(False)
            wbe.set_torque_limits(False)
            wbe.set_contact_penalty(2000.0)
            wbe.set_refinement(1)
            wbe.set_iterate(xiw, uiw)
            wbe.rti(1); torch.cuda.synchronize(dev)
            lin5_ms, ric5_ms = wbe.last_times()
            ref_ok = bool((wbe.status() == 0).all())
            extras["whole_body_b2z1"] = {"problems": Bw, "horizon": N, "ms_linearize": lin_ms, "ms_riccati": ric_ms,
                                         "ms_wall_one_rti": (t_b - t_a) * 1e3, "solves_per_s": Bw / ((lin_ms + ric_ms) * 1e-3),
                                         "mfma_f32_TFLOPs_riccati": Bw * mfma_flops / (ric_ms * 1e-3) / 1e12,
                                         "mfma_f32_frac_of_peak": Bw * mfma_flops / (ric_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
                                         "stages_with_a_saturated_torque_frac": sat,
                                         "second_iteration": {"ms_linearize": lin2_ms, "ms_riccati": ric2_ms,
                                                              "solves_per_s": Bw / ((lin2_ms + ric2_ms) * 1e-3)},
                                         "with_hard_contact_rows": {"ms_linearize_and_row_elimination": lin4_ms, "ms_riccati_and_forces": ric4_ms,
                                                                    "solves_per_s": Bw / ((lin4_ms + ric4_ms) * 1e-3), "all_steps_applied": rows_ok},
                                         "penalty_2000_with_one_refinement_step": {"ms_linearize": lin5_ms, "ms_riccati_twice_and_residuals": ric5_ms,
                                                                                   "solves_per_s": Bw / ((lin5_ms + ric5_ms) * 1e-3), "all_steps_applied": ref_ok},
                                         "with_contact_constraints_and_penalty": {"ms_linearize": lin3_ms, "ms_riccati": ric3_ms,
                                                                                  "solves_per_s": Bw / ((lin3_ms + ric3_ms) * 1e-3), "all_steps_applied": con_ok},
                                         "first_step_max": float(np.max(np.abs(dxw))), "finite": bool(np.isfinite(dxw).all() and np.isfinite(duw).all())}
            del wbe
        except Exception as e:  # pragma: no cover
            extras["whole_body_b2z1"] = {"error": f"{type(e).__name__}: {e}"}
        # LTV-MPC class (the `mpc` node): getCmd for B robots, 5 relinearisation passes, cold and warm working sets
        try:
            from alore_legged_manipulator_amd.ltv_mpc import BatchedLtvMpc
            rng = np.random.default_rng(11)
            lt = BatchedLtvMpc(B, device=local_rank)
            T = lt.cfg.predict_steps
            vv, ww = rng.uniform(0.5, 2.5, B), rng.uniform(-1.5, 1.5, B)
            ts_ = (np.arange(T) + 1) * lt.cfg.dt
            xr = np.stack([vv[:, None] / ww[:, None] * np.sin(ww[:, None] * ts_), vv[:, None] / ww[:, None] * (1 - np.cos(ww[:, None] * ts_)),
                           ww[:, None] * ts_], 2)
            dr = np.stack([np.repeat(vv[:, None], T, 1), np.repeat(ww[:, None], T, 1)], 2)
            st0 = rng.uniform(-0.1, 0.1, (B, 3))
            lt.set_refs(xr, dr)
            g0 = lt.get_cmd(st0, n_relin=5, reset=True)       # untimed: first launch of the kernel + the sweep counts of a cold start
            # a control tick returns the command column and the status (16 B per robot): cold = freshly constructed controller
            # (reset: no previous output, free working set), warm = the tick after
            t_a = time.perf_counter(); lt.tick(st0, n_relin=5, reset=True); t_cold = time.perf_counter() - t_a
            # the ten ticks after a reset still settle their working sets (kernel 0.40, 0.22, 0.20, ... 0.16 ms: profiles/r04_c_ltv_*);
            # warm = the ticks after those, what a running controller pays
            t_a = time.perf_counter()
            for i in range(10):
                cmd_w, st_w = lt.tick(st0, n_relin=5)     # commands + status only (16 B per robot over the bus)
            t_settle = (time.perf_counter() - t_a) / 10
            t_a = time.perf_counter()
            for i in range(20):
                cmd_w, st_w = lt.tick(st0, n_relin=5)
            t_warm = (time.perf_counter() - t_a) / 20
            g1 = lt.get_cmd(st0, n_relin=5)
            extras["ltv_mpc"] = {"robots": B, "relinearisations": 5, "cold_ms": t_cold * 1e3, "warm_ms": t_warm * 1e3,
                                 "first_ten_ticks_after_reset_ms": t_settle * 1e3,
                                 "robot_ticks_per_s_warm": B / t_warm, "sweeps_last_qp_cold_mean": float(g0["sweeps"].mean()),
                                 "sweeps_last_qp_warm_mean": float(g1["sweeps"].mean()), "unsettled": int((g1["status"] != 0).sum())}
            if not a.no_cpu_baseline:
                try:
                    extras["ltv_mpc"]["cpu_baseline"] = ltv_cpu_baseline(lt.cfg, st0, xr, dr, 5)
                except Exception as e:  # pragma: no cover
                    extras["ltv_mpc"]["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
            del lt
        except Exception as e:  # pragma: no cover
            extras["ltv_mpc"] = {"error": f"{type(e).__name__}: {e}"}
        # Monte-Carlo closed-loop rollout with no host in the loop: per tick, references from the plant's pose ->
        # one real-time iteration -> command into the simulator plant (alore_nmpc_closed_loop_tick); closed_loop_run issues ONE launch
        # per tick: [plant step of tick t - 1 -> solve of tick t] with the pose-independent sampling of tick t + 1 on extra workgroups
        # of the same grid (results of the ticks one by one: tests/test_closed_loop.py)
        try:
            from alore_legged_manipulator_amd.host import Polynome
            rng = np.random.default_rng(7)
            vw = rng.uniform([0.5, -1.0], [1.8, 1.0], (B, 2))
            T = np.array([1.0, 1.0, 1.0]); Tc = np.cumsum(T)
            msgs = [Polynome(np.stack([w * Tc[:-1], v * Tc[:-1]], 1), T, [0, 0, w, v, 0, 0], [w * Tc[-1], v * Tc[-1], w, v, 0, 0],
                             [0, 0, 0], [-0.3, 0.3, 0.1], 0.0) for v, w in vw]
            e7 = BatchedNmpc(B, N, device=local_rank, diagnostics=False)   # like the reference's tick
            e7.load({k: batch[k] for k in ("W", "WN", "lbValues", "ubValues")})
            e7.refs_init(max_pieces=4, max_checkpoints=40)
            e7.refs_set_polynomes(np.arange(B), msgs)
            e7.plant_init()
            e7.plant_set_state(np.zeros((B, 3)), np.tile([0.1, -0.3, 0.3], (B, 1)))
            for t in range(20):
                e7.closed_loop_tick(0.01 * (t + 1))
            torch.cuda.synchronize(dev)
            nt = 200
            t_a = time.perf_counter()
            e7.closed_loop_run(0.01 * 21, 0.01, nt)
            torch.cuda.synchronize(dev)
            t_b = time.perf_counter()
            pose, _, goal = e7.plant_get_state()
            extras["device_closed_loop"] = {"robots": B, "ticks": nt, "ms_per_tick": (t_b - t_a) / nt * 1e3,
                                            "robot_ticks_per_s": B * nt / (t_b - t_a),
                                            "launches_per_tick": 1,
                                            "unsolved_last_tick": int((e7.t["status"] != 0).sum().item()),
                                            "finite": bool(np.isfinite(pose).all())}
            del e7
            # other fleet sizes: 64 and 512 robots (the (32, 1) / (16, 2) mappings: one launch per tick), 8192 (one GPU's share of
            # configs[4]; the (8, 3) mapping has no room for the sampler's wavefronts: plant step + solve in one launch, the sampler behind)
            sizes = {}
            for Bf in (64, 512, 8192):
                ef = BatchedNmpc(Bf, N, device=local_rank, diagnostics=False)
                bf = make_batch(Bf, N)
                ef.load({k: bf[k] for k in ("W", "WN", "lbValues", "ubValues")})
                ef.refs_init(max_pieces=4, max_checkpoints=40)
                mf = [msgs[i % B] for i in range(Bf)]
                ef.refs_set_polynomes(np.arange(Bf), mf)
                ef.plant_init()
                ef.plant_set_state(np.zeros((Bf, 3)), np.tile([0.1, -0.3, 0.3], (Bf, 1)))
                ef.closed_loop_run(0.01, 0.01, 20)
                torch.cuda.synchronize(dev)
                t_a = time.perf_counter()
                ef.closed_loop_run(0.01 * 21, 0.01, nt)
                torch.cuda.synchronize(dev)
                sizes[str(Bf)] = (time.perf_counter() - t_a) / nt * 1e3
                del ef
            extras["device_closed_loop"]["ms_per_tick_other_fleet_sizes"] = sizes
        except Exception as e:  # pragma: no cover
            extras["device_closed_loop"] = {"error": f"{type(e).__name__}: {e}"}
        # compact I/O (secondary figure): weights, bounds and ICR parameters shared by the batch
        # (one object class, many poses): 4 (19 N + 19) bytes per solve instead of 4 (51 N + 28)
        try:
            e8 = BatchedNmpc(B, N, device=local_rank, slots=24)
            cb = dict(batch); cb["od"] = np.broadcast_to(batch["od"][:1], batch["od"].shape).copy()
            e8.load(cb, slot=None)
            e8.set_shared_members(W=True, bounds=True, od=True)
            e8.rti(1, slot=0); e8.rti(1, slot=1)
            torch.cuda.synchronize(dev)
            ev_a.record()
            for i in range(2, 24):
                e8.rti(1, slot=i)
            ev_b.record(); torch.cuda.synchronize(dev)
            msc = ev_a.elapsed_time(ev_b) / 22
            cbytes = 4 * (19 * N + 19)
            extras["compact_io"] = {"ms_per_launch": msc, "solves_per_s": B / (msc * 1e-3), "bytes_per_solve": cbytes,
                                    "hbm_frac": cbytes * B / (msc * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                    "unsolved": int((e8.ts["status"][2:24] != 0).sum().item())}
            del e8
        except Exception as e:  # pragma: no cover
            extras["compact_io"] = {"error": f"{type(e).__name__}: {e}"}
        # the reference's own route through the QP -- condense to the dense 2N x 2N problem, solve that (acado_condensePrep /
        # condenseFdb + acado_solve) -- on the GPU (alore_nmpc_condense + alore_nmpc_dense_qp), beside the stage-wise solve
        try:
            e10 = BatchedNmpc(B, N, device=local_rank)
            e10.load(batch)
            qp = e10.condense()
            xq, yq, stq, niq = e10.dense_qp(qp["H"], qp["g"], qp["lb"], qp["ub"])
            torch.cuda.synchronize(dev)
            ev_a.record()
            for i in range(5):
                qp = e10.condense()
            ev_b.record(); torch.cuda.synchronize(dev)
            ms_cond = ev_a.elapsed_time(ev_b) / 5
            ev_a.record()
            for i in range(5):
                xq, yq, stq, niq = e10.dense_qp(qp["H"], qp["g"], qp["lb"], qp["ub"])
            ev_b.record(); torch.cuda.synchronize(dev)
            ms_qp = ev_a.elapsed_time(ev_b) / 5
            extras["dense_route"] = {"batch": B, "order": 2 * N, "ms_condense": ms_cond, "ms_dense_qp": ms_qp,
                                     "solves_per_s": B / ((ms_cond + ms_qp) * 1e-3), "factorisations_mean": float(niq.float().mean().item()),
                                     "unsolved": int((stq != 0).sum().item()),
                                     "note": "what the stage-wise kernel replaces: H alone is 4 (2N)^2 bytes per solve"}
            del e10, qp
        except Exception as e:  # pragma: no cover
            extras["dense_route"] = {"error": f"{type(e).__name__}: {e}"}
        # the tick as the reference runs it: acado_getKKT / acado_getObjective are never called by its wrapper;
        # with NULL kkt / obj the kernel skips them (the headline computes them)
        try:
            e9 = BatchedNmpc(B, N, device=local_rank, slots=24, diagnostics=False)
            e9.load(batch, slot=None)
            e9.rti(1, slot=0); e9.rti(1, slot=1)
            torch.cuda.synchronize(dev)
            ev_a.record()
            for i in range(2, 24):
                e9.rti(1, slot=i)
            ev_b.record(); torch.cuda.synchronize(dev)
            msd = ev_a.elapsed_time(ev_b) / 22
            extras["without_diagnostics"] = {"ms_per_launch": msd, "solves_per_s": B / (msd * 1e-3),
                                             "unsolved": int((e9.ts["status"][2:24] != 0).sum().item())}
            del e9
        except Exception as e:  # pragma: no cover
            extras["without_diagnostics"] = {"error": f"{type(e).__name__}: {e}"}
        # round 6: the two-phase grid the round-5 verdict asked for (homogeneous wavefronts: a first pass without the working-set
        # prediction, problems whose working set moves queued for tail workgroups of the same grid) beside the one-pass grid that ships:
        # same bits, and the time of both on 20 batches of the bench batch (library events around the grid, best of 4)
        try:
            res_tp = {}
            for mode in (0, 1):
                e7 = BatchedNmpc(B, N, device=local_rank, slots=20)
                e7.set_two_phase(mode)
                times = []
                for rep in range(5):
                    e7.load(batch, slot=None)
                    torch.cuda.synchronize(dev)
                    e7.set_timing(rep > 0)
                    e7.rti_range(0, 20)
                    torch.cuda.synchronize(dev)
                    if rep > 0:
                        times.append(float(e7.launch_info()["last_kernel_ms"]))
                res_tp[mode] = ({k: e7.ts[k].clone() for k in ("x", "u", "dual", "status", "kkt", "obj")}, min(times), e7.two_phase_info())
                del e7
            same = all(bool(torch.equal(res_tp[0][0][k], res_tp[1][0][k])) for k in res_tp[0][0])
            extras["two_phase_grid"] = {"batches": 20, "one_pass_grid_ms": res_tp[0][1], "two_phase_grid_ms": res_tp[1][1],
                                        "bit_equal_x_u_dual_status_kkt_obj": same, "share_of_problems_queued": res_tp[1][2]["tail_share"],
                                        "two_phase_batches": res_tp[1][2]["two_phase_batches"],
                                        "what": "alore_nmpc_set_two_phase(h, 1) against the default: opt-in, slower on this distribution "
                                                "(a queued problem pays the wait for its inputs twice: profiles/r06_two_phase.txt)"}
        except Exception as e:  # pragma: no cover
            extras["two_phase_grid"] = {"error": f"{type(e).__name__}: {e}"}
        # the reference's own generated horizon (N = 50), same batch size: next to the compiled reference's
        # single-core time in cpu_baseline.reference_n50_single_core_us_per_solve
        try:
            e6 = BatchedNmpc(B, 50, device=local_rank, slots=42)
            b50 = make_batch(B, 50)
            e6.load(b50, slot=None)
            e6.rti(1, slot=0); e6.rti(1, slot=1)
            torch.cuda.synchronize(dev)
            ev_a.record()
            for i in range(2, 12):
                e6.rti(1, slot=i)
            ev_b.record(); torch.cuda.synchronize(dev)
            ms50 = ev_a.elapsed_time(ev_b) / 10
            lanes50 = e6.launch_info()["lanes_per_problem"]
            # the same with launches in flight (alore_nmpc_rti_many, eager): 40 fresh slots
            e6.set_launch_overlap(a.overlap)
            e6.load(b50, slot=None)
            e6.rti_range(0, 2)
            torch.cuda.synchronize(dev)
            ev_a.record()
            e6.rti_range(2, 40)
            ev_b.record(); torch.cuda.synchronize(dev)
            ms50f = ev_a.elapsed_time(ev_b) / 40
            extras["reference_horizon_n50"] = {"batch": B, "ms_per_launch": ms50, "solves_per_s": B / (ms50 * 1e-3),
                                               "lanes_per_problem": lanes50,
                                               "in_flight": {"launches_in_flight": a.overlap, "ms_per_step": ms50f, "solves_per_s": B / (ms50f * 1e-3),
                                                             "hbm_frac": algorithmic_bytes_per_solve(50) * B / (ms50f * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                             "unsolved": int((e6.ts["status"][2:42] != 0).sum().item())}}
            del e6
        except Exception as e:  # pragma: no cover
            extras["reference_horizon_n50"] = {"error": str(e)}
        result["extras"] = extras

    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(N, batch, a.cpu_seconds)

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
