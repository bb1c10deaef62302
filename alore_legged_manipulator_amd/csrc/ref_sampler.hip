// ref_sampler.hip -- reference sampling on the device.
//
// What the reference does on the host every control tick before the solver runs
//   MpcController::getRefPoints   P/nmpc_controller/src/mpc.cpp:407-461
//   TrajAnal::getPstate/getVstate P/nmpc_controller/include/nmpc_controller/traj_anal.hpp:105-134
//   Trajectory<5,2>::getPos/getVel P/back_end/include/gcopter/trajectory.hpp:75-103, 472-502
//   MpcController::smooth_yaw     P/nmpc_controller/src/mpc.cpp:248-277
//   MpcWrapper::setTrajectory / setICRParameters  P/nmpc_controller/src/mpc_wrapper.cpp:200-207, 242-264
// is done here for B robots by two small kernels that write y, yN, od and x0 of the batch directly, so
// that a tick uploads 24 bytes of odometry per robot instead of 5N+... floats of references.
// The trajectory itself (quintic coefficients of the minimum-jerk spline and the Simpson checkpoints)
// is prepared on the host when a new Polynome arrives, as in the reference (TrajCallback), and kept in
// device memory.  Arithmetic is float64 like the reference's; the cast to float32 happens on the store,
// as in MpcWrapper::setTrajectory.
#include "nmpc_kernels.h"

#include "minco_spline.h"
#include "ref_sampler_device.h"

namespace nmpc {

using minco::eval_pv;

// one thread per (robot, node j); j = 0..N (ref_sampler_device.h: ref_sample_node)
__global__ void ref_sample_kernel(RefStore s, alore_nmpc_batch b, int B, int N, double dt, double now,
                                  const double* est /* [B][3] */, const double* icr /* [B][3] xv yr yl */, int* at_goal,
                                  double* psi_raw /* [B][N+1] normalised headings for the unwrap pass, or null */)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)B * (N + 1)) return;
    const int r = (int)(t / (N + 1)), j = (int)(t % (N + 1));
    double psi = 0.0;
    if (ref_sample_node(s, b, N, dt, now, est, icr, at_goal, r, j, psi) && psi_raw) psi_raw[t] = psi;
}

// The same with smooth_yaw in the same launch (ref_sampler_device.h: ref_sample_smooth_body)
template <int G, bool AHEAD = false>
__global__ void ref_sample_smooth_kernel(RefStore s, alore_nmpc_batch b, int B, int N, double dt, double now, const double* est,
                                         const double* icr, int* at_goal, double* psi_rel = nullptr)
{
    ref_sample_smooth_body<G, AHEAD>(s, b, B, N, dt, now, est, icr, at_goal, psi_rel, (int)blockIdx.x, (int)threadIdx.x, (int)blockDim.x);
}

// smooth_yaw: sequential in the nodes, one thread per robot; unwraps the float64 headings and casts
// afterwards, like the reference (smooth_yaw, then the float cast in MpcWrapper::setTrajectory).
__global__ void ref_unwrap_kernel(RefStore s, alore_nmpc_batch b, int B, int N, const double* est, const double* psi_raw)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= B) return;
    if (s.meta[(size_t)r * 8 + 6] == 0.0) return;
    const double* pr = psi_raw + (size_t)r * (N + 1);
    double prev = pr[0];
    double dyaw = prev - est[(size_t)r * 3 + 2];
    while (dyaw >= M_PI / 2) { prev -= M_PI * 2; dyaw = prev - est[(size_t)r * 3 + 2]; }
    while (dyaw <= -M_PI / 2) { prev += M_PI * 2; dyaw = prev - est[(size_t)r * 3 + 2]; }
    float* y = const_cast<float*>(b.y) + (size_t)r * N * 5;
    float* yN = const_cast<float*>(b.yN) + (size_t)r * 3;
    if (N > 0) y[2] = (float)prev; else yN[2] = (float)prev;
    for (int i = 0; i < N; ++i) {
        double cur = pr[i + 1];
        dyaw = cur - prev;
        while (dyaw >= M_PI / 2) { cur -= M_PI * 2; dyaw = cur - prev; }
        while (dyaw <= -M_PI / 2) { cur += M_PI * 2; dyaw = cur - prev; }
        if (i + 1 < N) y[(size_t)(i + 1) * 5 + 2] = (float)cur; else yN[2] = (float)cur;
        prev = cur;
    }
}

// TrajAnal::getVstate / getAstate at t_cur for every robot (the if_mpc = false branch of CmdCallback)
__global__ void ref_eval_kernel(RefStore s, int B, double now, double* out)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= B) return;
    const double* m = s.meta + (size_t)r * 8;
    double v[2] = {0, 0}, a[2] = {0, 0}, p[2];
    if (m[6] != 0.0) {
        const double t = now - m[0];
        eval_pv(s.dur + (size_t)r * s.P, s.coef + (size_t)r * s.P * 12, (int)m[4], t, p, v);
        minco::eval_a(s.dur + (size_t)r * s.P, s.coef + (size_t)r * s.P * 12, (int)m[4], t, a);
    }
    out[(size_t)r * 4] = v[0]; out[(size_t)r * 4 + 1] = v[1]; out[(size_t)r * 4 + 2] = a[0]; out[(size_t)r * 4 + 3] = a[1];
}
hipError_t launch_ref_eval(const RefStore& s, int B, double now, double* out, hipStream_t st)
{
    hipLaunchKernelGGL(ref_eval_kernel, dim3((B + 127) / 128), dim3(128), 0, st, s, B, now, out);
    return hipGetLastError();
}

// ---- Polynome messages -> trajectory store, on the device (TrajAnal::setTraj: setConditions /
//      setParameters / getTrajectory, then getSeq; traj_anal.hpp:36-95).  Three kernels:
//      spline (one thread per (message, flat dimension): the SPD knot system of csrc/minco_spline.h, O(M) with
//      2 x 2 blocks, instead of the reference's 6M x 6M band LU), Simpson panels (one thread per (message,
//      panel): independent), checkpoints (one thread per message: the running sum in the reference's order).
__global__ void traj_spline_kernel(RefStore s, PolyBatch m, int count, double res, int res_int, double* knot_ws,
                                   int* n_panels, int* overflow)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= 2 * count) return;
    const int t = g >> 1, d = g & 1; // the two dimensions of a message sit on neighbouring lanes
    const int r = m.robot[t], M = m.n_pieces[t];
    double* meta = s.meta + (size_t)r * 8;
    if (d == 0) n_panels[t] = 0;
    if (M < 1 || M > s.P) {
        if (d == 0) { meta[6] = 0.0; atomicOr(overflow, 1); }
        return;
    }
    const double* T = m.t_pts + (size_t)t * m.P;
    const double* pva = m.pva + (size_t)t * 12; // init p0 p1 v0 v1 a0 a1, tail p0 p1 v0 v1 a0 a1
    // workspace of this thread: knot positions [P + 1], inverted Schur blocks [P] (3 doubles), rhs [2 P], coefficients [6 P]
    double* ws = knot_ws + (size_t)g * traj_ws_doubles(m.P);
    double* kp = ws;
    minco::Sym2* sinv = reinterpret_cast<minco::Sym2*>(ws + (m.P + 1));
    double* y = ws + (m.P + 1) + 3 * m.P;
    double* cf = y + 2 * m.P;
    const double* inner = m.inner + (size_t)t * (m.P > 1 ? m.P - 1 : 1) * 2;
    kp[0] = pva[d];
    for (int k = 1; k < M; ++k) kp[k] = inner[(k - 1) * 2 + d];
    kp[M] = pva[6 + d];
    minco::spline_1d(M, T, kp, pva[2 + d], pva[4 + d], pva[8 + d], pva[10 + d], sinv, y, cf);
    double* coef = s.coef + (size_t)r * s.P * 12;
    for (int i = 0; i < M; ++i)
        for (int k = 0; k < 6; ++k) coef[(size_t)i * 12 + d * 6 + k] = cf[6 * i + k];
    if (d != 0) return;
    double* dur = s.dur + (size_t)r * s.P;
    double total = 0.0;
    for (int i = 0; i < M; ++i) {
        dur[i] = T[i];
        total += T[i];
    }
    const double fine = res / res_int;
    const int seq = (int)floor(total / fine);
    const int nck = 1 + seq / res_int;
    if (nck > s.C) { meta[6] = 0.0; atomicOr(overflow, 2); return; }
    n_panels[t] = seq;
    meta[0] = m.t0[t]; meta[1] = total; meta[2] = m.icr[(size_t)t * 3 + 2]; meta[3] = res;
    meta[4] = (double)M; meta[5] = (double)nck; meta[6] = 1.0; meta[7] = 0.0;
    double* ck = s.ckpt + (size_t)r * s.C * 2;
    ck[0] = m.start[(size_t)t * 3];
    ck[1] = m.start[(size_t)t * 3 + 1];
}

__global__ void traj_panel_kernel(RefStore s, PolyBatch m, int count, double res, int res_int, const int* n_panels,
                                  int max_panels, double* inc)
{
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (long)count * max_panels) return;
    const int t = (int)(g / max_panels), i = (int)(g % max_panels);
    if (i >= n_panels[t]) return;
    const int r = m.robot ? m.robot[t] : t;
    const double* meta = s.meta + (size_t)r * 8;
    const double fine = res / res_int;
    double dx, dy;
    minco::simpson_panel(s.dur + (size_t)r * s.P, s.coef + (size_t)r * s.P * 12, (int)meta[4], meta[2], i * fine, fine, dx, dy);
    inc[g * 2] = dx;
    inc[g * 2 + 1] = dy;
}

__global__ void traj_checkpoint_kernel(RefStore s, PolyBatch m, int count, int res_int, const int* n_panels,
                                       int max_panels, const double* inc)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const int n = n_panels[t];
    const int r = m.robot ? m.robot[t] : t;
    if (n <= 0 && s.meta[(size_t)r * 8 + 6] == 0.0) return;
    double* ck = s.ckpt + (size_t)r * s.C * 2;
    double x = ck[0], y = ck[1];
    const double* in = inc + (size_t)t * max_panels * 2;
    for (int i = 0; i < n; ++i) {
        x += in[i * 2];
        y += in[i * 2 + 1];
        if (i % res_int == res_int - 1) {
            const int c = (i + 1) / res_int;
            ck[c * 2] = x;
            ck[c * 2 + 1] = y;
        }
    }
}

hipError_t launch_traj_build(const RefStore& s, const PolyBatch& m, int count, double res, int res_int, double* knot_ws,
                             int* n_panels, double* inc, int* overflow, hipStream_t st)
{
    const int max_panels = s.C * res_int;
    hipLaunchKernelGGL(traj_spline_kernel, dim3((2 * count + 63) / 64), dim3(64), 0, st, s, m, count, res, res_int, knot_ws,
                       n_panels, overflow);
    const long total = (long)count * max_panels;
    hipLaunchKernelGGL(traj_panel_kernel, dim3((unsigned)((total + 127) / 128)), dim3(128), 0, st, s, m, count, res, res_int,
                       n_panels, max_panels, inc);
    hipLaunchKernelGGL(traj_checkpoint_kernel, dim3((count + 63) / 64), dim3(64), 0, st, s, m, count, res_int, n_panels,
                       max_panels, inc);
    return hipGetLastError();
}

// ---- the planner's result slabs (alore_backend_device_view) -> trajectory store, device to device: what the
//      reference does with a process boundary in between (MSPlanner result -> PlanManager::MPCPathPub ->
//      Polynome -> MpcController::TrajCallback -> TrajAnal::setTraj re-solves the spline): the coefficients the
//      optimiser ended with ARE that spline (same knot system, tests/test_backend_gpu.py), so they are copied, and
//      only the Simpson checkpoints are built.  One thread per (problem, piece) for the copy.
__global__ void traj_from_backend_kernel(RefStore s, BackendView v, int count, double t0, double res, int res_int, double xv,
                                         int* n_panels, int* overflow)
{
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (long)count * v.P) return;
    const int t = (int)(g / v.P), i = (int)(g % v.P);
    const int M = v.n_pieces[t];
    double* meta = s.meta + (size_t)t * 8;
    const bool ok = v.ok[t] != 0 && M >= 1 && M <= s.P;
    if (i == 0) {
        n_panels[t] = 0;
        // a failed replan leaves the robot on the trajectory it is tracking: MSPlanner::minco_plan returns false before
        // final_traj_ is replaced (optimizer.cpp:204-209), nothing is published and the controller carries on.  Only
        // a plan that succeeded but does not fit the store invalidates the slot (and is reported).
        if (!ok && v.ok[t] != 0) { meta[6] = 0.0; atomicOr(overflow, 1); }
    }
    if (!ok || i >= M) return;
    s.dur[(size_t)t * s.P + i] = v.T[(size_t)t * v.P + i];
    const double* c = v.coef + ((size_t)t * v.P + i) * 12; // (6 i + q) * 2 + d
    double* o = s.coef + ((size_t)t * s.P + i) * 12;       // [d][q]
    for (int q = 0; q < 6; ++q) { o[q] = c[q * 2]; o[6 + q] = c[q * 2 + 1]; }
    if (i != 0) return;
    double total = 0.0;
    for (int k = 0; k < M; ++k) total += v.T[(size_t)t * v.P + k];
    const double fine = res / res_int;
    const int seq = (int)floor(total / fine);
    const int nck = 1 + seq / res_int;
    if (nck > s.C) { meta[6] = 0.0; atomicOr(overflow, 2); return; }
    n_panels[t] = seq;
    meta[0] = t0; meta[1] = total; meta[2] = xv; meta[3] = res;
    meta[4] = (double)M; meta[5] = (double)nck; meta[6] = 1.0; meta[7] = 0.0;
    s.ckpt[(size_t)t * s.C * 2] = v.start_xytheta[(size_t)t * 3];
    s.ckpt[(size_t)t * s.C * 2 + 1] = v.start_xytheta[(size_t)t * 3 + 1];
}

hipError_t launch_traj_from_backend(const RefStore& s, const BackendView& v, int count, double t0, double res, int res_int, double xv,
                                    int* n_panels, double* inc, int* overflow, hipStream_t st)
{
    const int max_panels = s.C * res_int;
    const long threads = (long)count * v.P;
    hipLaunchKernelGGL(traj_from_backend_kernel, dim3((unsigned)((threads + 127) / 128)), dim3(128), 0, st, s, v, count, t0, res, res_int,
                       xv, n_panels, overflow);
    PolyBatch m{};
    m.robot = nullptr; // identity: problem t -> store slot t
    const long total = (long)count * max_panels;
    hipLaunchKernelGGL(traj_panel_kernel, dim3((unsigned)((total + 127) / 128)), dim3(128), 0, st, s, m, count, res, res_int, n_panels,
                       max_panels, inc);
    hipLaunchKernelGGL(traj_checkpoint_kernel, dim3((count + 63) / 64), dim3(64), 0, st, s, m, count, res_int, n_panels, max_panels, inc);
    return hipGetLastError();
}

// ---- kinematic plant of the reference's simulator node, B robots (P/utils/simulator/include/simulator/
//      simulator.h:234-275): ControlSubCallback turns the wheel-speed command into (v, vy, omega) through the
//      ICR parameters, StatePropaCallback follows it with bounded acceleration and integrates the pose by
//      explicit Euler steps.  The command is input column `node` of the solver's prediction, or zero once
//      the robot is at its goal (mpc.cpp:186-194).  Deterministic: the simulator's optional Gaussian noise
//      on v and omega (simulator.h:222-229) is not drawn.
__global__ void plant_kernel(alore_nmpc_batch b, int B, int N, int node, const double* icr, const int* at_goal,
                             double* pose, double* vw, PlantParams p)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= B) return;
    double right = (double)b.u[((size_t)r * N + node) * 2], left = (double)b.u[((size_t)r * N + node) * 2 + 1];
    if (at_goal && at_goal[r]) { right = 0.0; left = 0.0; }
    const double xv = icr[(size_t)r * 3], yr = icr[(size_t)r * 3 + 1], yl = icr[(size_t)r * 3 + 2];
    const double desired_v = (left + right) / 2.0 - (right - left) / (yl - yr) * (yl + yr) / 2.0;
    const double vy = -(right - left) / (yl - yr) * xv;
    const double desired_w = (right - left) / (yl - yr);
    double x = pose[(size_t)r * 3], y = pose[(size_t)r * 3 + 1], th = pose[(size_t)r * 3 + 2];
    double v = vw[(size_t)r * 2], w = vw[(size_t)r * 2 + 1];
    plant_substeps(p, desired_v, desired_w, vy, x, y, th, v, w);
    pose[(size_t)r * 3] = x; pose[(size_t)r * 3 + 1] = y; pose[(size_t)r * 3 + 2] = th;
    vw[(size_t)r * 2] = v; vw[(size_t)r * 2 + 1] = w;
}

// The plant step of tick t and what the sampler of tick t + 1 could not do without the pose it produces (closed_loop_run): the
// at-goal flag of tick t (getRefPoints' test, from `now` of tick t), the plant as above, then x0 <- pose and smooth_yaw's first
// step -- node 0 against the measured heading, mpc.cpp:248-277 -- as a shift of all N + 1 headings of the walk by the same turns.
// `nb` is the batch of tick t + 1 (its y / yN hold the references sampled ahead), psi_rel their float64 headings.
__global__ void plant_ahead_kernel(PlantAhead a)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < a.B) plant_ahead_one(a, r);
}

// MpcWrapper::solve's reset (mpc_wrapper.cpp:267-275) from the plant's pose: one thread per (robot, node)
__global__ void iterate_reset_kernel(alore_nmpc_batch b, int B, int N, const double* pose, const unsigned char* mask)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)B * (N + 1)) return;
    const int r = (int)(t / (N + 1)), k = (int)(t % (N + 1));
    if (mask && !mask[r]) return;
    float* x = b.x + ((size_t)r * (N + 1) + k) * 3;
    x[0] = (float)pose[(size_t)r * 3]; x[1] = (float)pose[(size_t)r * 3 + 1]; x[2] = (float)pose[(size_t)r * 3 + 2];
    if (k < N) { b.u[((size_t)r * N + k) * 2] = 0.f; b.u[((size_t)r * N + k) * 2 + 1] = 0.f; }
}
hipError_t launch_iterate_reset(const alore_nmpc_batch& b, int B, int N, const double* pose, const unsigned char* mask, hipStream_t st)
{
    const long total = (long)B * (N + 1);
    hipLaunchKernelGGL(iterate_reset_kernel, dim3((unsigned)((total + 127) / 128)), dim3(128), 0, st, b, B, N, pose, mask);
    return hipGetLastError();
}

hipError_t launch_plant(const alore_nmpc_batch& b, int B, int N, int node, const double* icr, const int* at_goal,
                        double* pose, double* vw, const PlantParams& p, hipStream_t st)
{
    hipLaunchKernelGGL(plant_kernel, dim3((B + 127) / 128), dim3(128), 0, st, b, B, N, node, icr, at_goal, pose, vw, p);
    return hipGetLastError();
}

hipError_t launch_plant_ahead(const PlantAhead& a, hipStream_t st)
{
    hipLaunchKernelGGL(plant_ahead_kernel, dim3((a.B + 63) / 64), dim3(64), 0, st, a);
    return hipGetLastError();
}

bool ref_sample_ahead_supported(int N) { return N + 1 <= 64; }
hipError_t launch_ref_sample_ahead(const RefStore& s, const alore_nmpc_batch& b, int B, int N, double dt, double now, const double* icr,
                                   double* psi_rel, hipStream_t st)
{
    if (N + 1 <= 32)
        hipLaunchKernelGGL((ref_sample_smooth_kernel<32, true>), dim3((B + 3) / 4), dim3(128), 0, st, s, b, B, N, dt, now, nullptr, icr, nullptr, psi_rel);
    else
        hipLaunchKernelGGL((ref_sample_smooth_kernel<64, true>), dim3((B + 1) / 2), dim3(128), 0, st, s, b, B, N, dt, now, nullptr, icr, nullptr, psi_rel);
    return hipGetLastError();
}

hipError_t launch_ref_sample(const RefStore& s, const alore_nmpc_batch& b, int B, int N, double dt, double now,
                             const double* est, const double* icr, int* at_goal, double* psi_scratch, int do_smooth,
                             hipStream_t st)
{
    const long total = (long)B * (N + 1);
    const int threads = 128;
    const unsigned blocks = (unsigned)((total + threads - 1) / threads);
    if (do_smooth && N + 1 <= 64) { // sampling and smooth_yaw in one launch
        if (N + 1 <= 32) hipLaunchKernelGGL(ref_sample_smooth_kernel<32>, dim3((B + 3) / 4), dim3(128), 0, st, s, b, B, N, dt, now, est, icr, at_goal);
        else hipLaunchKernelGGL(ref_sample_smooth_kernel<64>, dim3((B + 1) / 2), dim3(128), 0, st, s, b, B, N, dt, now, est, icr, at_goal);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(ref_sample_kernel, dim3(blocks), dim3(threads), 0, st, s, b, B, N, dt, now, est, icr, at_goal,
                       do_smooth ? psi_scratch : nullptr);
    if (do_smooth) {
        hipLaunchKernelGGL(ref_unwrap_kernel, dim3((B + 63) / 64), dim3(64), 0, st, s, b, B, N, est, psi_scratch);
    }
    return hipGetLastError();
}

} // namespace nmpc
