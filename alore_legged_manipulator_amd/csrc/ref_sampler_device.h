// ref_sampler_device.h -- device side of the reference sampling of one (robot, node) and of the sampling + smooth_yaw walk of
// a workgroup (ref_sampler.hip: the kernels; nmpc_block_kernel.hip: the same walk on the workgroups that follow the solver's in
// one grid, alore_nmpc_closed_loop_run).  Reference: see ref_sampler.hip.
#ifndef ALORE_REF_SAMPLER_DEVICE_H
#define ALORE_REF_SAMPLER_DEVICE_H
#include "nmpc_kernels.h"

#include "minco_spline.h"

namespace nmpc {

// node j of robot r: everything of getRefPoints / setTrajectory / setICRParameters for that node; returns false when the
// robot has no trajectory (nothing written), else the normalised heading in `psi_out` (also written to y / yN as float)
// AHEAD: the pose-independent part only, for a tick whose pose does not exist yet (closed_loop_run samples tick t + 1 beside the
// solve of tick t): no x0, no at-goal flag, no od (constant over a run: the first tick of the run wrote it)
template <bool AHEAD = false>
__device__ __forceinline__ bool ref_sample_node(const RefStore& s, const alore_nmpc_batch& b, int N, double dt, double now,
                                                const double* est, const double* icr, int* at_goal, int r, int j, double& psi_out)
{
    const double* m = s.meta + (size_t)r * 8;
    if (m[6] == 0.0) { // no trajectory yet: leave the references alone, and the robot is not at a goal
        if (j == 0 && at_goal) at_goal[r] = 0;
        return false;
    }
    const double start_time = m[0], duration = m[1], xv = m[2], res = m[3];
    const int np = (int)m[4], nc = (int)m[5];
    const double* dur = s.dur + (size_t)r * s.P;
    const double* coef = s.coef + (size_t)r * s.P * 12;
    const double t_cur = now - start_time;
    double temp_t = t_cur + dt;
    for (int i = 0; i < j; ++i) temp_t += dt; // the reference accumulates (mpc.cpp:432)
    const bool inside = temp_t <= duration;
    const double tq = inside ? temp_t : duration;
    // TrajAnal::getPstate
    int index = (int)floor(tq / res);
    if (index > nc - 1) index = nc - 1;
    if (index < 0) index = 0; // now before start_time: the reference indexes out of bounds here; extrapolate from node 0
    const double floor_t = index * res, diff_t = tq - floor_t;
    double p1[2], p2[2], p3[2], v1[2], v2[2], v3[2];
    constexpr int PRE = 16; // piece durations fetched up front (independent loads); longer trajectories walk on in memory
    double dreg[PRE];
#pragma unroll
    for (int i = 0; i < PRE; ++i) dreg[i] = dur[min(i, np - 1)];
    minco::eval_pv_pre<PRE>(dreg, dur, coef, np, floor_t, p1, v1);
    minco::eval_pv_pre<PRE>(dreg, dur, coef, np, floor_t + diff_t / 2.0, p2, v2);
    minco::eval_pv_pre<PRE>(dreg, dur, coef, np, tq, p3, v3);
    // minco::xdot / ydot of the three Simpson nodes with ONE sincos per node (they call cos and sin separately: twelve
    // float64 trigonometric evaluations per thread were most of this kernel)
    double xd1, yd1, xd2, yd2, xd3, yd3;
    {
        double sn, cs;
        sincos(p1[0], &sn, &cs); xd1 = v1[1] * cs + v1[0] * xv * sn; yd1 = v1[1] * sn - v1[0] * xv * cs;
        sincos(p2[0], &sn, &cs); xd2 = v2[1] * cs + v2[0] * xv * sn; yd2 = v2[1] * sn - v2[0] * xv * cs;
        sincos(p3[0], &sn, &cs); xd3 = v3[1] * cs + v3[0] * xv * sn; yd3 = v3[1] * sn - v3[0] * xv * cs;
    }
    const double* ck = s.ckpt + ((size_t)r * s.C + index) * 2;
    const double X = ck[0] + diff_t / 6.0 * (xd1 + 4.0 * xd2 + xd3);
    const double Y = ck[1] + diff_t / 6.0 * (yd1 + 4.0 * yd2 + yd3);
    double psi = p3[0];
    while (psi > M_PI) psi -= 2 * M_PI; // normlize_theta
    while (psi < -M_PI) psi += 2 * M_PI;
    psi_out = psi;
    const double yr = icr[(size_t)r * 3 + 1], yl = icr[(size_t)r * 3 + 2];
    const double vr = inside ? v3[1] - v3[0] * yr : 0.0, vl = inside ? v3[1] - v3[0] * yl : 0.0;
    if (j < N) {
        float* y = const_cast<float*>(b.y) + ((size_t)r * N + j) * 5;
        y[0] = (float)X; y[1] = (float)Y; y[2] = (float)psi; y[3] = (float)vr; y[4] = (float)vl;
    } else {
        float* yN = const_cast<float*>(b.yN) + (size_t)r * 3;
        yN[0] = (float)X; yN[1] = (float)Y; yN[2] = (float)psi;
    }
    if (AHEAD) return true;
    float* od = const_cast<float*>(b.od) + ((size_t)r * (N + 1) + j) * 3; // setICRParameters
    od[0] = (float)icr[(size_t)r * 3]; od[1] = (float)yr; od[2] = (float)yl;
    if (j == 0) {
        float* x0 = const_cast<float*>(b.x0) + (size_t)r * 3;
        x0[0] = (float)est[(size_t)r * 3]; x0[1] = (float)est[(size_t)r * 3 + 1]; x0[2] = (float)est[(size_t)r * 3 + 2];
        if (at_goal) at_goal[r] = (t_cur > duration + 1.0) ? 1 : 0;
    }
    return true;
}

// The same with smooth_yaw in the same launch: a robot's nodes sit on G consecutive lanes of one wavefront (G = 32 or 64
// >= N + 1), the sequential walk of the reference (node i is unwrapped against the already unwrapped node i - 1; node 0
// against the measured heading) hands the value from lane to lane by shuffles, all in float64 and cast afterwards like
// ref_unwrap_kernel below.  Replaces two launches and the round trip of the raw headings through memory.
// AHEAD (see ref_sample_node): node 0 keeps its normalised heading (the walk is the same whatever node 0 is shifted by: every
// later node follows its predecessor), the float64 headings of the walk go to psi_rel [B][N + 1]; plant_ahead_kernel shifts the
// 21 headings of a robot by the turns that node 0 is away from the pose it produces.
// `block`, `tid`, `threads`: the workgroup's index, the thread's index in it and its size, as the caller numbers them (a grid of its own,
// or the workgroups that follow the solver's in one grid: nmpc_block_kernel.hip)
template <int G, bool AHEAD = false>
__device__ __forceinline__ void ref_sample_smooth_body(const RefStore& s, const alore_nmpc_batch& b, int B, int N, double dt, double now, const double* est,
                                                       const double* icr, int* at_goal, double* psi_rel, int block, int tid, int threads)
{
    const int per_block = threads / G;
    const int r0 = block * per_block + tid / G, j = tid % G;
    const bool in_range = r0 < B && j <= N;
    const int r = r0 < B ? r0 : B - 1;
    double cur = 0.0;
    const bool have = in_range && ref_sample_node<AHEAD>(s, b, N, dt, now, est, icr, at_goal, r, j, cur);
    const double th = AHEAD ? cur : est[(size_t)r * 3 + 2];
    double prev = th;                          // what this node is unwrapped against (lane 0: the measured heading)
    for (int i = 0; i <= N; ++i) {             // wavefront-uniform
        if (j == i) {
            double dyaw = cur - prev;
            while (dyaw >= M_PI / 2) { cur -= M_PI * 2; dyaw = cur - prev; }
            while (dyaw <= -M_PI / 2) { cur += M_PI * 2; dyaw = cur - prev; }
        }
        // the value of the lane to the left (the only reader is lane i + 1 of the group): a wavefront shift on the DPP path
        // (wave_shr:1 crosses the rows of 16) instead of an LDS permute and its latency in each of the N + 1 dependent steps
        const int clo = __builtin_amdgcn_update_dpp(0, __double2loint(cur), 0x138, 0xF, 0xF, false);
        const int chi = __builtin_amdgcn_update_dpp(0, __double2hiint(cur), 0x138, 0xF, 0xF, false);
        if (j == i + 1) prev = __hiloint2double(chi, clo);
    }
    if (have) {
        if (j < N) const_cast<float*>(b.y)[((size_t)r * N + j) * 5 + 2] = (float)cur;
        else const_cast<float*>(b.yN)[(size_t)r * 3 + 2] = (float)cur;
        if (AHEAD) psi_rel[(size_t)r * (N + 1) + j] = cur;
    }
}

// The substeps of one control tick (StatePropaCallback, simulator.h:234-275).  The heading changes once per substep and its sine /
// cosine after the update are those the next substep starts with: one evaluation per tick, then a rotation by the substep's angle
// w * period -- a few milliradians, whose sine and cosine are short series (next terms d^9 / 9! and d^8 / 8!: below the last bit
// for |d| < 0.03); a larger angle takes the full evaluation.  The reference calls cos and sin four times per substep; the rotation
// differs from them by rounding (1e-16 per substep, nothing carried over to the next tick).
__device__ __forceinline__ void plant_substeps(const PlantParams& p, double desired_v, double desired_w, double vy, double& x, double& y, double& th,
                                               double& v, double& w)
{
    double sn, cs;
    sincos(th, &sn, &cs);
    for (int k = 0; k < p.substeps; ++k) {
        if (fabs(v - desired_v) >= p.pose_pub_period * p.max_a) v += p.pose_pub_period * p.max_a * (desired_v - v) / fabs(desired_v - v);
        else v = desired_v;
        if (fabs(w - desired_w) >= p.pose_pub_period * p.max_domega) w += p.pose_pub_period * p.max_domega * (desired_w - w) / fabs(desired_w - w);
        else w = desired_w;
        x += v * p.propa_period * cs;
        y += v * p.propa_period * sn;
        const double d = w * p.propa_period;
        th += d;
        if (fabs(d) < 0.03) {
            const double d2 = d * d;
            const double sd = d * (1.0 - d2 * (1.0 / 6.0) * (1.0 - d2 * (1.0 / 20.0) * (1.0 - d2 * (1.0 / 42.0))));
            const double cd = 1.0 - d2 * 0.5 * (1.0 - d2 * (1.0 / 12.0) * (1.0 - d2 * (1.0 / 30.0)));
            const double c1 = cs * cd - sn * sd, s1 = sn * cd + cs * sd;
            cs = c1; sn = s1;
        } else {
            sincos(th, &sn, &cs);
        }
        x -= vy * p.propa_period * sn;
        y += vy * p.propa_period * cs;
    }
}

// The plant step of tick t for robot r and what the sampler of tick t + 1 could not do without the pose it produces: the at-goal
// flag of tick t (getRefPoints' test, from `now` of tick t), the plant (simulator.h:234-275, see plant_kernel), then x0 <- pose
// and smooth_yaw's first step -- node 0 against the measured heading, mpc.cpp:248-277 -- as a shift of all N + 1 headings of the
// walk by the same turns.  y / yN are the references of tick t + 1 (sampled ahead), psi_rel their float64 headings (null: the
// run ends with tick t).
__device__ __forceinline__ void plant_ahead_one(const PlantAhead& a, int r)
{
    const int N = a.N;
    const double* m = a.meta + (size_t)r * 8;
    const bool valid = m[6] != 0.0;
    const int goal = (valid && (a.now - m[0]) > m[1] + 1.0) ? 1 : 0;
    double right = (double)a.u[((size_t)r * N + a.node) * 2], left = (double)a.u[((size_t)r * N + a.node) * 2 + 1];
    const double c0 = (valid && a.psi_rel) ? a.psi_rel[(size_t)r * (N + 1)] : 0.0;
    if (goal) { right = 0.0; left = 0.0; }
    const double xv = a.icr[(size_t)r * 3], yr = a.icr[(size_t)r * 3 + 1], yl = a.icr[(size_t)r * 3 + 2];
    const double desired_v = (left + right) / 2.0 - (right - left) / (yl - yr) * (yl + yr) / 2.0;
    const double vy = -(right - left) / (yl - yr) * xv;
    const double desired_w = (right - left) / (yl - yr);
    double x = a.pose[(size_t)r * 3], y = a.pose[(size_t)r * 3 + 1], th = a.pose[(size_t)r * 3 + 2];
    double v = a.vw[(size_t)r * 2], w = a.vw[(size_t)r * 2 + 1];
    plant_substeps(a.p, desired_v, desired_w, vy, x, y, th, v, w);
    a.pose[(size_t)r * 3] = x; a.pose[(size_t)r * 3 + 1] = y; a.pose[(size_t)r * 3 + 2] = th;
    a.vw[(size_t)r * 2] = v; a.vw[(size_t)r * 2 + 1] = w;
    a.at_goal[r] = goal;
    if (!valid || !a.psi_rel) return; // no trajectory: the references and x0 stay as they are, like in the sampler
    float* x0 = a.x0 + (size_t)r * 3;
    x0[0] = (float)x; x0[1] = (float)y; x0[2] = (float)th;
    int turns = 0; // net steps of 2 pi that smooth_yaw's two loops move node 0 by
    {
        double cur = c0, dyaw = cur - th;
        while (dyaw >= M_PI / 2) { cur -= M_PI * 2; dyaw = cur - th; --turns; }
        while (dyaw <= -M_PI / 2) { cur += M_PI * 2; dyaw = cur - th; ++turns; }
    }
    if (turns == 0) return;
    const double* pr = a.psi_rel + (size_t)r * (N + 1);
    float* yy = a.y + (size_t)r * N * 5;
    for (int j = 0; j <= N; ++j) {
        double cur = pr[j];
        for (int k = 0; k < (turns > 0 ? turns : -turns); ++k) cur += turns > 0 ? M_PI * 2 : -M_PI * 2;
        if (j < N) yy[(size_t)j * 5 + 2] = (float)cur;
        else a.yN[(size_t)r * 3 + 2] = (float)cur;
    }
}

} // namespace nmpc
#endif
