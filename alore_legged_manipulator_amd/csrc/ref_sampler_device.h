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

} // namespace nmpc
#endif
