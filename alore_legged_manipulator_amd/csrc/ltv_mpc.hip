// ltv_mpc.hip -- batched linear time-varying MPC of the reference's `mpc` node (include/alore_ltv_mpc.h).
//
// Reference (P = /root/reference/planning_ddr_opt/mpc_controller/src/mpc.cpp):
//   stateTrans / predictMotion   P:233-269     rollout of the previous output (its clamping quirks kept)
//   getLinearModel               P:217-231     unicycle, explicit Euler, linearised about the rollout
//   solveMPCV                    P:304-535     the QP (states + inputs, dynamics rows, input box, rate limits) -> OSQP
//   getCmd                       P:569-614     relinearisation loop (wall-clock bound there, counted here), delay buffer
//   predictMotion(xopt)          P:271-302     linear prediction that is published
//   getRefPoints / smooth_yaw    P:634-690, 538-567
//
// Two kernels.  get_cmd_lanes_kernel (the one that runs): 16 lanes per robot, a lane owns two (four) consecutive
// stages, stage records in LDS, rollout / published prediction by prefix sums over the lanes, sweeps lane by lane --
// see its header below.  get_cmd_kernel (ALORE_LTV_KERNEL=thread, kept for A/B): one thread per robot, the per-stage
// records of all robots interleaved in a global workspace ([field][stage][robot]).  float64 like the reference.
//
// The QP is solved exactly by a working-set Riccati method.  State of stage j: xi = (x, y, theta, v_{j-1}, w_{j-1})
// -- augmenting by the previous input makes the rate penalty Rd and the rate limits stage-local.  Each input
// component is FREE, on a BOX bound or on a RATE limit.  For a given assignment the stage cost-to-go is a quadratic
// form in w = (xi, u0, u1); a component is eliminated by an affine substitution u = a.w_rest + f (the minimiser if
// free, the bound if boxed, previous input +- limit if rate-limited) -- one routine for all cases.  The forward
// sweep evaluates inputs, the multipliers of the non-free components (gradient rows kept from the backward
// sweep) and the violations of the free ones, and updates the assignment; done when it reproduces itself.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/alore_ltv_mpc.h"
#include "minco_spline.h"
#include "nmpc_kernels.h"

#include "ltv_mpc.h"

namespace ltv {

#define LTV_STAMP(i)                                                                         \
    if (d.stamps && b == 0) {                                                                \
        const long long now_ = (long long)__builtin_readcyclecounter();                      \
        d.stamps[i] += now_ - d.stamps[7];                                                   \
        d.stamps[7] = now_;                                                                  \
    }
constexpr int SINGLE_AFTER = 24; // from this sweep on only the most severe change is applied (breaks cycles)

struct Quad7 { // symmetric 7 x 7 form H and vector h in w = (xi0..4, u0, u1); only the entries that can be non-zero are kept dense
    double H[7][7];
    double h[7];
};

__device__ __forceinline__ double& W(const Dev& d, int j, int f, int b) { return d.ws[((size_t)j * NF + f) * d.stride + b]; }

// 1 if the float64 at p is NaN or Inf, by its exponent bits (this file is built without NaN / Inf semantics: no floating-point
// test would survive).  A plain 32-bit load of the high word; the empty asm keeps the optimiser from reasoning about the value.
__device__ __forceinline__ unsigned nf_bits(const double* p)
{
    unsigned hi = reinterpret_cast<const unsigned*>(p)[1];
    asm volatile("" : "+v"(hi));
    return ((hi >> 20) & 0x7ffu) == 0x7ffu ? 1u : 0u;
}

__global__ __launch_bounds__(64) void get_cmd_kernel(Dev d)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= d.B) return;
    const alore_ltv_config& c = d.c;
    const int T = c.predict_steps, dl = c.delay_num, K = T - dl;
    const double dt = c.dt;
    const double umax[2] = {c.max_vel, c.max_omega}, rmax[2] = {c.max_acc * dt, c.max_domega * dt};
    const double Qp2[3] = {2.0 * c.matrix_q[0], 2.0 * c.matrix_q[1], 2.0 * c.matrix_q[3]};
    const double Rd2[2] = {2.0 * c.matrix_rd[0], 2.0 * c.matrix_rd[1]};
    const double Ruu0[2] = {2.0 * (c.matrix_r[0] + c.matrix_q[2]), 2.0 * c.matrix_r[1]};
    const double tol = 1e-9;
    double* out = d.output + (size_t)b * T * 2;
    double* bf = d.buff + (size_t)b * (dl > 0 ? dl : 1) * 2;
    const double* xr = d.xref + (size_t)b * T * 3;
    const double* dr = d.dref + (size_t)b * T * 2;
    if (d.reset) {
        for (int i = 0; i < 2 * T; ++i) out[i] = 0.0;
        for (int i = 0; i < 2 * dl; ++i) bf[i] = 0.0;
        for (int j = 0; j < T; ++j) { d.st[((size_t)j * 2) * d.stride + b] = FREE; d.st[((size_t)j * 2 + 1) * d.stride + b] = FREE; }
    }
    const double x0 = d.now[(size_t)b * 3], y0 = d.now[(size_t)b * 3 + 1], th0 = d.now[(size_t)b * 3 + 2];
    int sweeps = 0, status = 0;
    { // this file is built without NaN / Inf semantics: a robot with a non-finite input is not solved at all (bit tests)
        unsigned bad = 0u;
        for (int i = 0; i < 3; ++i) bad |= nf_bits(d.now + (size_t)b * 3 + i);
        for (int i = 0; i < 3 * T; ++i) bad |= nf_bits(xr + i);
        for (int i = 0; i < 2 * T; ++i) bad |= nf_bits(dr + i) | nf_bits(out + i);
        for (int i = 0; i < 2 * dl; ++i) bad |= nf_bits(bf + i);
        if (bad) {
            d.sweeps[b] = 0;
            d.status[b] = STATUS_NON_FINITE;
            d.cmd[2 * b] = 0.0; d.cmd[2 * b + 1] = 0.0;
            if (d.cmd_host) { d.cmd_host[2 * b] = 0.0; d.cmd_host[2 * b + 1] = 0.0; d.status_host[b] = STATUS_NON_FINITE; }
            return;
        }
    }

    if (d.stamps && b == 0) d.stamps[7] = (long long)__builtin_readcyclecounter();
    for (int relin = 0; relin < d.n_relin; ++relin) {
        // ---- predictMotion: rollout of the last output; the linear model about xbar[dl + j] goes to the records
        double px = x0, py = y0, pth = th0, pv = 0.0; // now_state.v = 0 (odometry callback)
        double pos0[3] = {x0, y0, th0};
        // the previous output is read eight steps at a time (16 loads in flight; a load issued behind the record stores of
        // the step before would wait out its own round trip), one sincos per step serves the model and the rollout
        for (int i0 = 0; i0 <= T; i0 += 8) {
            double oa[8], oy[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int i = i0 + q;
                oa[q] = i < T ? out[2 * i] : 0.0;
                oy[q] = i < T ? out[2 * i + 1] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int i = i0 + q;
                if (i > T) break;
                double sn, cs;
                sincos(pth, &sn, &cs);
                if (i >= dl && i < T) {
                    const int j = i - dl;
                    const double B00 = cs * dt, B10 = sn * dt;
                    const double A02 = -B10 * pv, A12 = B00 * pv;
                    W(d, j, 0, b) = A02; W(d, j, 1, b) = A12; W(d, j, 2, b) = B00; W(d, j, 3, b) = B10;
                    W(d, j, 4, b) = -A02 * pth; W(d, j, 5, b) = -A12 * pth;
                    if (j == 0) { pos0[0] = px; pos0[1] = py; pos0[2] = pth; }
                }
                if (i == T) break;
                // stateTrans(temp, a = output(0, i), yaw_dot = output(1, i))
                const double a = oa[q];
                const double yd = fmin(fmax(oy[q], -c.max_omega), c.max_omega);
                px += a * cs * dt;
                py += a * sn * dt;
                pth += yd * dt;
                pv = a;
            }
        }
        LTV_STAMP(0)
        // ---- working-set iterations
        bool settled = false;
        for (sweeps = 0; sweeps < c.max_sweeps && !settled; ++sweeps) {
            // backward sweep: cost-to-go V(xi) = 1/2 xi' P xi + p' xi, P symmetric 5 x 5
            double P[5][5], p[5];
#pragma unroll
            for (int r = 0; r < 5; ++r) { p[r] = 0.0;
#pragma unroll
                for (int s = 0; s < 5; ++s) P[r][s] = 0.0; }
            // effective boxes: a stage that sits on a rate limit hands its box on to its predecessor, shifted by the limit
            // (u_j = u_{j-1} + r and u_j <= hi  =>  u_{j-1} <= hi - r); a BOX status means "on the effective box"
            double lo_eff[2] = {-umax[0], -umax[1]}, hi_eff[2] = {umax[0], umax[1]};
            for (int j = K - 1; j >= 0; --j) {
                const double A02 = W(d, j, 0, b), A12 = W(d, j, 1, b), B00 = W(d, j, 2, b), B10 = W(d, j, 3, b);
                const double C0 = W(d, j, 4, b), C1 = W(d, j, 5, b);
                // every load of the stage before its first store (a load behind a possibly aliasing store waits for its own
                // round trip): references, working-set status
                const double xr0 = xr[(dl + j) * 3], xr1 = xr[(dl + j) * 3 + 1], xr2 = xr[(dl + j) * 3 + 2], dr0 = dr[(dl + j) * 2];
                const int st1 = d.st[((size_t)j * 2 + 1) * d.stride + b], st0 = d.st[((size_t)j * 2) * d.stride + b];
                W(d, j, 34, b) = lo_eff[0]; W(d, j, 35, b) = hi_eff[0]; W(d, j, 36, b) = lo_eff[1]; W(d, j, 37, b) = hi_eff[1];
                // tracking on the next position, then pull back through xi' = Fw w + c:
                //   x' = x + A02 th + B00 u0 + C0 ; y' = y + A12 th + B10 u0 + C1 ; th' = th + dt u1 ; v' = u0 ; om' = u1
#pragma unroll
                for (int r = 0; r < 3; ++r) { P[r][r] += Qp2[r]; p[r] -= Qp2[r] * (r == 0 ? xr0 : (r == 1 ? xr1 : xr2)); }
                // s = P c + p
                double s[5];
#pragma unroll
                for (int r = 0; r < 5; ++r) s[r] = P[r][0] * C0 + P[r][1] * C1 + p[r];
                // Pull-back H = Fw' P Fw, h = Fw' s through the columns of Fw (5 x 7):
                //   c0 = e0, c1 = e1, c2 = (A02, A12, 1, 0, 0), c3 = c4 = 0, c5 = (B00, B10, 0, 1, 0), c6 = (0, 0, dt, 0, 1)
                // written out term by term (the columns are mostly unit vectors: 50 products instead of the 420 of the
                // dense triple loop, whose multiplications by literal zeros IEEE arithmetic does not let the compiler drop)
                Quad7 q;
                double PF[7][5]; // P c_k
#pragma unroll
                for (int r = 0; r < 5; ++r) {
                    PF[0][r] = P[r][0];
                    PF[1][r] = P[r][1];
                    PF[2][r] = A02 * P[r][0] + A12 * P[r][1] + P[r][2];
                    PF[3][r] = 0.0;
                    PF[4][r] = 0.0;
                    PF[5][r] = B00 * P[r][0] + B10 * P[r][1] + P[r][3];
                    PF[6][r] = dt * P[r][2] + P[r][4];
                }
#pragma unroll
                for (int k = 0; k < 7; ++k) { // row m of H: c_m' (P c_k)
                    q.H[0][k] = PF[k][0];
                    q.H[1][k] = PF[k][1];
                    q.H[2][k] = A02 * PF[k][0] + A12 * PF[k][1] + PF[k][2];
                    q.H[3][k] = 0.0;
                    q.H[4][k] = 0.0;
                    q.H[5][k] = B00 * PF[k][0] + B10 * PF[k][1] + PF[k][3];
                    q.H[6][k] = dt * PF[k][2] + PF[k][4];
                }
                q.h[0] = s[0];
                q.h[1] = s[1];
                q.h[2] = A02 * s[0] + A12 * s[1] + s[2];
                q.h[3] = 0.0;
                q.h[4] = 0.0;
                q.h[5] = B00 * s[0] + B10 * s[1] + s[3];
                q.h[6] = dt * s[2] + s[4];
                q.H[5][5] += Ruu0[0]; q.H[6][6] += Ruu0[1];
                q.h[5] += -2.0 * c.matrix_q[2] * dr0;
                if (j >= 1) {
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc) {
                        q.H[5 + cc][5 + cc] += Rd2[cc]; q.H[3 + cc][3 + cc] += Rd2[cc];
                        q.H[5 + cc][3 + cc] -= Rd2[cc]; q.H[3 + cc][5 + cc] -= Rd2[cc];
                    }
                }
                // ---- eliminate u1 (index 6): u1 = a1 . w[0..5] + f1
                {
                    double a1[6], f1;
                    const double inv = 1.0 / q.H[6][6];
                    const bool fr = st1 == FREE, bx = (st1 == BOX_LO || st1 == BOX_HI);
#pragma unroll
                    for (int k = 0; k < 6; ++k) a1[k] = fr ? -q.H[6][k] * inv : ((!bx && k == 4) ? 1.0 : 0.0);
                    f1 = fr ? -q.h[6] * inv : (bx ? (st1 == BOX_LO ? lo_eff[1] : hi_eff[1]) : (st1 == RATE_LO ? -rmax[1] : rmax[1]));
#pragma unroll
                    for (int k = 0; k < 6; ++k) W(d, j, 6 + k, b) = a1[k];
                    W(d, j, 12, b) = f1;
#pragma unroll
                    for (int k = 0; k < 7; ++k) W(d, j, 13 + k, b) = q.H[6][k]; // gradient row of u1
                    W(d, j, 20, b) = q.h[6];
                    const double Hkk = q.H[6][6], hk = q.h[6];
                    double Hk[6];
#pragma unroll
                    for (int k = 0; k < 6; ++k) Hk[k] = q.H[6][k];
#pragma unroll
                    for (int r = 0; r < 6; ++r) {
                        q.h[r] += a1[r] * hk + (Hk[r] + a1[r] * Hkk) * f1;
#pragma unroll
                        for (int s2 = 0; s2 < 6; ++s2) q.H[r][s2] += a1[r] * Hk[s2] + Hk[r] * a1[s2] + Hkk * a1[r] * a1[s2];
                    }
                }
                // ---- eliminate u0 (index 5): u0 = a0 . xi + f0
                {
                    double a0[5], f0;
                    const double inv = 1.0 / q.H[5][5];
                    const bool fr = st0 == FREE, bx = (st0 == BOX_LO || st0 == BOX_HI);
#pragma unroll
                    for (int k = 0; k < 5; ++k) a0[k] = fr ? -q.H[5][k] * inv : ((!bx && k == 3) ? 1.0 : 0.0);
                    f0 = fr ? -q.h[5] * inv : (bx ? (st0 == BOX_LO ? lo_eff[0] : hi_eff[0]) : (st0 == RATE_LO ? -rmax[0] : rmax[0]));
#pragma unroll
                    for (int k = 0; k < 5; ++k) W(d, j, 21 + k, b) = a0[k];
                    W(d, j, 26, b) = f0;
#pragma unroll
                    for (int k = 0; k < 6; ++k) W(d, j, 27 + k, b) = q.H[5][k]; // reduced gradient row of u0
                    W(d, j, 33, b) = q.h[5];
                    const double Hkk = q.H[5][5], hk = q.h[5];
                    double Hk[5];
#pragma unroll
                    for (int k = 0; k < 5; ++k) Hk[k] = q.H[5][k];
#pragma unroll
                    for (int r = 0; r < 5; ++r) {
                        p[r] = q.h[r] + a0[r] * hk + (Hk[r] + a0[r] * Hkk) * f0;
#pragma unroll
                        for (int s2 = 0; s2 < 5; ++s2) P[r][s2] = q.H[r][s2] + a0[r] * Hk[s2] + Hk[r] * a0[s2] + Hkk * a0[r] * a0[s2];
                    }
                }
                // boxes of stage j - 1
                {
                    const double l0 = lo_eff[0], h0 = hi_eff[0], l1 = lo_eff[1], h1 = hi_eff[1];
                    lo_eff[0] = (st0 == RATE_LO) ? fmax(-umax[0], l0 + rmax[0]) : -umax[0];
                    hi_eff[0] = (st0 == RATE_HI) ? fmin(umax[0], h0 - rmax[0]) : umax[0];
                    lo_eff[1] = (st1 == RATE_LO) ? fmax(-umax[1], l1 + rmax[1]) : -umax[1];
                    hi_eff[1] = (st1 == RATE_HI) ? fmin(umax[1], h1 - rmax[1]) : umax[1];
                }
            }
            LTV_STAMP(1)
            // forward sweep: inputs, multipliers, violations -> next working set
            double xi[5] = {pos0[0], pos0[1], pos0[2], 0.0, 0.0};
            int changes = 0;
            // multiplier carried down a chain of rate-limited stages that hangs from a tightened box (the box of the
            // chain's last stage, active through the chain): gradient of the anchor; chain_dir: -1 lower, +1 upper
            double chain_mu[2] = {0.0, 0.0};
            int chain_dir[2] = {0, 0};
            const bool single = sweeps >= SINGLE_AFTER;
            double best_sev = -1.0;
            int best_j = 0, best_c = 0, best_ns = 0;
            for (int j = 0; j < K; ++j) {
                // the whole stage record in one burst (38 independent loads in flight): read one by one between the
                // stores to out[] / st[] below, every load would wait out its own round trip to L2
                double R[NF];
#pragma unroll
                for (int f = 0; f < NF; ++f) R[f] = W(d, j, f, b);
                const int st_j[2] = {d.st[((size_t)j * 2) * d.stride + b], d.st[((size_t)j * 2 + 1) * d.stride + b]};
                double w7[7];
#pragma unroll
                for (int k = 0; k < 5; ++k) w7[k] = xi[k];
                double u0 = R[26];
#pragma unroll
                for (int k = 0; k < 5; ++k) u0 += R[21 + k] * xi[k];
                w7[5] = u0;
                double u1 = R[12];
#pragma unroll
                for (int k = 0; k < 6; ++k) u1 += R[6 + k] * w7[k];
                w7[6] = u1;
                double g1 = R[20], g0 = R[33];
#pragma unroll
                for (int k = 0; k < 7; ++k) g1 += R[13 + k] * w7[k];
#pragma unroll
                for (int k = 0; k < 6; ++k) g0 += R[27 + k] * w7[k];
                const double uu[2] = {u0, u1}, gg[2] = {g0, g1};
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    int* sp = d.st + ((size_t)j * 2 + cc) * d.stride + b;
                    const int s = st_j[cc];
                    const double val = uu[cc], prev = xi[3 + cc], grad = gg[cc];
                    const double lo = R[34 + 2 * cc], hi = R[35 + 2 * cc];
                    int ns = s;
                    double sev = 0.0;
                    if (s == FREE) {
                        chain_mu[cc] = 0.0; chain_dir[cc] = 0;
                        const double vb = fmax(lo - val, val - hi);
                        const double vr = (j >= 1) ? fmax(-rmax[cc] - (val - prev), (val - prev) - rmax[cc]) : -1.0;
                        if (vb > tol && vb >= vr) { ns = (val < lo) ? BOX_LO : BOX_HI; sev = vb; }
                        else if (vr > tol) { ns = (val - prev < 0.0) ? RATE_LO : RATE_HI; sev = vr; }
                    } else if (s == BOX_LO || s == BOX_HI) {
                        const bool lower = (s == BOX_LO);
                        const bool tightened = lower ? (lo > -umax[cc] + 1e-12) : (hi < umax[cc] - 1e-12);
                        chain_mu[cc] = tightened ? grad : 0.0;
                        chain_dir[cc] = tightened ? (lower ? -1 : 1) : 0;
                        if ((lower && grad < -tol) || (!lower && grad > tol)) { ns = FREE; sev = fabs(grad); }
                        else if (j >= 1 && fabs(val - prev) > rmax[cc] + tol) { ns = (val - prev < 0.0) ? RATE_LO : RATE_HI; sev = fabs(val - prev) - rmax[cc]; }
                    } else {
                        const bool lower = (s == RATE_LO);
                        if (chain_dir[cc] != (lower ? -1 : 1)) { chain_mu[cc] = 0.0; chain_dir[cc] = 0; }
                        const double g_eff = grad - chain_mu[cc];
                        if ((lower && g_eff < -tol) || (!lower && g_eff > tol)) {
                            // inside a chain that hangs from a box further down: this stage becomes the anchor of the rest
                            ns = (chain_mu[cc] == 0.0) ? FREE : (lower ? BOX_LO : BOX_HI);
                            sev = fabs(g_eff);
                            chain_mu[cc] = 0.0; chain_dir[cc] = 0;
                        }
                    }
                    if (ns != s) {
                        ++changes;
                        if (!single) *sp = ns;
                        else if (sev > best_sev) { best_sev = sev; best_j = j; best_c = cc; best_ns = ns; }
                    }
                }
                out[2 * (dl + j)] = u0;
                out[2 * (dl + j) + 1] = u1;
                const double A02 = R[0], A12 = R[1], B00 = R[2], B10 = R[3];
                const double nx = xi[0] + A02 * xi[2] + B00 * u0 + R[4], ny = xi[1] + A12 * xi[2] + B10 * u0 + R[5];
                xi[2] = xi[2] + dt * u1; xi[0] = nx; xi[1] = ny; xi[3] = u0; xi[4] = u1;
            }
            if (single && changes > 0) d.st[((size_t)best_j * 2 + best_c) * d.stride + b] = best_ns;
            settled = (changes == 0);
            LTV_STAMP(2)
        }
        status = settled ? 0 : 1;
        for (int i = 0; i < dl; ++i) { out[2 * i] = bf[2 * i]; out[2 * i + 1] = bf[2 * i + 1]; } // solveMPCV: the delayed inputs
    }
    // ---- predictMotion(xopt): the linear prediction about the last rollout (what the node publishes as cmd_path)
    {
        double* xo = d.xopt + (size_t)b * (T + 1) * 3;
        // the rollout xbar of the last pass is not stored: redo stateTrans to get xbar[i-1] for the linear model
        double bth = th0, bv = 0.0; // heading and speed of xbar[i - 1] (its position does not enter the linear model)
        double tx = x0, ty = y0, tth = th0;          // temp (linear prediction)
        xo[0] = tx; xo[1] = ty; xo[2] = tth;
        // NOTE: the reference's xbar at this point is the rollout of the output BEFORE the last solve (predictMotion
        // runs at the top of each pass); `prev_out` is not kept, so the published path uses the final output for
        // both roles -- identical once the relinearisation has converged.
        for (int i = 1; i <= T; ++i) {
            double sb_, cb_;
            sincos(bth, &sb_, &cb_);
            const double B00 = cb_ * dt, B10 = sb_ * dt, A02 = -B10 * bv, A12 = B00 * bv;
            const double u0 = out[2 * (i - 1)], u1 = out[2 * (i - 1) + 1];
            const double nx = tx + A02 * tth + B00 * u0 - A02 * bth, ny = ty + A12 * tth + B10 * u0 - A12 * bth;
            tth = tth + dt * u1; tx = nx; ty = ny;
            xo[3 * i] = tx; xo[3 * i + 1] = ty; xo[3 * i + 2] = tth;
            double yd = fmin(fmax(u1, -c.max_omega), c.max_omega);
            bth += yd * dt; bv = u0;
        }
    }
    if (dl > 0) { // output_buff: drop the oldest, append the command just computed
        for (int i = 0; i + 1 < dl; ++i) { bf[2 * i] = bf[2 * (i + 1)]; bf[2 * i + 1] = bf[2 * (i + 1) + 1]; }
        bf[2 * (dl - 1)] = out[2 * dl]; bf[2 * (dl - 1) + 1] = out[2 * dl + 1];
    }
    LTV_STAMP(3)
    d.sweeps[b] = sweeps;
    d.status[b] = status;
    d.cmd[2 * b] = out[2 * dl]; d.cmd[2 * b + 1] = out[2 * dl + 1];
    if (d.cmd_host) { d.cmd_host[2 * b] = out[2 * dl]; d.cmd_host[2 * b + 1] = out[2 * dl + 1]; d.status_host[b] = status; }
}

// ---------------------------------------------------------------------------------------------------------------
// L = 16 lanes per robot, every lane owns S consecutive stages with their records (38 doubles each) in REGISTERS.
//   * rollout + linear model: the heading is a prefix sum of the clamped yaw rates, the position a prefix sum of
//     a cos / sin(heading) dt -- serial inside the lane's block, log2(L) DPP steps across the group; one sincos per
//     stage, all stages in parallel (the one-thread-per-robot kernel above walks the T steps one after the other and
//     keeps the records in a global workspace: 143 k + 133 k + 121 k cycles per relinearisation, most of them
//     memory round trips);
//   * backward / forward sweeps: the same stage algebra as above, lane by lane; the cost-to-go (20 doubles + the
//     effective boxes) resp. the augmented state and the chain multipliers travel to the next lane by DPP row shifts;
//   * the published linear prediction (predictMotion(xopt)): prefix sums again.
// 4096 robots are 1024 wavefronts (one per SIMD) instead of 64.  Numerics: the sums of the rollout associate
// differently (1e-16 relative); everything else is the arithmetic of the kernel above.
template <int CTRL>
__device__ __forceinline__ double dppd(double old, double x)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(x), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(x), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ int dppi(int old, int x) { return __builtin_amdgcn_update_dpp(old, x, CTRL, 0xF, 0xF, false); }
__device__ __forceinline__ double next_d(double x) { return dppd<0x101>(x, x); } // lane i <- lane i + 1 (row of 16)
__device__ __forceinline__ double prev_d(double x) { return dppd<0x111>(x, x); } // lane i <- lane i - 1
__device__ __forceinline__ int prev_i(int x) { return dppi<0x111>(x, x); }
// inclusive prefix sum over the 16 lanes of a DPP row
__device__ __forceinline__ double rowprefix_d(double x)
{
    x += dppd<0x111>(0.0, x);
    x += dppd<0x112>(0.0, x);
    x += dppd<0x114>(0.0, x);
    x += dppd<0x118>(0.0, x);
    return x;
}

// reciprocal of a positive pivot to float64 accuracy: hardware estimate + two Newton steps (the IEEE division sequence is
// three times as long and sits on the sweep's dependent chain twice per stage)
__device__ __forceinline__ double pivot_rcp_d(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

#define LSTAMP(i)                                                                            \
    if (d.stamps && blockIdx.x == 0 && lane == 0) {                                          \
        const long long now_ = (long long)__builtin_readcyclecounter();                      \
        d.stamps[i] += now_ - d.stamps[7];                                                   \
        d.stamps[7] = now_;                                                                  \
    }
template <int S>
__global__ __launch_bounds__(64) void get_cmd_lanes_kernel(Dev d)
{
    constexpr int L = 16, G = 4;
    (void)L;
    extern __shared__ double lds_rec[]; // stage records: [stage][field][robot of the wavefront], stage stride REC_STRIDE
    const int lane = threadIdx.x, j = lane & 15, g = lane >> 4;
    int b = blockIdx.x * G + g;
    const bool valid = b < d.B;
    if (!valid) b = d.B - 1; // padding groups shadow the last robot, never store
    const alore_ltv_config& c = d.c;
    const int T = c.predict_steps, dl = c.delay_num, K = T - dl;
    const int top = (K - 1) / S; // lane that owns the last stage
    const double dt = c.dt;
    const double umax[2] = {c.max_vel, c.max_omega}, rmax[2] = {c.max_acc * dt, c.max_domega * dt};
    const double Qp2[3] = {2.0 * c.matrix_q[0], 2.0 * c.matrix_q[1], 2.0 * c.matrix_q[3]};
    const double Rd2[2] = {2.0 * c.matrix_rd[0], 2.0 * c.matrix_rd[1]};
    const double Ruu0[2] = {2.0 * (c.matrix_r[0] + c.matrix_q[2]), 2.0 * c.matrix_r[1]};
    const double tol = 1e-9;
    double* out = d.output + (size_t)b * T * 2;
    double* bf = d.buff + (size_t)b * (dl > 0 ? dl : 1) * 2;
    const double* xrg = d.xref + (size_t)b * T * 3;
    const double* drg = d.dref + (size_t)b * T * 2;
    // the measured state: read once from wherever the host put it (device memory, or the pinned slab itself on the tick
    // path: one trip over the bus) into LDS behind the records, re-read from there where needed (no registers held)
    double* nowp = lds_rec + (size_t)K * REC_STRIDE + g * 3;
    if (j < 3) nowp[j] = d.now[(size_t)b * 3 + j];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // the lane's stages: k = j S + s (valid while k < K); output column of stage k is dl + k
    int st[S][2];
    double ua[S], uw[S];           // output of the lane's stages (acceleration-like input a, yaw rate): in/out
    double xr[S][3], dr0[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int k = j * S + s, kc = min(k, K - 1);
        const bool vs = k < K;
        ua[s] = (vs && !d.reset) ? out[2 * (dl + kc)] : 0.0;
        uw[s] = (vs && !d.reset) ? out[2 * (dl + kc) + 1] : 0.0;
        st[s][0] = (vs && !d.reset) ? d.st[((size_t)kc * 2) * d.stride + b] : FREE;
        st[s][1] = (vs && !d.reset) ? d.st[((size_t)kc * 2 + 1) * d.stride + b] : FREE;
        xr[s][0] = xrg[(dl + kc) * 3]; xr[s][1] = xrg[(dl + kc) * 3 + 1]; xr[s][2] = xrg[(dl + kc) * 3 + 2];
        dr0[s] = drg[(dl + kc) * 2];
    }
    // This file is built without NaN / Inf semantics.  A robot with a non-finite measured state, reference or stored output
    // (bit tests) is not solved: its inputs are replaced by zeros so that the arithmetic of its 16 lanes stays finite next to
    // its wavefront mates, it gets status 2 and a zero command, and its stored output / working set / delay buffer stay.
    bool poisoned;
    {
        // (plain loads of the high words, all in flight together, made opaque to the optimiser: a volatile load per value is a
        // system-scope access of its own, and behind short-circuit ORs thirteen of them ran one after the other -- ~20 us)
        unsigned bad = (j < 3) ? nf_bits(d.now + (size_t)b * 3 + j) : 0u;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int kc = min(j * S + s, K - 1);
            bad |= nf_bits(xrg + (dl + kc) * 3) | nf_bits(xrg + (dl + kc) * 3 + 1) | nf_bits(xrg + (dl + kc) * 3 + 2) | nf_bits(drg + (dl + kc) * 2);
            if (!d.reset) bad |= nf_bits(out + 2 * (dl + kc)) | nf_bits(out + 2 * (dl + kc) + 1);
        }
        if (!d.reset)
            for (int i = j; i < 2 * dl; i += 16) bad |= nf_bits(out + i) | nf_bits(bf + i);
        poisoned = ((__ballot(bad != 0u) >> (lane & ~15)) & 0xffffull) != 0ull;
        if (poisoned) {
#pragma unroll
            for (int s = 0; s < S; ++s) { ua[s] = uw[s] = 0.0; xr[s][0] = xr[s][1] = xr[s][2] = 0.0; dr0[s] = 0.0; st[s][0] = st[s][1] = FREE; }
            if (j < 3) nowp[j] = 0.0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    const int zero_hist = (d.reset || poisoned) ? 1 : 0; // the delayed inputs read as zeros
    // record field f of stage k of this lane's robot
    auto REC = [&](int k_, int f) -> double& { return lds_rec[(size_t)k_ * REC_STRIDE + f * 4 + g]; };
    int sweeps = 0, status = 0;
    double pos0[3] = {0.0, 0.0, 0.0};
    if (d.stamps && blockIdx.x == 0 && lane == 0) d.stamps[7] = (long long)__builtin_readcyclecounter();

    for (int relin = 0; relin < d.n_relin; ++relin) {
        // ---- predictMotion.  The first dl columns of the output are the delayed inputs (every lane walks them: dl is 1
        //      in the reference's configuration); columns dl + k belong to the stages.
        const double x0 = nowp[0], y0 = nowp[1], th0 = nowp[2];
        double px = x0, py = y0, pth = th0, pv = 0.0; // now_state.v = 0 (odometry callback)
        for (int i = 0; i < dl; ++i) {
            const double* src = (relin == 0) ? out : bf; // solveMPCV copies the delay buffer into the first columns
            const double a = zero_hist ? 0.0 : src[2 * i];
            const double yd = fmin(fmax(zero_hist ? 0.0 : src[2 * i + 1], -c.max_omega), c.max_omega);
            double sn, cs;
            sincos(pth, &sn, &cs);
            px += a * cs * dt; py += a * sn * dt; pth += yd * dt; pv = a;
        }
        pos0[0] = px; pos0[1] = py; pos0[2] = pth;
        {
            // heading entering stage k: exclusive prefix of the clamped yaw rates
            double inc[S], acc = 0.0, hd[S];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const bool vs = j * S + s < K;
                inc[s] = vs ? fmin(fmax(uw[s], -c.max_omega), c.max_omega) * dt : 0.0;
                hd[s] = acc; acc += inc[s];
            }
            const double ex = rowprefix_d(acc) - acc + pth;
            double sn[S], cs[S];
            // speed entering stage k: the input of the stage before (the last delayed input for k = 0)
            const double a_prev_lane = prev_d(ua[S - 1]);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int k = j * S + s;
                const bool vs = k < K;
                hd[s] += ex;
                sincos(hd[s], &sn[s], &cs[s]);
                const double v_in = (k == 0) ? pv : ((s == 0) ? a_prev_lane : ua[(s > 0) ? s - 1 : 0]);
                const double B00 = cs[s] * dt, B10 = sn[s] * dt;
                const double A02 = -B10 * v_in, A12 = B00 * v_in;
                if (vs) {
                    REC(k, 0) = A02; REC(k, 1) = A12; REC(k, 2) = B00; REC(k, 3) = B10; REC(k, 4) = -A02 * hd[s]; REC(k, 5) = -A12 * hd[s];
                }
            }
        }
        LSTAMP(0)
        // ---- working-set iterations
        bool settled = false;
        sweeps = 0;
        while (__any(!settled && sweeps < c.max_sweeps)) {
            const bool act = !settled && sweeps < c.max_sweeps;
            // backward sweep: cost-to-go V(xi) = 1/2 xi' P xi + p' xi, P symmetric 5 x 5, handed from lane to lane
            double P[5][5], p[5];
#pragma unroll
            for (int r = 0; r < 5; ++r) { p[r] = 0.0;
#pragma unroll
                for (int s2 = 0; s2 < 5; ++s2) P[r][s2] = 0.0; }
            double lo_eff[2] = {-umax[0], -umax[1]}, hi_eff[2] = {umax[0], umax[1]};
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            double lin[S][6]; // A02 A12 B00 B10 C0 C1 of the lane's stages: fixed during the sweeps of this relinearisation
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int kc = min(j * S + s, K - 1);
#pragma unroll
                for (int f = 0; f < 6; ++f) lin[s][f] = REC(kc, f);
            }
            for (int t = top; t >= 0; --t) {
                // Every lane runs the stage algebra on its own slot with whatever cost-to-go it holds; only lane t holds
                // the real one and only it stores the record.  (No heavy code under a partial EXEC mask: values that are
                // live across such a region in the lanes that sit it out did not survive the register allocator's
                // spilling there.)
                {
#pragma unroll
                    for (int s = S - 1; s >= 0; --s) {
                        const int k = j * S + s;
                        if (t * S + s < K) { // wavefront-uniform: the stage of lane t exists
                            double W_[NF];
                            const double A02 = lin[s][0], A12 = lin[s][1], B00 = lin[s][2], B10 = lin[s][3], C0 = lin[s][4], C1 = lin[s][5];
                            const int st1 = st[s][1], st0 = st[s][0];
                            W_[34] = lo_eff[0]; W_[35] = hi_eff[0]; W_[36] = lo_eff[1]; W_[37] = hi_eff[1];
#pragma unroll
                            for (int r = 0; r < 3; ++r) { P[r][r] += Qp2[r]; p[r] -= Qp2[r] * xr[s][r]; }
                            double sv[5];
#pragma unroll
                            for (int r = 0; r < 5; ++r) sv[r] = P[r][0] * C0 + P[r][1] * C1 + p[r];
                            // Pull-back H = Fw' P Fw, h = Fw' s (columns of Fw: c0 = e0, c1 = e1, c2 = (A02, A12, 1, 0, 0),
                            // c3 = c4 = 0, c5 = (B00, B10, 0, 1, 0), c6 = (0, 0, dt, 0, 1)) and the two eliminations, on the UPPER
                            // TRIANGLES only (P is symmetric; the one-thread kernel above carries all entries): 11 + 21 + 15
                            // two-term updates instead of 49 + 36 + 25 four-term ones
                            Quad7 q;
                            double PF[7][5];
#pragma unroll
                            for (int r = 0; r < 5; ++r) {
                                PF[0][r] = P[r][0];
                                PF[1][r] = P[r][1];
                                PF[2][r] = A02 * P[r][0] + A12 * P[r][1] + P[r][2];
                                PF[3][r] = 0.0;
                                PF[4][r] = 0.0;
                                PF[5][r] = B00 * P[r][0] + B10 * P[r][1] + P[r][3];
                                PF[6][r] = dt * P[r][2] + P[r][4];
                            }
#pragma unroll
                            for (int kk = 0; kk < 7; ++kk) {
                                q.H[0][kk] = PF[kk][0];
                                q.H[1][kk] = PF[kk][1];
                                q.H[2][kk] = (kk >= 2) ? A02 * PF[kk][0] + A12 * PF[kk][1] + PF[kk][2] : 0.0;
                                q.H[3][kk] = 0.0;
                                q.H[4][kk] = 0.0;
                                q.H[5][kk] = (kk >= 5) ? B00 * PF[kk][0] + B10 * PF[kk][1] + PF[kk][3] : 0.0;
                                q.H[6][kk] = (kk >= 6) ? dt * PF[kk][2] + PF[kk][4] : 0.0;
                            }
#pragma unroll
                            for (int r = 0; r < 7; ++r)
#pragma unroll
                                for (int s2 = 0; s2 < r; ++s2) q.H[r][s2] = q.H[s2][r]; // mirror
                            q.h[0] = sv[0];
                            q.h[1] = sv[1];
                            q.h[2] = A02 * sv[0] + A12 * sv[1] + sv[2];
                            q.h[3] = 0.0;
                            q.h[4] = 0.0;
                            q.h[5] = B00 * sv[0] + B10 * sv[1] + sv[3];
                            q.h[6] = dt * sv[2] + sv[4];
                            q.H[5][5] += Ruu0[0]; q.H[6][6] += Ruu0[1];
                            q.h[5] += -2.0 * c.matrix_q[2] * dr0[s];
                            if (k >= 1) {
#pragma unroll
                                for (int cc = 0; cc < 2; ++cc) {
                                    q.H[5 + cc][5 + cc] += Rd2[cc]; q.H[3 + cc][3 + cc] += Rd2[cc];
                                    q.H[5 + cc][3 + cc] -= Rd2[cc]; q.H[3 + cc][5 + cc] -= Rd2[cc];
                                }
                            }
                            { // eliminate u1 (index 6): u1 = a1 . w[0..5] + f1
                                double a1[6], f1;
                                const double inv = pivot_rcp_d(q.H[6][6]);
                                const bool fr = st1 == FREE, bx = (st1 == BOX_LO || st1 == BOX_HI);
                                const double invm = fr ? -inv : 0.0, rate1 = (fr || bx) ? 0.0 : 1.0; // masks instead of per-entry selects
#pragma unroll
                                for (int kk = 0; kk < 6; ++kk) a1[kk] = (kk == 4) ? q.H[6][kk] * invm + rate1 : q.H[6][kk] * invm;
                                f1 = fr ? -q.h[6] * inv : (bx ? (st1 == BOX_LO ? lo_eff[1] : hi_eff[1]) : (st1 == RATE_LO ? -rmax[1] : rmax[1]));
#pragma unroll
                                for (int kk = 0; kk < 6; ++kk) W_[6 + kk] = a1[kk];
                                W_[12] = f1;
#pragma unroll
                                for (int kk = 0; kk < 7; ++kk) W_[13 + kk] = q.H[6][kk];
                                W_[20] = q.h[6];
                                const double Hkk = q.H[6][6], hk = q.h[6];
                                double Hk[6], tr[6];
#pragma unroll
                                for (int kk = 0; kk < 6; ++kk) { Hk[kk] = q.H[6][kk]; tr[kk] = Hk[kk] + a1[kk] * Hkk; }

                                // H += a Hk' + tr a' on the upper triangle.  tr = Hk + a Hkk vanishes for a free component, a vanishes
                                // for a boxed one, and a rate-limited one has a = e_4: the second term lives in column 4 only
#pragma unroll
                                for (int r = 0; r < 6; ++r) {
                                    q.h[r] += a1[r] * hk + tr[r] * f1;
#pragma unroll
                                    for (int s2 = r; s2 < 6; ++s2) {
                                        q.H[r][s2] += a1[r] * Hk[s2] + ((s2 == 4) ? tr[r] * rate1 : 0.0);
                                        q.H[s2][r] = q.H[r][s2];
                                    }
                                }
                            }
                            { // eliminate u0 (index 5): u0 = a0 . xi + f0
                                double a0[5], f0;
                                const double inv = pivot_rcp_d(q.H[5][5]);
                                const bool fr = st0 == FREE, bx = (st0 == BOX_LO || st0 == BOX_HI);
                                const double invm = fr ? -inv : 0.0, rate1 = (fr || bx) ? 0.0 : 1.0;
#pragma unroll
                                for (int kk = 0; kk < 5; ++kk) a0[kk] = (kk == 3) ? q.H[5][kk] * invm + rate1 : q.H[5][kk] * invm;
                                f0 = fr ? -q.h[5] * inv : (bx ? (st0 == BOX_LO ? lo_eff[0] : hi_eff[0]) : (st0 == RATE_LO ? -rmax[0] : rmax[0]));
#pragma unroll
                                for (int kk = 0; kk < 5; ++kk) W_[21 + kk] = a0[kk];
                                W_[26] = f0;
#pragma unroll
                                for (int kk = 0; kk < 6; ++kk) W_[27 + kk] = q.H[5][kk];
                                W_[33] = q.h[5];
                                const double Hkk = q.H[5][5], hk = q.h[5];
                                double Hk[5], tr[5];
#pragma unroll
                                for (int kk = 0; kk < 5; ++kk) { Hk[kk] = q.H[5][kk]; tr[kk] = Hk[kk] + a0[kk] * Hkk; }

#pragma unroll
                                for (int r = 0; r < 5; ++r) {
                                    p[r] = q.h[r] + a0[r] * hk + tr[r] * f0;
#pragma unroll
                                    for (int s2 = r; s2 < 5; ++s2) { // as above; a rate-limited u0 has a = e_3
                                        P[r][s2] = q.H[r][s2] + a0[r] * Hk[s2] + ((s2 == 3) ? tr[r] * rate1 : 0.0);
                                        P[s2][r] = P[r][s2];
                                    }
                                }
                            }
                            { // boxes of stage k - 1
                                const double l0 = lo_eff[0], h0 = hi_eff[0], l1 = lo_eff[1], h1 = hi_eff[1];
                                lo_eff[0] = (st0 == RATE_LO) ? fmax(-umax[0], l0 + rmax[0]) : -umax[0];
                                hi_eff[0] = (st0 == RATE_HI) ? fmin(umax[0], h0 - rmax[0]) : umax[0];
                                lo_eff[1] = (st1 == RATE_LO) ? fmax(-umax[1], l1 + rmax[1]) : -umax[1];
                                hi_eff[1] = (st1 == RATE_HI) ? fmin(umax[1], h1 - rmax[1]) : umax[1];
                            }
                            if (act && j == t) { // the record of the stage: stores only
#pragma unroll
                                for (int f = 6; f < NF; ++f) REC(k, f) = W_[f];
                            }
                        }
                    }
                }
                // hand the cost-to-go to the lane below (upper triangle)
                if (t == 0) break;
#pragma unroll
                for (int r = 0; r < 5; ++r) {
                    p[r] = next_d(p[r]);
#pragma unroll
                    for (int s2 = r; s2 < 5; ++s2) { P[r][s2] = next_d(P[r][s2]); P[s2][r] = P[r][s2]; }
                }
                lo_eff[0] = next_d(lo_eff[0]); lo_eff[1] = next_d(lo_eff[1]); hi_eff[0] = next_d(hi_eff[0]); hi_eff[1] = next_d(hi_eff[1]);
            }
            LSTAMP(1)
            // ---- forward sweep, all lanes at once (the records of a lane's stages do not change during it):
            //   1. every stage's closed-loop map xi+ = M xi + m from its record (u0 = a0.xi + f0, u1 = a1.(xi, u0) + f1),
            //      composed over the lane's block;
            //   2. the state entering every block: L - 1 rounds of "apply the block map, hand to the lane above" -- after
            //      round r the lanes 0 .. r + 1 hold their true entry state (lane 0 holds pos0 throughout), 25 multiply-adds
            //      and 5 DPP row shifts a round;
            //   3. every lane walks its own stages: inputs, gradient rows of the non-free components, violations;
            //   4. the multiplier chain of the rate-limited runs (a tightened box followed by rate limits of the same
            //      direction passes its multiplier along): a two-value state per component, by two scans over the stages;
            //   5. the next working set: every change at once, or (from sweep SINGLE_AFTER on) only the most severe one.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const bool single = sweeps >= SINGLE_AFTER;
            int changes = 0;
            double best_sev = -1.0;
            int best_j = 0, best_c = 0, best_ns = 0;
            {
                double Mb[5][5], mb[5]; // block map
                double xin[5] = {pos0[0], pos0[1], pos0[2], 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const int k = j * S + s, kc = min(k, K - 1);
                    const bool vs = k < K;
                    double a0[5], b1[5];
#pragma unroll
                    for (int kk = 0; kk < 5; ++kk) a0[kk] = REC(kc, 21 + kk);
                    const double f0 = REC(kc, 26), a15 = REC(kc, 11), f1 = REC(kc, 12);
#pragma unroll
                    for (int kk = 0; kk < 5; ++kk) b1[kk] = REC(kc, 6 + kk) + a15 * a0[kk];
                    const double e1 = f1 + a15 * f0;
                    const double A02 = REC(kc, 0), A12 = REC(kc, 1), B00 = REC(kc, 2), B10 = REC(kc, 3), C0 = REC(kc, 4), C1 = REC(kc, 5);
                    if (s == 0) {
#pragma unroll
                        for (int cc = 0; cc < 5; ++cc) {
                            Mb[3][cc] = a0[cc]; Mb[4][cc] = b1[cc];
                            Mb[0][cc] = B00 * a0[cc] + (cc == 0 ? 1.0 : 0.0) + (cc == 2 ? A02 : 0.0);
                            Mb[1][cc] = B10 * a0[cc] + (cc == 1 ? 1.0 : 0.0) + (cc == 2 ? A12 : 0.0);
                            Mb[2][cc] = dt * b1[cc] + (cc == 2 ? 1.0 : 0.0);
                        }
                        mb[0] = B00 * f0 + C0; mb[1] = B10 * f0 + C1; mb[2] = dt * e1; mb[3] = f0; mb[4] = e1;
                    } else {
                        double n3[5], n4[5], n0[5], n1[5], n2[5];
#pragma unroll
                        for (int cc = 0; cc < 5; ++cc) {
                            n3[cc] = a0[0] * Mb[0][cc] + a0[1] * Mb[1][cc] + a0[2] * Mb[2][cc] + a0[3] * Mb[3][cc] + a0[4] * Mb[4][cc];
                            n4[cc] = b1[0] * Mb[0][cc] + b1[1] * Mb[1][cc] + b1[2] * Mb[2][cc] + b1[3] * Mb[3][cc] + b1[4] * Mb[4][cc];
                            n0[cc] = Mb[0][cc] + A02 * Mb[2][cc] + B00 * n3[cc];
                            n1[cc] = Mb[1][cc] + A12 * Mb[2][cc] + B10 * n3[cc];
                            n2[cc] = Mb[2][cc] + dt * n4[cc];
                        }
                        const double u0c = a0[0] * mb[0] + a0[1] * mb[1] + a0[2] * mb[2] + a0[3] * mb[3] + a0[4] * mb[4] + f0;
                        const double u1c = b1[0] * mb[0] + b1[1] * mb[1] + b1[2] * mb[2] + b1[3] * mb[3] + b1[4] * mb[4] + e1;
                        const double c0 = mb[0] + A02 * mb[2] + B00 * u0c + C0, c1 = mb[1] + A12 * mb[2] + B10 * u0c + C1, c2 = mb[2] + dt * u1c;
#pragma unroll
                        for (int cc = 0; cc < 5; ++cc) { // stages past the horizon are identities
                            Mb[0][cc] = vs ? n0[cc] : Mb[0][cc]; Mb[1][cc] = vs ? n1[cc] : Mb[1][cc]; Mb[2][cc] = vs ? n2[cc] : Mb[2][cc];
                            Mb[3][cc] = vs ? n3[cc] : Mb[3][cc]; Mb[4][cc] = vs ? n4[cc] : Mb[4][cc];
                        }
                        mb[0] = vs ? c0 : mb[0]; mb[1] = vs ? c1 : mb[1]; mb[2] = vs ? c2 : mb[2]; mb[3] = vs ? u0c : mb[3]; mb[4] = vs ? u1c : mb[4];
                    }
                }
                for (int r = 0; r < top; ++r) { // wavefront-uniform trip count
                    double o[5];
#pragma unroll
                    for (int rr = 0; rr < 5; ++rr)
                        o[rr] = Mb[rr][0] * xin[0] + Mb[rr][1] * xin[1] + Mb[rr][2] * xin[2] + Mb[rr][3] * xin[3] + Mb[rr][4] * xin[4] + mb[rr];
#pragma unroll
                    for (int rr = 0; rr < 5; ++rr) xin[rr] = dppd<0x111>(xin[rr], o[rr]); // lane 0 of the row keeps pos0
                }
                // 3. the lane's own stages
                double grad[S][2], val[S][2], prv[S][2], lo[S][2], hi[S][2];
                {
                    double xi[5] = {xin[0], xin[1], xin[2], xin[3], xin[4]};
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        const int k = j * S + s, kc = min(k, K - 1);
                        double Rr[NF];
#pragma unroll
                        for (int f = 0; f < NF; ++f) Rr[f] = REC(kc, f);
                        double w7[7];
#pragma unroll
                        for (int kk = 0; kk < 5; ++kk) w7[kk] = xi[kk];
                        double u0 = Rr[26];
#pragma unroll
                        for (int kk = 0; kk < 5; ++kk) u0 += Rr[21 + kk] * xi[kk];
                        w7[5] = u0;
                        double u1 = Rr[12];
#pragma unroll
                        for (int kk = 0; kk < 6; ++kk) u1 += Rr[6 + kk] * w7[kk];
                        w7[6] = u1;
                        double g1 = Rr[20], g0 = Rr[33];
#pragma unroll
                        for (int kk = 0; kk < 7; ++kk) g1 += Rr[13 + kk] * w7[kk];
#pragma unroll
                        for (int kk = 0; kk < 6; ++kk) g0 += Rr[27 + kk] * w7[kk];
                        grad[s][0] = g0; grad[s][1] = g1; val[s][0] = u0; val[s][1] = u1; prv[s][0] = xi[3]; prv[s][1] = xi[4];
                        lo[s][0] = Rr[34]; hi[s][0] = Rr[35]; lo[s][1] = Rr[36]; hi[s][1] = Rr[37];
                        const bool keep = act && (k < K);
                        ua[s] = keep ? u0 : ua[s]; uw[s] = keep ? u1 : uw[s];
                        const double nx = xi[0] + Rr[0] * xi[2] + Rr[2] * u0 + Rr[4], ny = xi[1] + Rr[1] * xi[2] + Rr[3] * u0 + Rr[5];
                        xi[2] = xi[2] + dt * u1; xi[0] = nx; xi[1] = ny; xi[3] = u0; xi[4] = u1;
                    }
                }
                // 4. multiplier chains.  Per stage and component: what the stage makes of the chain state (mu, dir) it is
                //    handed -- FREE: clears it; BOX: sets it (its own gradient row if the box was tightened by a rate-limited
                //    run above, else clear); RATE: keeps it if the direction matches and the row is not violated against it.
                //    Everything that does not depend on the incoming chain state is worked out once per sweep.
                bool is_free[S][2], is_box[S][2], lower_[S][2];
                int ns_fb[S][2], cd_box[S][2];
                double sev_fb[S][2], cm_box[S][2];
#pragma unroll
                for (int s = 0; s < S; ++s) {
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc) {
                        const int k = j * S + s;
                        const int sc = st[s][cc];
                        const double v = val[s][cc], gr = grad[s][cc], l = lo[s][cc], h = hi[s][cc];
                        const double dv = v - prv[s][cc];
                        const bool fr = sc == FREE, bx = (sc == BOX_LO || sc == BOX_HI);
                        const bool lower = (sc == BOX_LO || sc == RATE_LO);
                        // FREE: the more violated of box and rate limit becomes active
                        const double vb = fmax(l - v, v - h);
                        const double vr = (k >= 1) ? fmax(-rmax[cc] - dv, dv - rmax[cc]) : -1.0;
                        const bool f_box = vb > tol && vb >= vr, f_rate = !f_box && vr > tol;
                        const int ns_free = f_box ? ((v < l) ? BOX_LO : BOX_HI) : (f_rate ? ((dv < 0.0) ? RATE_LO : RATE_HI) : FREE);
                        const double sev_free = f_box ? vb : (f_rate ? vr : 0.0);
                        // BOX: released by the sign of its multiplier, else handed to a violated rate limit
                        const bool tightened = lower ? (l > -umax[cc] + 1e-12) : (h < umax[cc] - 1e-12);
                        const bool b_rel = (lower && gr < -tol) || (!lower && gr > tol);
                        const bool b_rate = !b_rel && k >= 1 && fabs(dv) > rmax[cc] + tol;
                        const int ns_box = b_rel ? FREE : (b_rate ? ((dv < 0.0) ? RATE_LO : RATE_HI) : sc);
                        const double sev_box = b_rel ? fabs(gr) : (b_rate ? fabs(dv) - rmax[cc] : 0.0);
                        is_free[s][cc] = fr; is_box[s][cc] = bx; lower_[s][cc] = lower;
                        ns_fb[s][cc] = fr ? ns_free : ns_box; sev_fb[s][cc] = fr ? sev_free : sev_box;
                        cm_box[s][cc] = (bx && tightened) ? gr : 0.0;
                        cd_box[s][cc] = (bx && tightened) ? (lower ? -1 : 1) : 0;
                    }
                }
                // a rate-limited stage against the chain state it is handed; FREE / BOX stages replace the state
                auto chain_step = [&](int s, int cc, double& cm, int& cd, int& ns, double& sev) {
                    const bool lower = lower_[s][cc];
                    const int md = lower ? -1 : 1;
                    const bool match = cd == md;
                    const double cmr = match ? cm : 0.0;
                    const double g_eff = grad[s][cc] - cmr;
                    const bool r_viol = lower ? (g_eff < -tol) : (g_eff > tol);
                    const bool rate = !(is_free[s][cc] || is_box[s][cc]);
                    const int ns_rate = r_viol ? ((cmr == 0.0) ? FREE : (lower ? BOX_LO : BOX_HI)) : st[s][cc];
                    ns = rate ? ns_rate : ns_fb[s][cc];
                    sev = rate ? (r_viol ? fabs(g_eff) : 0.0) : sev_fb[s][cc];
                    const bool r_keep = match && !r_viol;
                    cm = rate ? (r_keep ? cm : 0.0) : cm_box[s][cc];
                    cd = rate ? (r_keep ? cd : 0) : cd_box[s][cc];
                };
                // The state a stage is handed is that of the nearest FREE / BOX stage before it ("setter"; the empty chain before
                // stage 0) if every rate-limited stage in between kept it, else the empty chain -- two scans over the stages
                // (serial inside the lane, DPP row shifts across): the setter's state, then "kept so far" with restarts.
                double in_m[S][2];
                int in_d[S][2];
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    bool has = false;
                    double lm = 0.0;
                    int ld = 0;
                    double e_m[S]; int e_d[S]; bool e_has[S]; // setter state before stage s, from inside the lane
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        e_m[s] = lm; e_d[s] = ld; e_has[s] = has;
                        const bool set = (j * S + s < K) && (is_free[s][cc] || is_box[s][cc]);
                        has = has || set; lm = set ? cm_box[s][cc] : lm; ld = set ? cd_box[s][cc] : ld;
                    }
                    int hc = has ? 1 : 0;
                    auto copy_level = [&](auto tag) { // (hc, lm, ld) <- the lane's own if it has a setter, else the one from below
                        constexpr int CTRL = decltype(tag)::value;
                        const int oh = dppi<CTRL>(0, hc), od = dppi<CTRL>(0, ld);
                        const double om = dppd<CTRL>(0.0, lm);
                        const bool take = hc == 0;
                        lm = take ? om : lm; ld = take ? od : ld; hc = take ? oh : hc;
                    };
                    copy_level(std::integral_constant<int, 0x111>{});
                    copy_level(std::integral_constant<int, 0x112>{});
                    copy_level(std::integral_constant<int, 0x114>{});
                    copy_level(std::integral_constant<int, 0x118>{});
                    const double xm = dppd<0x111>(0.0, lm); // exclusive: what the lanes below leave (lane 0: the empty chain)
                    const int xd = dppi<0x111>(0, ld);
                    int acc = 0, before[S]; // bit 1: a setter seen, bit 0: a rate-limited stage after the last setter dropped the chain
                    auto combine = [](int l, int r) { return (r & 2) ? r : (l | r); };
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        e_m[s] = e_has[s] ? e_m[s] : xm; e_d[s] = e_has[s] ? e_d[s] : xd;
                        const bool vs = j * S + s < K;
                        const bool set = is_free[s][cc] || is_box[s][cc];
                        const bool lower = lower_[s][cc];
                        const double g_eff = grad[s][cc] - e_m[s];
                        const bool keeps = (e_d[s] == (lower ? -1 : 1)) && !(lower ? (g_eff < -tol) : (g_eff > tol));
                        before[s] = acc;
                        acc = combine(acc, vs ? (set ? 2 : (keeps ? 0 : 1)) : 0);
                    }
                    auto and_level = [&](auto tag) {
                        constexpr int CTRL = decltype(tag)::value;
                        acc = combine(dppi<CTRL>(0, acc), acc);
                    };
                    and_level(std::integral_constant<int, 0x111>{});
                    and_level(std::integral_constant<int, 0x112>{});
                    and_level(std::integral_constant<int, 0x114>{});
                    and_level(std::integral_constant<int, 0x118>{});
                    const int xacc = dppi<0x111>(0, acc);
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        const bool kept = (combine(xacc, before[s]) & 1) == 0;
                        in_m[s][cc] = kept ? e_m[s] : 0.0;
                        in_d[s][cc] = kept ? e_d[s] : 0;
                    }
                }
                // 5. the verdicts
#pragma unroll
                for (int s = 0; s < S; ++s) {
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc) {
                        int ns_; double sev_;
                        double m_ = in_m[s][cc]; int d_ = in_d[s][cc];
                        chain_step(s, cc, m_, d_, ns_, sev_);
                        const bool vs = j * S + s < K;
                        const bool ch = vs && ns_ != st[s][cc];
                        changes += ch ? 1 : 0;
                        const bool better = ch && sev_ > best_sev; // first of the most severe, in stage order
                        best_sev = better ? sev_ : best_sev; best_j = better ? j * S + s : best_j; best_c = better ? cc : best_c;
                        best_ns = better ? ns_ : best_ns;
                        st[s][cc] = (ch && act && !single) ? ns_ : st[s][cc];
                    }
                }
                // group totals: any change; the most severe one (ties: the lowest lane, i.e. the earliest stage)
                {
                    const unsigned long long mball = __ballot(changes != 0);
                    changes = (int)((mball >> (lane & ~15)) & 0xFFFFull);
                    auto red = [&](auto tag) {
                        constexpr int CTRL = decltype(tag)::value;
                        const double osev = dppd<CTRL>(-1.0, best_sev);
                        const int oj = dppi<CTRL>(0, best_j), oc = dppi<CTRL>(0, best_c), on = dppi<CTRL>(0, best_ns);
                        const bool take = osev >= best_sev && osev > 0.0; // the value from the lower lanes wins ties
                        best_sev = take ? osev : best_sev; best_j = take ? oj : best_j; best_c = take ? oc : best_c; best_ns = take ? on : best_ns;
                    };
                    if (single) {
                        red(std::integral_constant<int, 0x111>{});
                        red(std::integral_constant<int, 0x112>{});
                        red(std::integral_constant<int, 0x114>{});
                        red(std::integral_constant<int, 0x118>{});
                        best_sev = dppd<0x15F>(best_sev, best_sev);
                        best_j = dppi<0x15F>(best_j, best_j); best_c = dppi<0x15F>(best_c, best_c); best_ns = dppi<0x15F>(best_ns, best_ns);
                    }
                }
            }
            LSTAMP(2)
            if (act) {
                if (single && changes > 0) {
#pragma unroll
                    for (int s = 0; s < S; ++s)
                        if (j * S + s == best_j) { if (best_c == 0) st[s][0] = best_ns; else st[s][1] = best_ns; }
                }
                settled = (changes == 0);
                ++sweeps;
            }
        }
        status = settled ? 0 : 1;
    }
    // ---- predictMotion(xopt): the linear prediction about the rollout of the final output (see the kernel above for
    //      which rollout the reference uses), by prefix sums; column m of the output drives step m -> m + 1
    {
        double* xo = d.xopt + (size_t)b * (T + 1) * 3;
        const double x0 = nowp[0], y0 = nowp[1], th0 = nowp[2];
        double bth = th0, bv = 0.0, tx = x0, ty = y0, tth = th0;
        if (valid && j == 0) { xo[0] = tx; xo[1] = ty; xo[2] = tth; }
        for (int i = 0; i < dl; ++i) { // the delayed inputs (now the delay buffer), every lane
            const double u0 = zero_hist ? 0.0 : bf[2 * i], u1 = zero_hist ? 0.0 : bf[2 * i + 1];
            double sb_, cb_;
            sincos(bth, &sb_, &cb_);
            const double B00 = cb_ * dt, B10 = sb_ * dt, A02 = -B10 * bv, A12 = B00 * bv;
            const double nx = tx + A02 * tth + B00 * u0 - A02 * bth, ny = ty + A12 * tth + B10 * u0 - A12 * bth;
            tth = tth + dt * u1; tx = nx; ty = ny;
            if (valid && j == 0) { xo[3 * (i + 1)] = tx; xo[3 * (i + 1) + 1] = ty; xo[3 * (i + 1) + 2] = tth; }
            bth += fmin(fmax(u1, -c.max_omega), c.max_omega) * dt; bv = u0;
        }
        double hb[S], ht[S], ab = 0.0, at = 0.0;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const bool vs = j * S + s < K;
            hb[s] = ab; ht[s] = at;
            ab += vs ? fmin(fmax(uw[s], -c.max_omega), c.max_omega) * dt : 0.0;
            at += vs ? dt * uw[s] : 0.0;
        }
        const double eb = rowprefix_d(ab) - ab + bth, et = rowprefix_d(at) - at + tth;
        const double a_prev_lane = prev_d(ua[S - 1]);
        double ix[S], iy[S], sx = 0.0, sy = 0.0;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int k = j * S + s;
            const bool vs = k < K;
            const double bth_k = hb[s] + eb, tth_k = ht[s] + et;
            const double bv_k = (k == 0) ? bv : ((s == 0) ? a_prev_lane : ua[(s > 0) ? s - 1 : 0]);
            double sb_, cb_;
            sincos(bth_k, &sb_, &cb_);
            const double B00 = cb_ * dt, B10 = sb_ * dt, A02 = -B10 * bv_k, A12 = B00 * bv_k;
            sx += vs ? (A02 * tth_k + B00 * ua[s] - A02 * bth_k) : 0.0;
            sy += vs ? (A12 * tth_k + B10 * ua[s] - A12 * bth_k) : 0.0;
            ix[s] = sx; iy[s] = sy;
            ht[s] = tth_k + dt * uw[s]; // heading after the step
        }
        const double ex = rowprefix_d(sx) - sx + tx, ey = rowprefix_d(sy) - sy + ty;
        if (valid) {
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int k = j * S + s;
                if (k < K) {
                    const int i = dl + k + 1;
                    xo[3 * i] = ix[s] + ex; xo[3 * i + 1] = iy[s] + ey; xo[3 * i + 2] = ht[s];
                }
            }
        }
    }
    LSTAMP(3)
    // ---- state kept between calls: the output (its first columns are the delay buffer), the working set, the buffer
    const double c0 = __shfl(ua[0], lane & ~15), c1 = __shfl(uw[0], lane & ~15); // stage 0 = the command just computed
    if (valid && poisoned) {
        if (j == 0) {
            d.sweeps[b] = 0;
            d.status[b] = STATUS_NON_FINITE;
            d.cmd[2 * b] = 0.0; d.cmd[2 * b + 1] = 0.0;
            if (d.cmd_host) { d.cmd_host[2 * b] = 0.0; d.cmd_host[2 * b + 1] = 0.0; d.status_host[b] = STATUS_NON_FINITE; }
        }
    } else if (valid) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int k = j * S + s;
            if (k < K) {
                out[2 * (dl + k)] = ua[s]; out[2 * (dl + k) + 1] = uw[s];
                d.st[((size_t)k * 2) * d.stride + b] = st[s][0];
                d.st[((size_t)k * 2 + 1) * d.stride + b] = st[s][1];
            }
        }
        if (j == 0) {
            for (int i = 0; i < dl; ++i) { out[2 * i] = d.reset ? 0.0 : bf[2 * i]; out[2 * i + 1] = d.reset ? 0.0 : bf[2 * i + 1]; }
            if (dl > 0) { // output_buff: drop the oldest, append the command just computed
                for (int i = 0; i + 1 < dl; ++i) { bf[2 * i] = d.reset ? 0.0 : bf[2 * (i + 1)]; bf[2 * i + 1] = d.reset ? 0.0 : bf[2 * (i + 1) + 1]; }
                bf[2 * (dl - 1)] = c0; bf[2 * (dl - 1) + 1] = c1;
            }
            d.sweeps[b] = sweeps;
            d.status[b] = status;
            d.cmd[2 * b] = c0; d.cmd[2 * b + 1] = c1;
            if (d.cmd_host) { d.cmd_host[2 * b] = c0; d.cmd_host[2 * b + 1] = c1; d.status_host[b] = status; }
        }
    }
}

hipError_t launch_get_cmd(const Dev& d, bool thread_kernel, hipStream_t s)
{
    const int B = d.B, K = d.c.predict_steps - d.c.delay_num;
    if (thread_kernel)
        hipLaunchKernelGGL(get_cmd_kernel, dim3((B + 63) / 64), dim3(64), 0, s, d);
    else if (K <= 32)
        hipLaunchKernelGGL(get_cmd_lanes_kernel<2>, dim3((B + 3) / 4), dim3(64), ((size_t)K * REC_STRIDE + 16) * sizeof(double), s, d);
    else {
        const size_t lds = ((size_t)K * REC_STRIDE + 16) * sizeof(double);
        static bool raised = false;
        if (!raised) {
            const hipError_t e = hipFuncSetAttribute((const void*)get_cmd_lanes_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (64 * REC_STRIDE + 16) * 8);
            if (e != hipSuccess) return e;
            raised = true;
        }
        hipLaunchKernelGGL(get_cmd_lanes_kernel<4>, dim3((B + 3) / 4), dim3(64), lds, s, d);
    }
    return hipGetLastError();
}

} // namespace ltv
