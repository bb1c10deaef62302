// wb_kernels.hip -- whole-body NMPC class (include/alore_wb.h): linearisation and Riccati kernels + C ABI.
//
// No reference code exists for this class (SURVEY.md 8(a) row A-RB); the OCP is stated in include/alore_wb.h.
//
// Kernel 1, stage_kernel: ONE WAVEFRONT PER (problem, stage).  Lanes are independent RNEA evaluations
// (wb_dynamics.h) that differ by a unit vector or a perturbation; sines / cosines come from one table per stage:
//   pass 0    24 lanes: columns of M(q) (unit accelerations, no gravity) | 1 lane: bias RNEA(q, v, 0, f) |
//             12 lanes: columns of -J_c' (unit foot forces)
//   M^-1 by an in-register Gauss-Jordan (wave_linalg.h: lane = row, pivot rows by v_readlane), a = M^-1 ([0; tau] - bias)
//   pass 1    45 lanes: d RNEA(q, v, a, f) / d (rpy, joints, v) by forward differences against the base point (lane 45)
//   57 lanes: M^-1 times the derivative and foot-force columns -> da/dq, da/dv, da/df;  da/dtau = M^-1 itself
//   assembly, lane = column of [A_k | B_k] (48 x 48, 48 x 30), and f(x_k, u_k) for the semi-implicit Euler step, to HBM
//   (float32 for the Riccati kernel, float64 on request).  18.4 KB of LDS, 8 wavefronts per CU.
// Kernel 2, riccati_kernel: one workgroup (4 wavefronts) per problem, float32 MFMA 16x16x4 on the dense blocks, torque
// limits inside the sweep; 53.6 KB of LDS, 3 workgroups per CU.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/alore_wb.h"
#include "wave_linalg.h"
#include "wb_aba.h"
#include "wb_dynamics.h"

namespace wb {

// Diagnostic phase stamps (alore_wb_debug_stamps): workgroup 0 adds the cycles since its previous stamp to slot i.
#define WB_STAMP(buf, i)                                                                     \
    if ((buf) && blockIdx.x == 0 && threadIdx.x == 0) {                                      \
        const long long now_ = (long long)__builtin_readcyclecounter();                      \
        (buf)[i] += now_ - (buf)[31];                                                        \
        (buf)[31] = now_;                                                                    \
    }

constexpr int NQ = 24, NV = 24, NX = 48, NU = 30, NUP = 32; // NUP: input dimension padded to the MFMA tile
constexpr int NCOL = 57;                                     // columns kept in LDS: 21 (rpy, joints) + 24 (v) + 12 (f); da/dtau is M^-1 itself
constexpr double HQ = 1e-7, HV = 1e-7;                       // forward-difference steps (float64; A, B are stored in float32)

struct StageArgs {
    const double* x; // [B][N+1][48]
    const double* u; // [B][N][30]
    int N;           // stages per problem
    int n_items;     // B * N
    double dt;
    float* A32;      // [n][48][48]
    float* B32;      // [n][48][32]
    double* next;    // [n][48]
    double* A64;     // optional [n][48][48]
    double* B64;     // optional [n][48][30]
    double* M64;     // optional [n][24][24]
    double* a64;     // optional [n][24]
    long long* stamps; // optional [32] diagnostic
    const double* xref = nullptr; // [B][N+1][48]  with uref and vec: the stage's right-hand sides for the Riccati kernel
    const double* uref = nullptr; // [B][N][30]
    float* vec = nullptr;         // [n][160]: defect f(x_k, u_k) - x_{k+1} | x_k - xref_k | u_k (32) | u_k - uref_k (32), float32
    // contact-consistency penalty 1/2 rho |J_c(q_k) v_k|^2 over the stance feet (alore_wb_set_contact_penalty): per stage the
    // scaled Jacobian sqrt(rho) J_c (rows of swing feet zero) [12][24] and the Gauss-Newton gradient rho J_c' (J_c v_k) [24]
    float* pen = nullptr;         // [n][PEN]
    double rho = 0.0;
    const unsigned char* stance = nullptr; // [n][4]
};

constexpr int VEC = 160; // floats per stage in StageArgs::vec
constexpr int PEN = 12 * 24 + 24; // floats per stage in StageArgs::pen
constexpr int MS = 25; // row stride of M / column stride of D in LDS (doubles)

struct StageLds {
    double q[NQ], v[NV], u[NU], a[NV], zero[NV];
    double M[NV * MS];
    double bias[NV];
    double D[NCOL * MS]; // D[col][i]
    double vn[NV];       // v+
    double trig[42];     // sin, cos of rpy and the joint angles at the base point
    double tb[NV];       // RNEA at the base point (= [0; tau] up to the rounding of M^-1)
    double R0[9], E[9], Gq[3][6];
    double xn[NX], xr[NX], ur[32]; // x_{k+1}, xref_k, uref_k: fetched with the stage's inputs, used for StageArgs::vec at the end
};

template <bool PENALTY>
__global__ __launch_bounds__(64, 2) void stage_kernel(StageArgs g)
{
    __shared__ StageLds S;
    const int item = blockIdx.x;
    if (item >= g.n_items) return;
    if (g.stamps && blockIdx.x == 0 && threadIdx.x == 0) g.stamps[31] = (long long)__builtin_readcyclecounter();
    const int lane = threadIdx.x;
    const int b = item / g.N, k = item % g.N;
    const double* xk = g.x + ((size_t)b * (g.N + 1) + k) * NX;
    const double* uk = g.u + ((size_t)b * g.N + k) * NU;
    if (lane < NQ) { S.q[lane] = xk[lane]; S.v[lane] = xk[NQ + lane]; S.zero[lane] = 0.0; }
    if (lane < NU) S.u[lane] = uk[lane];
    if (g.vec) {
        if (lane < NX) { S.xn[lane] = xk[NX + lane]; S.xr[lane] = g.xref[((size_t)b * (g.N + 1) + k) * NX + lane]; }
        if (lane < NU) S.ur[lane] = g.uref[((size_t)b * g.N + k) * NU + lane];
    }
    if (lane < 21) sincos(xk[3 + lane], &S.trig[2 * lane], &S.trig[2 * lane + 1]);
    __syncthreads();
    WB_STAMP(g.stamps, 0)

    // Two passes over ONE inlined copy of the RNEA (the loop is kept rolled): pass 0 = M columns, bias, foot-force
    // columns; pass 1 = the 45 perturbed evaluations of d RNEA / d (rpy, joints, v) and the base point (forward
    // differences).  Outputs go straight to LDS; sines and cosines come from the table of the base point.
    double ph_s, ph_c;
    sincos(HQ, &ph_s, &ph_c);
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        Eval e{S.q, S.v, S.zero, S.u + 18, 0.0, 0.0, 0.0, -1, -1, -1, -1, 0.0, 0.0, 0.0, 0.0, 0.0};
        e.trig = S.trig; e.ps = ph_s; e.pc = ph_c;
        Sink sink{S.D + lane * MS, 1, 1.0, 0};
        bool active;
        if (pass == 0) {
            active = lane < 37;
            if (lane < 24) { e.ua = lane; e.da = 1.0; sink.out = S.M + lane; sink.stride = MS; }
            else if (lane == 24) { e.sv = 1.0; e.sf = 1.0; e.g = b2z1::GRAVITY; sink.out = S.bias; }
            else { e.uf = lane - 25; e.df = 1.0; sink.out = S.D + (45 + lane - 25) * MS; sink.scale = -1.0; } // RNEA = ... - J_c' f
        } else {
            active = lane < 46;
            e.a = S.a; e.sv = 1.0; e.sa = 1.0; e.sf = 1.0; e.g = b2z1::GRAVITY;
            if (lane < 21) { e.uq = 3 + lane; e.dq = HQ; } else if (lane < 45) { e.uv = lane - 21; e.dv = HV; } else sink.out = S.tb;
        }
        if (active) rnea(e, sink);
        if (pass == 0) {
            __syncthreads();
    WB_STAMP(g.stamps, 1)
            if (g.M64 && lane < 24)
                for (int i = 0; i < NV; ++i) g.M64[((size_t)item * NV + i) * NV + lane] = S.M[i * MS + lane];
            // ---- M^-1 in place (lane i owns row i; wave_linalg.h), then a = M^-1 ([0; tau] - bias)
            // (the two-lanes-per-row variant used by the Riccati kernel does not pay here: a float64 element costs two
            //  cross-lane permutes, measured equal)
            {
                double row[NV];
                const int rr = lane < NV ? lane : 0;
#pragma unroll
                for (int j = 0; j < NV; ++j) row[j] = S.M[rr * MS + j];
                wavela::spd_inverse_rows<double, NV>(row, lane);
                __syncthreads();
                if (lane < NV) {
                    double acc = 0.0;
#pragma unroll
                    for (int j = 0; j < NV; ++j) {
                        S.M[lane * MS + j] = row[j];
                        acc += row[j] * ((j >= 6 ? S.u[j - 6] : 0.0) - S.bias[j]);
                    }
                    S.a[lane] = acc;
                }
            }
            __syncthreads();
    WB_STAMP(g.stamps, 2)
            if (g.a64 && lane < NV) g.a64[(size_t)item * NV + lane] = S.a[lane];
            if (lane < NV) S.vn[lane] = S.v[lane] + g.dt * S.a[lane];
            if (PENALTY) { // the foot columns of pass 0 ARE the contact Jacobian: D[45 + i][r] = J_c[i][r] (world-frame point velocity)
                float* pk = g.pen + (size_t)item * PEN;
                const unsigned char* stc = g.stance ? g.stance + (size_t)item * 4 : nullptr;
                double* rt = &S.Gq[0][0]; // 12 doubles of scratch: Gq is written after pass 1
                if (lane < 12) {
                    double acc = 0.0;
#pragma unroll
                    for (int r = 0; r < NV; ++r) acc += S.D[(45 + lane) * MS + r] * S.v[r];
                    rt[lane] = (!stc || stc[lane / 3]) ? g.rho * acc : 0.0;
                }
                __syncthreads();
                if (lane < NV) {
                    double acc = 0.0;
#pragma unroll
                    for (int i = 0; i < 12; ++i) acc += S.D[(45 + i) * MS + lane] * rt[i];
                    pk[288 + lane] = (float)acc;
                }
                const double sq = sqrt(g.rho);
                for (int e = lane; e < 288; e += 64) {
                    const int i = e / 24, r = e % 24;
                    pk[e] = (!stc || stc[i / 3]) ? (float)(sq * S.D[(45 + i) * MS + r]) : 0.f;
                }
            }
            __syncthreads();
    WB_STAMP(g.stamps, 3)
        }
    }
    __syncthreads();
    WB_STAMP(g.stamps, 4)
    if (lane < 45) { // D <- -(RNEA(x + h e) - RNEA(x)) / h
        const double sc = -1.0 / (lane < 21 ? HQ : HV);
#pragma unroll
        for (int i = 0; i < NV; ++i) S.D[lane * MS + i] = sc * (S.D[lane * MS + i] - S.tb[i]);
    }
    if (lane >= 45 && lane < 48) { // derivative of the kinematic map G(q) v+ with respect to rpy (v+ fixed)
        const int c = lane - 45;
        const V3 wn = {S.vn[0], S.vn[1], S.vn[2]}, vl = {S.vn[3], S.vn[4], S.vn[5]};
        const double h = 1e-6;
        // sines / cosines of rpy from the stage's table; the perturbed angle by the addition theorem (sin h, cos h of h = 1e-6)
        constexpr double sh = 9.99999999999833333e-07, ch = 0.9999999999995;
        BaseRot Rp, Rm;
        Rp.sr = Rm.sr = S.trig[0]; Rp.cr = Rm.cr = S.trig[1]; Rp.sp = Rm.sp = S.trig[2]; Rp.cp = Rm.cp = S.trig[3];
        Rp.sy = Rm.sy = S.trig[4]; Rp.cy = Rm.cy = S.trig[5];
        {
            const double s0 = S.trig[2 * c], c0 = S.trig[2 * c + 1];
            const double sp_ = s0 * ch + c0 * sh, cp_ = c0 * ch - s0 * sh, sm_ = s0 * ch - c0 * sh, cm_ = c0 * ch + s0 * sh;
            if (c == 0) { Rp.sr = sp_; Rp.cr = cp_; Rm.sr = sm_; Rm.cr = cm_; }
            else if (c == 1) { Rp.sp = sp_; Rp.cp = cp_; Rm.sp = sm_; Rm.cp = cm_; }
            else { Rp.sy = sp_; Rp.cy = cp_; Rm.sy = sm_; Rm.cy = cm_; }
        }
        const V3 dp = (0.5 / h) * (Rp.toWorld(vl) - Rm.toWorld(vl)), dr = (0.5 / h) * (Rp.rates(wn) - Rm.rates(wn));
        S.Gq[c][0] = dp.x; S.Gq[c][1] = dp.y; S.Gq[c][2] = dp.z; S.Gq[c][3] = dr.x; S.Gq[c][4] = dr.y; S.Gq[c][5] = dr.z;
    } else if (lane == 48) {
        BaseRot R;
        R.sr = S.trig[0]; R.cr = S.trig[1]; R.sp = S.trig[2]; R.cp = S.trig[3]; R.sy = S.trig[4]; R.cy = S.trig[5];
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const V3 em = {m == 0 ? 1.0 : 0.0, m == 1 ? 1.0 : 0.0, m == 2 ? 1.0 : 0.0};
            const V3 cw = R.toWorld(em), ce = R.rates(em);
            S.R0[0 * 3 + m] = cw.x; S.R0[1 * 3 + m] = cw.y; S.R0[2 * 3 + m] = cw.z;
            S.E[0 * 3 + m] = ce.x; S.E[1 * 3 + m] = ce.y; S.E[2 * 3 + m] = ce.z;
        }
    }
    __syncthreads();
    WB_STAMP(g.stamps, 5)

    // ---- M^-1 times the 57 columns that need it (45 derivative columns, 12 foot-force columns) on the float64 matrix
    //      cores: v_mfma_f64_16x16x4_f64, 2 row tiles x 4 column tiles x 6 k-steps.  Lane l supplies A[l & 15][l >> 4] and
    //      B[l >> 4][l & 15] (one double each) and receives C[(l >> 4) + 4 r][l & 15], r = 0..3.  Rows >= 24 and columns
    //      >= 57 are computed from clamped (duplicate) operands and not stored.  D is overwritten in place: a column
    //      tile's operands are all in registers before its results are stored (one wavefront, LDS in order).
    {
        typedef double d4 __attribute__((ext_vector_type(4)));
        const int r16 = lane & 15, kq = lane >> 4;
        const int row1 = 16 + r16 < NV ? 16 + r16 : NV - 1;
        double a0[6], a1[6];
#pragma unroll
        for (int t = 0; t < 6; ++t) { a0[t] = S.M[r16 * MS + 4 * t + kq]; a1[t] = S.M[row1 * MS + 4 * t + kq]; }
#pragma unroll 1
        for (int tj = 0; tj < 4; ++tj) {
            const int cfull = tj * 16 + r16, col = cfull < NCOL ? cfull : NCOL - 1;
            double bq[6];
#pragma unroll
            for (int t = 0; t < 6; ++t) bq[t] = S.D[col * MS + 4 * t + kq];
            d4 c0 = {0.0, 0.0, 0.0, 0.0}, c1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[t], bq[t], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[t], bq[t], c1, 0, 0, 0);
            }
            if (cfull < NCOL) {
#pragma unroll
                for (int r = 0; r < 4; ++r) S.D[col * MS + kq + 4 * r] = c0[r];
#pragma unroll
                for (int r = 0; r < 2; ++r) S.D[col * MS + 16 + kq + 4 * r] = c1[r]; // rows 16 .. 23
            }
        }
    }
    __syncthreads();
    WB_STAMP(g.stamps, 6)

    // ---- assembly: lane = column c of [A | B] (78 + 2 padding columns, two rounds); its da / d(var) column is read
    //      once, all 48 rows come from registers; for a fixed row the lanes store consecutive floats.
    //      dvn_i / d var = dt Z_i + [var is v_i];  q rows: [row == var] + dt (G dvn + dG/drpy v+)
    const double dt = g.dt;
    float* A32 = g.A32 + (size_t)item * NX * NX;
    float* B32 = g.B32 + (size_t)item * NX * NUP;
    {
        const int c = lane; // the first 64 columns: the 48 of A and the first 16 of B
        // source of this column's da / d(var): a derivative column of D, a row of M^-1 (torque j: column 6 + j of the
        // symmetric inverse), a foot-force column of D, or nothing (base position, padding)
        const double* zsrc = c < 3 ? nullptr
                             : (c < 24 ? S.D + (c - 3) * MS
                                : (c < 48 ? S.D + (21 + c - 24) * MS
                                   : (c < 48 + 18 ? S.M + (6 + c - 48) * MS : (c < NX + NU ? S.D + (45 + c - 66) * MS : nullptr))));
        double z[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) z[i] = (zsrc ? dt * zsrc[i] : 0.0) + ((c == 24 + i) ? 1.0 : 0.0);
        const bool pad = c >= NX + NU;
        float* out = c < NX ? (A32 + c) : (B32 + (c - NX));
        const int ld = c < NX ? NX : NUP;
        double* out64 = c < NX ? (g.A64 ? g.A64 + (size_t)item * NX * NX + c : nullptr) : ((g.B64 && !pad) ? g.B64 + (size_t)item * NX * NU + (c - NX) : nullptr);
        const int ld64 = c < NX ? NX : NU;
#pragma unroll
        for (int r = 0; r < NX; ++r) {
            double val;
            if (r >= 24) val = z[r - 24];
            else if (r >= 6) val = ((r == c) ? 1.0 : 0.0) + dt * z[r];
            else {
                const double* Gm = (r < 3) ? (S.R0 + 3 * r) : (S.E + 3 * (r - 3));
                const int off = (r < 3) ? 3 : 0; // p rows use v_lin (3..5), rpy rows use omega (0..2)
                double acc = Gm[0] * z[off] + Gm[1] * z[off + 1] + Gm[2] * z[off + 2];
                if (c >= 3 && c < 6) acc += S.Gq[c - 3][r];
                val = ((r == c) ? 1.0 : 0.0) + dt * acc;
            }
            if (pad) val = 0.0;
            out[r * ld] = (float)val;
            if (out64) out64[(size_t)r * ld64] = val;
        }
    }
    {   // the last 16 columns of B (torques 16, 17, the 12 foot forces, 2 padding columns) on FOUR lanes each, 12 rows per
        // lane: one lane per column would leave 48 lanes idle for a whole round of 48 rows.  No identity terms here
        // (r == c and c == 24 + i cannot happen for c >= 64).
        const int c = 64 + (lane >> 2), part = lane & 3;
        const bool pad = c >= NX + NU;
        const double* zsrc = c < 48 + 18 ? S.M + (6 + c - 48) * MS : (pad ? S.zero : S.D + (45 + c - 66) * MS);
        float* out = B32 + (c - NX);
        double* out64 = (g.B64 && !pad) ? g.B64 + (size_t)item * NX * NU + (c - NX) : nullptr;
        for (int rr = 0; rr < NX / 4; ++rr) {
            const int r = part + 4 * rr;
            double val;
            if (r >= 24) val = dt * zsrc[r - 24];
            else if (r >= 6) val = dt * (dt * zsrc[r]);
            else {
                const double* Gm = (r < 3) ? (S.R0 + 3 * r) : (S.E + 3 * (r - 3));
                const int off = (r < 3) ? 3 : 0;
                val = dt * (Gm[0] * (dt * zsrc[off]) + Gm[1] * (dt * zsrc[off + 1]) + Gm[2] * (dt * zsrc[off + 2]));
            }
            if (pad) val = 0.0;
            out[r * NUP] = (float)val;
            if (out64) out64[(size_t)r * NU] = val;
        }
    }
    if (lane < NX) { // f(x_k, u_k)
        double val;
        const int r = lane;
        if (r >= 24) val = S.vn[r - 24];
        else if (r >= 6) val = S.q[r] + dt * S.vn[r];
        else {
            const double* Gm = (r < 3) ? (S.R0 + 3 * r) : (S.E + 3 * (r - 3));
            const int off = (r < 3) ? 3 : 0;
            val = S.q[r] + dt * (Gm[0] * S.vn[off] + Gm[1] * S.vn[off + 1] + Gm[2] * S.vn[off + 2]);
        }
        g.next[(size_t)item * NX + r] = val;
        if (g.vec) {
            float* vo = g.vec + (size_t)item * VEC;
            vo[r] = (float)(val - S.xn[r]);
            vo[48 + r] = (float)((r < NQ ? S.q[r] : S.v[r - NQ]) - S.xr[r]);
        }
    }
    if (g.vec && lane < 32) {
        const int j = lane;
        float* vo = g.vec + (size_t)item * VEC;
        const float ucur = j < NU ? (float)S.u[j] : 0.f;
        vo[96 + j] = ucur;
        vo[128 + j] = j < NU ? ucur - (float)S.ur[j] : 0.f;
    }
    WB_STAMP(g.stamps, 7)
}

// one lane per evaluation point (alore_wb_rnea)
__global__ __launch_bounds__(64) void rnea_kernel(int n, const double* q, const double* v, const double* a, const double* f, double grav, double* tau)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    Eval e{q + (size_t)t * NQ, v + (size_t)t * NV, a + (size_t)t * NV, f ? f + (size_t)t * 12 : q, 1.0, 1.0, f ? 1.0 : 0.0,
           -1, -1, -1, -1, 0.0, 0.0, 0.0, 0.0, grav};
    rnea(e, Sink{tau + (size_t)t * NV, 1, 1.0, 0});
}


// =====================================================================================================================
// Kernel 2: Riccati sweep of the LQ problem of one real-time iteration.  One workgroup (4 wavefronts) per problem, all
// matrices of the current stage in LDS (float32), the dense products on the matrix cores:
//   v_mfma_f32_16x16x4_f32:  lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15]; it receives
//   C[4 (l >> 4) + r][l & 15], r = 0..3  (cdna_hip_programming.md, "A/B operands ... 16x16x4").
// Backward, stage k = N-1 .. 0 (P, p: cost-to-go 1/2 dx' P dx + p' dx of stage k + 1), arranged around the inversion of Quu,
// which is ONE wavefront's dependent chain and the longest phase of a stage:
//   1. PB = P B (6 tiles), s = P d + p                                                    | barrier
//   2. Quu = B' PB (3 upper-triangular tiles, mirrored)                                   | barrier
//   3. wavefront 0: Quu + R, Quu^-1 (2 x 2 blocks of 16: two in-register Gauss-Jordan eliminations on 64 lanes, five
//      16^3 products; wave_linalg.h)
//      wavefront w = 1..3 owns column block j = w - 1: PA(:, j) = P A(:, j) -> Qux(:, j) = B' PA(:, j) and two tiles of
//      Qxx(., j) = A' PA(:, j), as one chain of matrix instructions: an accumulator tile is the next product's B operand
//      (mfma_acc_regb); PA and Qxx stay in registers; qx, qu                              | barrier
//   4. wavefront w = 1..3, column block w - 1, again one chain in registers: K0 = -Quu^-1 Qux, R = Qux + Quu K0,
//      K = K0 - Quu^-1 R (one refinement step); wavefront 0 the same chain on [qu | 0] for kff, and the torque-limit
//      test (clamp + re-solve of the free inputs: back to 3.)                             | barrier
//   5. K (from the accumulators), kff -> HBM (forward sweep); P <- Q + Qxx + Qux' K, two tiles per column block, on the
//      wavefront that holds Qxx and K; p <- qx + Qux' kff on wavefront 0                  | barrier
// The next stage's A, B and right-hand sides are requested into registers during 3. / 4. and deposited when their buffers die.
// Forward: dx_0 = x0 - x_0, du_k = K_k dx_k + kff_k, dx_{k+1} = A_k dx_k + B_k du_k + d_k; then x += dx, u += du with
// the joint torques clipped to the URDF effort limits.
// =====================================================================================================================
constexpr int LDX = 49;  // row stride of the 48-column matrices in LDS (floats): 49 keeps column walks off one bank
constexpr int LDU = 33;  // row stride of the 32-column matrices

struct RicArgs {
    const float* A32;    // [B][N][48][48]
    const float* B32;    // [B][N][48][32]
    const double* next;  // [B][N][48]
    double* x;           // [B][N+1][48]  in/out
    double* u;           // [B][N][30]    in/out
    const double* x0;    // [B][48]
    const double* xref;  // [B][N+1][48]
    const double* uref;  // [B][N][30]
    const double* w;     // Q[48] R[30] QN[48]
    float* K;            // [B][N][32][48] workspace
    float* kff;          // [B][N][32]
    double* dx;          // [B][N+1][48]
    double* du;          // [B][N][30]
    int N;
    int apply;           // 1: x += dx, u += du (clipped)
    int* status;         // [B] 0 ok, 1 the step is not finite (indefinite Quu, overflow): x, u are then left untouched
    int limits;          // 1: torque limits inside the sweep (control-limited DDP); 0: only the applied inputs are clipped
    long long* stamps;   // optional [32] diagnostic
    const float* vec;    // [B][N][160] right-hand sides written by the stage kernel
    int cones = 0;       // 1: contact constraints on the 12 foot-force inputs inside the sweep (alore_wb.h)
    float mu = 0.f;      // friction coefficient of the pyramid
    const unsigned char* stance = nullptr; // [B][N][4] 1 = foot in contact at that stage; null = every foot, every stage
    const float* pen = nullptr;            // [B][N][PEN] contact-consistency penalty written by the stage kernel (StageArgs::pen)
    // refinement pass (alore_wb_set_refinement): the same LQ problem with the float64 residuals of a first solution as its
    // right-hand sides -- `vec` then holds [defect residual | stage gradient at the first solution | - | input gradient] as
    // they are (no weights applied, no penalty gradient added), resN the terminal gradient, the initial state step is zero
    int refine = 0;
    const float* resN = nullptr;           // [B][48]
    // exact working-set mode (alore_wb_set_constraint_mode): the sweep solves the REDUCED problem of the current working set --
    // B, vec, next are the reduced copies, the input weights differ per stage (a force component tied to a face of the friction
    // pyramid adds mu^2 R to the normal force it follows) and the input gradients arrive as they are
    const float* wrs = nullptr;            // [B][N][32] diagonal of R per stage, null = the shared weights w
    int gu_direct = 0;                     // 1: vec[128..159] is the input gradient itself, not (u - uref)
};

// 44.2 KB: three workgroups per CU (the register budget of the kernel asks for no more).  P A, Qxx, the gains K0 / R / K
// and Qux' K never exist in LDS: they are accumulator tiles that the next product takes as its B operand as they stand.
// Quu^-1 (and the 16 x 16 temporary of its block form) lives in PB, dead once Quu = B' PB exists.
struct RicLds {
    float P[48 * LDX], A[48 * LDX];
    float B[48 * LDU], PB[48 * LDU];
    float Qux[32 * LDX];
    float Quu[32 * LDU];
    float wq[48], wr[32];
    float p[48], s[48], d[48], gx[48], qx[48], dxk[48], dxn[48];
    float gu[32], qu[32], kff[32], duk[32];
};

// one 16 x 16 output tile: C[i0.., j0..] = scale * op(A) op(B) (+ Cinit) (+ alpha_diag * diag on the diagonal), K a
// multiple of 4 known at compile time: all operands of the tile are fetched from LDS first (2 K / 4 independent
// reads in flight), then the K / 4 matrix instructions run back to back.
//   TA: A is stored transposed (element (i, k) at A[k * lda + i]);  TB likewise for B
typedef float f4 __attribute__((ext_vector_type(4)));

template <bool TA, bool TB, int K>
__device__ __forceinline__ f4 mfma_acc(const float* A, int lda, const float* Bm, int ldb, int i0, int j0)
{
    int l = threadIdx.x & 63;
    asm volatile("" : "+v"(l)); // opaque: the addresses of a tile are recomputed per call instead of being kept in registers
                                  // across the whole stage loop (hoisted, they cost ~100 VGPRs and spill)
    const int r16 = l & 15, kq = l >> 4;
    float a[K / 4], b[K / 4];
#pragma unroll
    for (int s = 0; s < K / 4; ++s) {
        const int k = 4 * s + kq;
        a[s] = TA ? A[k * lda + i0 + r16] : A[(i0 + r16) * lda + k];
        b[s] = TB ? Bm[(j0 + r16) * ldb + k] : Bm[k * ldb + j0 + r16];
    }
    // two accumulation chains (even / odd K steps): a dependent v_mfma_f32_16x16x4_f32 waits 40 cycles for its predecessor
    f4 acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < K / 4; s += 2) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], acc, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s + 1], b[s + 1], acc1, 0, 0, 0);
    }
    return acc + acc1;
}

template <bool MIRROR>
__device__ __forceinline__ void tile_store(f4 acc, int i0, int j0, float* Cm, int ldc, const float* Cinit, int ldi, float alpha_diag,
                                           const float* diag, float scale = 1.f)
{
    int l = threadIdx.x & 63;
    asm volatile("" : "+v"(l));
    const int r16 = l & 15, kq = l >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = i0 + 4 * kq + r, col = j0 + r16;
        if (MIRROR && i0 == j0 && row > col) continue; // symmetric result: one triangle is computed and written twice
        float v = scale * acc[r];
        if (Cinit) v += Cinit[row * ldi + col];
        if (diag && row == col) v += alpha_diag * diag[row];
        Cm[row * ldc + col] = v;
        if (MIRROR && (i0 != j0 || row < col)) Cm[col * ldc + row] = v;
    }
}

template <bool TA, bool TB, int K, bool MIRROR = false>
__device__ __forceinline__ void mfma_tile(const float* A, int lda, const float* Bm, int ldb, int i0, int j0, float* Cm, int ldc,
                                          const float* Cinit, int ldi, float alpha_diag, const float* diag, float scale = 1.f)
{
    tile_store<MIRROR>(mfma_acc<TA, TB, K>(A, lda, Bm, ldb, i0, j0), i0, j0, Cm, ldc, Cinit, ldi, alpha_diag, diag, scale);
}

// acc += op(A)(i0 .. i0 + 15, k0 .. k0 + 15) X, with the 16 x 16 matrix X held in the C layout of a previous product (lane
// (c, q) register s = X[4 q + s][c]): an accumulator is the B operand of the next product WITHOUT leaving the registers.
// The sum over k may visit the 16 rows in any order as long as both operands agree, so k-slot (s, q) stands for row
// 4 q + s here (the layout the accumulator already has) instead of 4 s + q; only the A operand is fetched accordingly.
template <bool TA>
__device__ __forceinline__ f4 mfma_acc_regb(const float* A, int lda, int i0, int k0, f4 x, f4 acc)
{
    int l = threadIdx.x & 63;
    asm volatile("" : "+v"(l));
    const int r16 = l & 15, kq = l >> 4;
    float a[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int k = k0 + 4 * kq + s;
        a[s] = TA ? A[k * lda + i0 + r16] : A[(i0 + r16) * lda + k];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], x[s], acc, 0, 0, 0);
    return acc;
}
// the two halves of mfma_acc_regb: the A operand of a 16 x 16 block in the permuted k order, and its application
template <bool TA>
__device__ __forceinline__ void load_a4(const float* A, int lda, int i0, int k0, float (&a)[4])
{
    int l = threadIdx.x & 63;
    asm volatile("" : "+v"(l));
    const int r16 = l & 15, kq = l >> 4;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int k = k0 + 4 * kq + s;
        a[s] = TA ? A[k * lda + i0 + r16] : A[(i0 + r16) * lda + k];
    }
}
__device__ __forceinline__ f4 mfma4(const float (&a)[4], f4 x, f4 acc)
{
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], x[s], acc, 0, 0, 0);
    return acc;
}
// a 16 x 16 tile of an LDS matrix in the C layout
__device__ __forceinline__ f4 tile_load(const float* M, int ld, int i0, int j0)
{
    int l = threadIdx.x & 63;
    asm volatile("" : "+v"(l));
    const int r16 = l & 15, kq = l >> 4;
    f4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = M[(i0 + 4 * kq + r) * ld + j0 + r16];
    return v;
}

// upper-triangular 16 x 16 tiles of a symmetric 48 x 48 / 32 x 32 result
__device__ constexpr int SYM3_I[6] = {0, 0, 0, 1, 1, 2}, SYM3_J[6] = {0, 1, 2, 1, 2, 2}, SYM2_I[3] = {0, 0, 1}, SYM2_J[3] = {0, 1, 1};

constexpr int RIC_WAVES = 4, RIC_THREADS = 64 * RIC_WAVES, RIC_LAST = 64 * (RIC_WAVES - 1); // RIC_LAST: first thread of the last wave

template <bool PENALTY>
__global__ __launch_bounds__(RIC_THREADS, 3) void riccati_kernel(RicArgs g)
{
    extern __shared__ __align__(16) unsigned char smem[];
    RicLds& S = *reinterpret_cast<RicLds*>(smem);
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6;
    if (g.stamps && blockIdx.x == 0 && threadIdx.x == 0) g.stamps[31] = (long long)__builtin_readcyclecounter();
    const int N = g.N;
    const double *Qd = g.w, *Rd = g.w + NX, *QNd = g.w + NX + NU;
    const double* xb = g.x + (size_t)b * (N + 1) * NX;
    const double* ub = g.u + (size_t)b * N * NU;
    const double* xr = g.xref + (size_t)b * (N + 1) * NX;
    const double* ur = g.uref + (size_t)b * N * NU;
    // terminal cost
    for (int i = tid; i < 48 * 48; i += RIC_THREADS) { const int r = i / 48, c = i % 48; S.P[r * LDX + c] = (r == c) ? (float)QNd[r] : 0.f; }
    if (tid < 48) {
        S.p[tid] = g.refine ? g.resN[(size_t)b * NX + tid] : (float)(QNd[tid] * (xb[(size_t)N * NX + tid] - xr[(size_t)N * NX + tid]));
        S.wq[tid] = (float)Qd[tid];
    }
    else if (tid >= 64 && tid < 96) S.wr[tid - 64] = (tid - 64) < NU ? (float)Rd[tid - 64] : 1.f; // identity on the 2 padding inputs
    __syncthreads();

    // A stage's A, B (16-byte pieces) and right-hand sides travel global -> registers -> LDS and are requested AHEAD of
    // their use: A_{k-1} while Quu of stage k is inverted (S.A is dead when the inversion ends), B_{k-1} and the vectors
    // under the gain computation (S.B holds the refinement residual until the gains exist), so that no HBM round trip
    // stands in front of a stage.  Every thread loads with clamped indices (no divergent register state).
    const int vi = tid < VEC ? tid : VEC - 1;
    const int ia0 = tid, ia1 = tid + RIC_THREADS, ia2 = (tid + 2 * RIC_THREADS) < 48 * 12 ? tid + 2 * RIC_THREADS : 48 * 12 - 1;
    const int ib0 = tid, ib1 = (tid + RIC_THREADS) < 48 * 8 ? tid + RIC_THREADS : 48 * 8 - 1;
    float4 pa0, pa1, pa2, pb0, pb1;
    float pvec, ppen = 0.f, pwr = 1.f;
#define RIC_REQUEST_A(kk)                                                                                    \
    {                                                                                                        \
        const float4* Ag_ = reinterpret_cast<const float4*>(g.A32 + ((size_t)b * N + (kk)) * NX * NX);       \
        pa0 = Ag_[ia0]; pa1 = Ag_[ia1]; pa2 = Ag_[ia2];                                                      \
    }
#define RIC_REQUEST_B(kk)                                                                                    \
    {                                                                                                        \
        const float4* Bg_ = reinterpret_cast<const float4*>(g.B32 + ((size_t)b * N + (kk)) * NX * NUP);      \
        pb0 = Bg_[ib0]; pb1 = Bg_[ib1];                                                                      \
        pvec = g.vec[((size_t)b * N + (kk)) * VEC + vi];                                                     \
        if (PENALTY && !g.refine && tid >= 72 && tid < 96) ppen = g.pen[((size_t)b * N + (kk)) * PEN + 288 + tid - 72]; \
        if (g.wrs && tid >= 128 && tid < 160) pwr = g.wrs[((size_t)b * N + (kk)) * 32 + tid - 128];           \
    }
#define RIC_PUT4(base, ld, per_row, idx, v)                                                                  \
    { float* dst_ = (base) + ((idx) / (per_row)) * (ld) + 4 * ((idx) % (per_row)); dst_[0] = (v).x; dst_[1] = (v).y; dst_[2] = (v).z; dst_[3] = (v).w; }
#define RIC_DEPOSIT_A()                                                                                      \
    {                                                                                                        \
        RIC_PUT4(S.A, LDX, 12, ia0, pa0) RIC_PUT4(S.A, LDX, 12, ia1, pa1)                                    \
        if (tid + 2 * RIC_THREADS < 48 * 12) RIC_PUT4(S.A, LDX, 12, ia2, pa2)                                \
    }
#define RIC_DEPOSIT_B()                                                                                      \
    {                                                                                                        \
        RIC_PUT4(S.B, LDU, 8, ib0, pb0)                                                                      \
        if (tid + RIC_THREADS < 48 * 8) RIC_PUT4(S.B, LDU, 8, ib1, pb1)                                      \
        /* d | gx = Q (x - xref) | current input (for the torque limits; dxn is free during the backward sweep) | gu = R (u - uref) */ \
        if (tid < 48) S.d[tid] = pvec;                                                                       \
        else if (tid < 96) S.gx[tid - 48] = g.refine ? pvec : S.wq[tid - 48] * pvec + ppen; /* + rho J_c' (J_c v) on the velocities */ \
        else if (tid < 128) S.dxn[tid - 96] = pvec;                                                          \
        else if (tid < 160) {                                                                                \
            if (g.wrs) S.wr[tid - 128] = pwr; /* read by the inverting wavefront after the next barrier */       \
            S.gu[tid - 128] = (g.refine || g.gu_direct) ? pvec : S.wr[tid - 128] * pvec;                       \
        }                                                                                                    \
    }
    RIC_REQUEST_A(N - 1)
    RIC_REQUEST_B(N - 1)
    RIC_DEPOSIT_A()
    RIC_DEPOSIT_B()
    __syncthreads();

    for (int k = N - 1; k >= 0; --k) {
        WB_STAMP(g.stamps, 0)
        // ---- PB = P B (6 tiles), s = P d + p
        for (int t = wave; t < 6; t += RIC_WAVES)
            mfma_tile<false, false, 48>(S.P, LDX, S.B, LDU, (t / 2) * 16, (t % 2) * 16, S.PB, LDU, nullptr, 0, 0.f, nullptr);
        if (tid >= RIC_LAST && tid < RIC_LAST + 48) { // the last wave has one tile only
            const int i = tid - RIC_LAST;
            float acc = S.p[i];
            for (int j = 0; j < 48; ++j) acc += S.P[i * LDX + j] * S.d[j];
            S.s[i] = acc;
        }
        __syncthreads();
        WB_STAMP(g.stamps, 1)
        // ---- Quu = B' PB (3 upper-triangular tiles, mirrored; R is added by the inverting wavefront)
        if (wave != 0)
            mfma_tile<true, false, 48, true>(S.B, LDU, S.PB, LDU, SYM2_I[wave - 1] * 16, SYM2_J[wave - 1] * 16, S.Quu, LDU, nullptr, 0, 0.f, nullptr);
        __syncthreads();
        WB_STAMP(g.stamps, 2)
        // ---- Wavefront 0 inverts Quu (the first pass of the loop below: one wavefront's dependent chain, the longest
        //      phase of a stage) WHILE wavefronts 1 .. 3 do everything that hangs on A: wavefront w owns COLUMN BLOCK
        //      j = w - 1 -- PA(:, j) = P A(:, j) (3 tiles), Qux(:, j) = B' PA(:, j) (2), Qxx(i <= j, j) = A(:, i)' PA(:, j)
        //      (j + 1 upper-triangular tiles) all read only the wavefront's own block of PA, so no barrier separates
        //      them.  The Qxx tiles stay in registers until the end of the stage, where the same wavefront adds Qux' K.  qx = gx + A' s, qu = gu + B' s on the wavefront with fewest tiles.
        const int role = wave; // 0 inverts, 1 .. 3 multiply (rotating the role over the wavefronts / SIMDs was measured: no effect)
        const int wl = tid & 63;
        f4 qxx[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}; // Qxx tiles (j, j) and ((j + 2) % 3, j) of the wavefront's column block j
        f4 kt[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}; // the wavefront's column block of the gain K (rows 0..15, 16..31), C layout
        f4 quxt[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}; // ... and of Qux
        if (role != 0) {
            const int j0 = (role - 1) * 16;
            // PA(:, j) stays in registers (three accumulator tiles) and is the B operand of the products below as it stands
            f4 pa[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) pa[t] = mfma_acc<false, false, 48>(S.P, LDX, S.A, LDX, t * 16, j0);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 3; ++t) acc = mfma_acc_regb<true>(S.B, LDU, i * 16, 16 * t, pa[t], acc);
                quxt[i] = acc;
                tile_store<false>(acc, i * 16, j0, S.Qux, LDX, nullptr, 0, 0.f, nullptr); // the cost-to-go tiles of the other wavefronts read it
            }
            // the symmetric Qxx as 6 tiles, two per column block: (j, j) and ((j + 2) % 3, j) -- (0,0) (2,0) | (1,1) (0,1) | (2,2) (1,2)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int i0 = n == 0 ? j0 : ((role + 1) % 3) * 16;
                f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 3; ++t) acc = mfma_acc_regb<true>(S.A, LDX, i0, 16 * t, pa[t], acc);
                qxx[n] = acc;
            }
            const int l = wl;
            if (role == 1 && l < 48) {
                float acc = S.gx[l];
                for (int j = 0; j < 48; ++j) acc += S.A[j * LDX + l] * S.s[j];
                S.qx[l] = acc;
            }
            if (role == 1 && l < 32) {
                float acc = S.gu[l];
                for (int j = 0; j < 48; ++j) acc += S.B[j * LDU + l] * S.s[j];
                S.qu[l] = acc;
            }
        }
        if (k > 0) RIC_REQUEST_A(k - 1) // lands while Quu is inverted
        float* Qinv = S.PB;   // PB is dead once Quu = B' PB exists
        // clamp flags / values of the control limits live in dxk / s (free during the backward sweep / dead after qx, qu):
        // the LDS block must not grow -- 53.6 KB is the last size of which three fit a CU at the hardware's allocation
        // granularity (one more 256 B and only two workgroups are resident: measured 2.8 -> 4.2 ms)
        float* clampm = S.dxk;
        float* clampv = S.s;
        // Round 0 solves the unconstrained stage problem.  If its feed-forward step drives a joint torque past the URDF
        // effort limit, those inputs are clamped to the limit and the free ones are re-solved against them
        // (control-limited DDP, Tassa et al. 2014, one projection): in Quu the clamped rows / columns become an
        // identity block, their right-hand side the clamped value, and their gain rows are zeroed afterwards.  P' and p'
        // keep their form (Qxx + Qux' K, qx + Qux' kff) because (qu + Quu kff) and (Qux + Quu K) vanish on the free rows.
#pragma unroll 1
        for (int round = 0; round < 2; ++round) { // kept rolled: the inversion is ~2000 unrolled instructions
            // ---- Quu^-1 (wave 0: lane i owns row i, Gauss-Jordan in registers; wave_linalg.h) -> Qinv (in the PB buffer)
            if (role == 0) {
                // diagonal of Quu: + R on the 30 real inputs, identity on the 2 padding rows (the tiles above wrote B' P B);
                // LDS operations of one wavefront complete in order, so the row loads below see it
                if (round == 0 && wl < 32) S.Quu[wl * LDU + wl] += S.wr[wl];
                __builtin_amdgcn_wave_barrier();
                // 2 x 2 blocks of 16 (the 2 padding inputs are an identity block inside the second one, so all 32 rows go
                // through): Quu = [A11 A12; A12' A22],  T = A11^-1 A12,  S = A22 - A12' T,
                //   Quu^-1 = [A11^-1 + T S^-1 T', -T S^-1; -S^-1 T', S^-1].
                // The two 16 x 16 inverses are in-register Gauss-Jordan eliminations (16 steps of 16 v_readlane + 16 FMA
                // instead of 30 x (30 + 30): 12.8 k cycles against 15.4 k for the one-piece elimination); the five
                // 16 x 16 x 16 products run on the matrix cores of the same wavefront, through the spare 16 rows of the PB buffer.
                float* T = Qinv + 32 * LDU;               // 16 x 16, stride LDU (PB has 48 rows, Qinv uses 32)
                float* Q22 = Qinv + 16 * LDU + 16;
                const int rr = wl & 15;
                const int qc = 4 * (wl >> 4); // lane 16 p + r: columns 4 p .. 4 p + 3 of row r
                {
                    float rq[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) rq[j] = S.Quu[rr * LDU + qc + j];
                    wavela::spd_inverse16_quarters(rq, wl);
#pragma unroll
                    for (int j = 0; j < 4; ++j) Qinv[rr * LDU + qc + j] = rq[j];
                }
                mfma_tile<false, false, 16>(Qinv, LDU, S.Quu + 16, LDU, 0, 0, T, LDU, nullptr, 0, 0.f, nullptr);                      // T = A11^-1 A12
                mfma_tile<false, false, 16>(S.Quu + 16 * LDU, LDU, T, LDU, 0, 0, Q22, LDU, S.Quu + 16 * LDU + 16, LDU, 0.f, nullptr, -1.f); // S = A22 - A21 T
                {
                    float rq[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) rq[j] = Q22[rr * LDU + qc + j];
                    wavela::spd_inverse16_quarters(rq, wl);
#pragma unroll
                    for (int j = 0; j < 4; ++j) Q22[rr * LDU + qc + j] = rq[j];
                }
                mfma_tile<false, false, 16>(T, LDU, Q22, LDU, 0, 0, Qinv + 16, LDU, nullptr, 0, 0.f, nullptr, -1.f);                 // -T S^-1
                mfma_tile<false, true, 16>(Q22, LDU, T, LDU, 0, 0, Qinv + 16 * LDU, LDU, nullptr, 0, 0.f, nullptr, -1.f);            // -S^-1 T'
                mfma_tile<false, true, 16>(Qinv + 16, LDU, T, LDU, 0, 0, Qinv, LDU, Qinv, LDU, 0.f, nullptr, -1.f);                  // A11^-1 + T S^-1 T'
            }
            WB_STAMP(g.stamps, 3)
            __syncthreads();
            WB_STAMP(g.stamps, 4)
            if (round == 0) {
                if (k > 0) { RIC_DEPOSIT_A() RIC_REQUEST_B(k - 1) } // S.A is dead; B_{k-1} and the vectors land under the gain computation
            }
            // ---- K0 = -Qinv Qux, then one refinement step against Quu itself (the explicit float32 inverse alone costs a
            //      factor 40 in accuracy): R = Qux + Quu K0, K = K0 - Qinv R.  A COLUMN BLOCK of K needs only the same
            //      column block of K0 and R, and an accumulator tile is the B operand of the next product as it stands
            //      (mfma_acc_regb): wavefront w = 1..3 runs the three products of block w - 1 in registers, 48 matrix
            //      instructions fed by the A operands only; wavefront 0 does the same for the vector:
            //      kff0 = -Qinv qu, r = qu + Quu kff0, kff = kff0 - Qinv r
            {
                // wavefront 0 runs the same chain on the 32 x 16 "block" [qu | 0 ... 0]: column 0 of its result is kff (the
                // matrix cores of its SIMD are idle, and three 32-term dot products per lane out of LDS took twice as long)
                const int c16 = wl & 15, q4 = wl >> 4;
                f4 in[2];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) in[t][r] = role != 0 ? quxt[t][r] : (c16 == 0 ? S.qu[16 * t + 4 * q4 + r] : 0.f);
                // all A operands first (Quu^-1 serves two of the three products): one LDS latency in front of 48 chained
                // matrix instructions instead of one in front of each product
                float aq[2][2][4], au[2][2][4];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int t = 0; t < 2; ++t) { load_a4<false>(Qinv, LDU, i * 16, 16 * t, aq[i][t]); load_a4<false>(S.Quu, LDU, i * 16, 16 * t, au[i][t]); }
                f4 k0t[2], rt[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc = mfma4(aq[i][t], in[t], acc);
                    k0t[i] = -acc;
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    f4 acc = in[i];
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc = mfma4(au[i][t], k0t[t], acc);
                    rt[i] = acc;
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc = mfma4(aq[i][t], rt[t], acc);
                    kt[i] = k0t[i] - acc;
                }
                if (role == 0) {
                    if (c16 == 0) {
#pragma unroll
                        for (int t = 0; t < 2; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r) S.kff[16 * t + 4 * q4 + r] = kt[t][r];
                    }
                    __builtin_amdgcn_wave_barrier(); // one wavefront: its LDS operations complete in order
                    if (round == 0 && wl < 32) { // torque limits on the feed-forward step of this stage: flags / values for the pass below
                        const int i = wl;
                        float m = 0.f;
                        if (g.limits && i < b2z1::NJ) {
                            const float kf = S.kff[i], ucur = S.dxn[i], eff = (float)b2z1::EFFORT[i];
                            const float lo = -eff - ucur, hi = eff - ucur;
                            if (kf < lo - 1e-4f * eff) { m = 1.f; clampv[i] = lo; }
                            else if (kf > hi + 1e-4f * eff) { m = 1.f; clampv[i] = hi; }
                        } else if (g.cones && i >= b2z1::NJ && i < NU) {
                            // contact constraints of the foot-force inputs (world frame, flat ground), projected like the
                            // torque boxes: a foot in the air carries no force; a stance foot pushes (fz >= 0) inside the
                            // friction pyramid |fx|, |fy| <= mu fz, with fz the projected normal force of this stage
                            const int foot = (i - b2z1::NJ) / 3, ax = (i - b2z1::NJ) % 3, iz = b2z1::NJ + 3 * foot + 2;
                            const bool st = g.stance ? g.stance[((size_t)b * N + k) * 4 + foot] != 0 : true;
                            const float kf = S.kff[i], ucur = S.dxn[i];
                            const float fz = st ? fmaxf(S.dxn[iz] + S.kff[iz], 0.f) : 0.f;
                            const float bnd = (ax == 2) ? 3.0e38f : g.mu * fz;
                            const float lo = ((ax == 2 || !st) ? 0.f : -bnd) - ucur, hi = (st ? bnd : 0.f) - ucur;
                            const float tol = 1e-4f * fmaxf(1.f, g.mu * fz);
                            if (kf < lo - tol) { m = 1.f; clampv[i] = lo; }
                            else if (kf > hi + tol) { m = 1.f; clampv[i] = hi; }
                        }
                        clampm[i] = m;
                    }
                }
            }
            __syncthreads();
            WB_STAMP(g.stamps, 7)
            if (round == 1) break;
            // "any input clamped?" without __syncthreads_or (its LDS temporary pushes the block past the size of which
            // three fit a CU): every wavefront ballots the 18 flags itself
            if ((!g.limits && !g.cones) || !__any(((tid & 63) < NU) && clampm[tid & 63] != 0.f)) break;
            // masked system, in place: qu first (it needs the unmasked rows of Quu), then Quu
            if (tid < 32) {
                float acc = S.qu[tid];
                for (int j = 0; j < NU; ++j) acc += clampm[j] != 0.f ? S.Quu[tid * LDU + j] * clampv[j] : 0.f;
                S.duk[tid] = clampm[tid] != 0.f ? -clampv[tid] : acc;
            }
            __syncthreads();
            if (tid < 32) S.qu[tid] = S.duk[tid];
            for (int i = tid; i < 32 * 32; i += RIC_THREADS) {
                const int r = i >> 5, c = i & 31;
                if (clampm[r] != 0.f || clampm[c] != 0.f) S.Quu[r * LDU + c] = (r == c) ? 1.f : 0.f;
            }
            __syncthreads();
        }
        const bool any_clamp = __any(((tid & 63) < NU) && clampm[tid & 63] != 0.f);
        if (k > 0) RIC_DEPOSIT_B() // S.B and the stage's vectors are dead; read again after the barrier below
        // ---- K, kff -> HBM (forward sweep); P <- Q + Qxx + Qux' K: the 6 upper-triangular tiles, each on the wavefront that
        //      holds its Qxx tile AND its column block of K in registers (neither goes through LDS), mirrored on store;
        //      p <- qx + Qux' kff on wavefront 0
        if (role != 0) {
            const int j0 = (role - 1) * 16, c16 = wl & 15, q4 = wl >> 4;
            float* Kg = g.K + ((size_t)b * N + k) * 32 * 48;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * t + 4 * q4 + r;
                    if (any_clamp && clampm[row] != 0.f) kt[t][r] = 0.f; // clamped inputs do not react to dx
                    Kg[row * 48 + j0 + c16] = kt[t][r];
                }
            // contact-consistency penalty: + rho J_c' J_c on the velocity block (state indices 24 .. 47), i.e. on the tiles
            // (1,1) (1,2) (2,2).  The operands come straight from HBM into the lanes: lane (c16, q4) supplies
            // Jext[4 t + q4][16 m + c16], Jext = [0 (12 x 24) | sqrt(rho) J_c], for column blocks m = 1, 2
            float ja[3] = {0.f, 0.f, 0.f}, jb[3] = {0.f, 0.f, 0.f};
            if (PENALTY && role >= 2) {
                const float* Jg = g.pen + ((size_t)b * N + k) * PEN;
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const int kk = 4 * t + q4;
                    ja[t] = (c16 >= 8) ? Jg[kk * 24 + c16 - 8] : 0.f;
                    jb[t] = Jg[kk * 24 + 8 + c16];
                }
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int i0 = n == 0 ? j0 : ((role + 1) % 3) * 16;
                f4 acc = qxx[n];
#pragma unroll
                for (int t = 0; t < 2; ++t) acc = mfma_acc_regb<true>(S.Qux, LDX, i0, 16 * t, kt[t], acc);
                if (PENALTY && role >= 2 && (n == 0 || role == 3)) { // role 2: (1,1); role 3: (2,2) and (1,2)
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        const float av = (role == 2 || n == 1) ? ja[t] : jb[t]; // rows: block 1 for (1,1) and (1,2), block 2 for (2,2)
                        const float bv = (role == 2) ? ja[t] : jb[t];          // columns: the wavefront's own block
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
                    }
                }
                tile_store<true>(acc, i0, j0, S.P, LDX, nullptr, 0, 1.f, S.wq);
            }
        } else {
            if (wl < 32) g.kff[((size_t)b * N + k) * 32 + wl] = S.kff[wl];
            // p <- qx + Qux' kff: column 0 of Qux' [kff | 0 ... 0], the wavefront's own tiles again
            const int c16 = wl & 15, q4 = wl >> 4;
#pragma unroll
            for (int ib = 0; ib < 3; ++ib) {
                f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 2; ++t) acc = mfma_acc_regb<true>(S.Qux, LDX, ib * 16, 16 * t, kt[t], acc);
                if (c16 == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) S.p[ib * 16 + 4 * q4 + r] = S.qx[ib * 16 + 4 * q4 + r] + acc[r];
                }
            }
        }
        __syncthreads();
        WB_STAMP(g.stamps, 8)
        WB_STAMP(g.stamps, 9)
    }

    // ---- forward sweep
    double* dxb = g.dx + (size_t)b * (N + 1) * NX;
    double* dub = g.du + (size_t)b * N * NU;
    if (tid < 48) { S.dxk[tid] = g.refine ? 0.f : (float)(g.x0[(size_t)b * NX + tid] - xb[tid]); dxb[tid] = S.dxk[tid]; }
    __syncthreads();
    // row r of a product on the four lanes of quad r (thread 4 r + part): each lane reads a contiguous quarter of the
    // matrix row with 16-byte loads and the quad adds up over DPP -- 4 x fewer dependent loads per lane than one lane
    // per row, and the four lanes of a quad read one 192-byte run
    const int qrow = tid >> 2, part = tid & 3;
    auto quad_total = [](float v) {
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)); // quad_perm [1,0,3,2]
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true)); // quad_perm [2,3,0,1]
        return v;
    };
    // The rows of K_k, A_k, B_k, kff_k and the defect of a stage are requested one stage ahead (registers), so that the HBM /
    // L2 round trip of stage k + 1 runs under the two products of stage k; dx alternates between two LDS vectors.
    const int krow = qrow < 32 ? qrow : 31, arow = qrow < 48 ? qrow : 47;
    float4 kn[3], an[3], bn[2];
    float kffn = 0.f;
    double nxn = 0.0, xnn = 0.0;
#define RIC_FWD_REQUEST(kk)                                                                                  \
    {                                                                                                        \
        const float4* kr_ = reinterpret_cast<const float4*>(g.K + ((size_t)b * N + (kk)) * 32 * 48 + krow * 48 + part * 12);     \
        const float4* ar_ = reinterpret_cast<const float4*>(g.A32 + ((size_t)b * N + (kk)) * NX * NX + arow * 48 + part * 12);   \
        const float4* br_ = reinterpret_cast<const float4*>(g.B32 + ((size_t)b * N + (kk)) * NX * NUP + arow * NUP + part * 8);  \
        kn[0] = kr_[0]; kn[1] = kr_[1]; kn[2] = kr_[2];                                                      \
        an[0] = ar_[0]; an[1] = ar_[1]; an[2] = ar_[2];                                                      \
        bn[0] = br_[0]; bn[1] = br_[1];                                                                      \
        kffn = g.kff[((size_t)b * N + (kk)) * 32 + krow];                                                    \
        if (g.refine) { nxn = (double)g.vec[((size_t)b * N + (kk)) * VEC + arow]; xnn = 0.0; }               \
        else { nxn = g.next[((size_t)b * N + (kk)) * NX + arow]; xnn = xb[(size_t)((kk) + 1) * NX + arow]; } \
    }
    RIC_FWD_REQUEST(0)
    float* dxc = S.dxk;  // dx_k
    float* dxw = S.dxn;  // dx_{k+1}
    for (int k = 0; k < N; ++k) {
        float4 kc[3] = {kn[0], kn[1], kn[2]}, ac[3] = {an[0], an[1], an[2]}, bc[2] = {bn[0], bn[1]};
        const float kffc = kffn;
        const double nxc = nxn, xnc = xnn;
        if (k + 1 < N) RIC_FWD_REQUEST(k + 1)
        {   // du_k = K dx + kff: rows 0..31 on threads 0..127
            float acc = 0.f;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const float* dxp = dxc + part * 12 + 4 * q;
                acc += kc[q].x * dxp[0] + kc[q].y * dxp[1] + kc[q].z * dxp[2] + kc[q].w * dxp[3];
            }
            acc = quad_total(acc);
            if (qrow < 32 && part == 0) {
                acc += kffc;
                S.duk[qrow] = acc;
                if (qrow < NU) dub[(size_t)k * NU + qrow] = acc;
            }
        }
        __syncthreads();
        {   // dx_{k+1} = A dx + B du + d: rows 0..47 on threads 0..191
            float acc = 0.f;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const float* dxp = dxc + part * 12 + 4 * q;
                acc += ac[q].x * dxp[0] + ac[q].y * dxp[1] + ac[q].z * dxp[2] + ac[q].w * dxp[3];
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) { // the two padding columns of B are zero and duk[30..31] = 0
                const float* dup = S.duk + part * 8 + 4 * q;
                acc += bc[q].x * dup[0] + bc[q].y * dup[1] + bc[q].z * dup[2] + bc[q].w * dup[3];
            }
            acc = quad_total(acc);
            if (qrow < 48 && part == 0) {
                acc += (float)(nxc - xnc);
                dxw[qrow] = acc;
                dxb[(size_t)(k + 1) * NX + qrow] = acc;
            }
        }
        __syncthreads();
        float* t_ = dxc; dxc = dxw; dxw = t_;
    }
    WB_STAMP(g.stamps, 10)
    // a step that is not finite (an indefinite Quu from negative weights, an overflow) must not touch the iterate
    int bad = 0;
    for (int i = tid; i < (N + 1) * NX; i += RIC_THREADS) bad |= !isfinite(dxb[i]);
    for (int i = tid; i < N * NU; i += RIC_THREADS) bad |= !isfinite(dub[i]);
    __syncthreads(); // the dx / du written by other threads above are visible (same workgroup, global memory)
    S.qu[tid & 31] = 0.f;
    __syncthreads();
    if (bad) S.qu[0] = 1.f;
    __syncthreads();
    const bool failed = S.qu[0] != 0.f;
    if (tid == 0 && g.status) g.status[b] = failed ? 1 : 0;
    if (g.apply && !failed) {
        double* xw = g.x + (size_t)b * (N + 1) * NX;
        double* uw = g.u + (size_t)b * N * NU;
        for (int i = tid; i < (N + 1) * NX; i += RIC_THREADS) xw[i] += dxb[i];
        for (int i = tid; i < N * NU; i += RIC_THREADS) {
            const int j = i % NU;
            if (g.cones && j >= b2z1::NJ) continue; // the foot forces: below, one thread per (stage, foot)
            double val = uw[i] + dub[i];
            if (j < b2z1::NJ) { const double lim = b2z1::EFFORT[j]; val = val > lim ? lim : (val < -lim ? -lim : val); }
            uw[i] = val;
        }
        if (g.cones) { // the applied forces satisfy the contact constraints exactly
            for (int it = tid; it < N * 4; it += RIC_THREADS) {
                const int kk = it >> 2, foot = it & 3, i0 = kk * NU + b2z1::NJ + 3 * foot;
                const bool st = g.stance ? g.stance[((size_t)b * N + kk) * 4 + foot] != 0 : true;
                const double fzr = uw[i0 + 2] + dub[i0 + 2], fz = st ? (fzr > 0.0 ? fzr : 0.0) : 0.0, lim = (double)g.mu * fz;
                double fx = uw[i0] + dub[i0], fy = uw[i0 + 1] + dub[i0 + 1];
                fx = fx > lim ? lim : (fx < -lim ? -lim : fx);
                fy = fy > lim ? lim : (fy < -lim ? -lim : fy);
                uw[i0] = fx; uw[i0 + 1] = fy; uw[i0 + 2] = fz;
            }
        }
    }
}

// ---- hard contact rows (alore_wb_set_contact_rows) ------------------------------------------------------------------------
// Stance feet do not move: J_c(q_k) v_{k+1} = 0 for every foot in contact at stage k, as EQUALITY rows of the LQ problem
// (velocity level: with the semi-implicit Euler step this is J_c qdd + J_c v / dt = 0, the acceleration row with its
// stabilisation term).  In the linearised dynamics the row reads
//     J_c [A dx + B du + next]_v = 0,     [.]_v = the 24 velocity components,
// and its block with respect to the 12 foot forces, G = J_c (dt M^-1 J_c') -- the velocity rows of the force columns of B --
// is symmetric positive definite: the rows are solved for the foot forces,
//     df = F_x dx + F_tau dtau + f0,      [F_x | F_tau | f0] = -G^-1 [J_c A_v | J_c B_v,tau | J_c next_v],
// and the forces leave the problem (the null space of the rows is parametrised by the joint torques): A <- A + B_f F_x,
// B_tau <- B_tau + B_f F_tau, B_f <- 0, next <- next + B_f f0.  The Riccati kernel then sweeps an ordinary LQ problem in
// (dx, dtau) -- the foot forces carry no cost of their own in this mode, they are what the contacts need -- and the forces
// follow from the solved dx, dtau afterwards.  A foot in the air (contact schedule) has the row f = 0 instead.
// One wavefront per (problem, stage), float64; G^-1 by Gauss-Jordan on the 12 x 79 augmented block in LDS.
constexpr int FCOL = 68; // floats per force row of the stored map: F_x (48) | F_tau (18) | f0 | pad
__device__ long long g_rows_stamps[8]; // diagnostic: cycles per phase of workgroup 0 of the last launch (tools/wb_rows_time.py reads it)
// ... and a scheduling fence: without it the loads of a later phase are hoisted over the earlier ones (338 registers, one wavefront per SIMD)
#define ROWS_STAMP(i) __builtin_amdgcn_sched_barrier(0); if (blockIdx.x == 0 && threadIdx.x == 0) g_rows_stamps[i] = (long long)__builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0);
__global__ __launch_bounds__(64) void contact_rows_kernel(float* A32, float* B32, double* next, float* vec, const float* pen, const unsigned char* stance,
                                                             const double* u, int N, float* Fg)
{
    constexpr int MC = 80; // G (12) | J A_v (48) | J B_v,tau (18) | J next_v (1) | pad
    __shared__ double Mx[12 * MC];
    __shared__ double Bf[NX * 12];
    __shared__ double colp[144]; // the twelve pivot columns of the elimination
    const int item = blockIdx.x, lane = threadIdx.x;
    float* A = A32 + (size_t)item * NX * NX;
    float* Bm = B32 + (size_t)item * NX * NUP;
    double* nx = next + (size_t)item * NX;
    const float* J = pen + (size_t)item * PEN; // written with rho = 1: J_c itself, rows of swing feet zero
    const unsigned char* st = stance ? stance + (size_t)item * 4 : nullptr;
    ROWS_STAMP(0)
    // every global read of the set-up is issued before the first LDS store (a rolled loop waits for one load per trip)
    float jav[6];
    {
        float bv[9];
#pragma unroll
        for (int t = 0; t < 6; ++t) jav[t] = (lane & 15) < 12 ? J[(lane & 15) * NV + 4 * t + (lane >> 4)] : 0.f; // J_c as the matrix cores' A operand, straight from global memory (2.3 KB of LDS less: twelve workgroups per CU)
#pragma unroll
        for (int q = 0; q < 9; ++q) { const int e = lane + 64 * q; bv[q] = Bm[(e / 12) * NUP + b2z1::NJ + e % 12]; } // 576 = 9 x 64
#pragma unroll
        for (int q = 0; q < 9; ++q) { const int e = lane + 64 * q; Bf[e] = (double)bv[q]; }
    }
    __syncthreads();
    ROWS_STAMP(1)
    // [G | J A_v | J B_v,tau | J next_v] = J_c [B_f | A | B_tau | next]_v on the float64 matrix cores (v_mfma_f64_16x16x4_f64: lane l supplies
    // A[l & 15][l >> 4] and B[l >> 4][l & 15], receives C[(l >> 4) + 4 r][l & 15]): J_c padded to 16 rows, 5 column tiles x 6 k-steps
    typedef double d4 __attribute__((ext_vector_type(4)));
    const int r16 = lane & 15, kq = lane >> 4;
    {
        double ja[6];
#pragma unroll
        for (int t = 0; t < 6; ++t) ja[t] = (double)jav[t];
#pragma unroll
        for (int tj = 0; tj < 5; ++tj) {
            // the B operand -- the velocity rows of [B_f | A | B_tau | next | 0], column 16 tj + r16 -- straight from where it lives: B_f from
            // LDS, A and B_tau from global memory (16 consecutive floats per row piece), next from its float64 array
            const int col = 16 * tj + r16;
            const float* fsrc = col < 60 ? A + NQ * NX + (col >= 12 ? col - 12 : 0) : Bm + NQ * NUP + (col < 78 ? col - 60 : 0);
            const int fld = col < 60 ? NX : NUP;
            double sb[6];
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                const int row = 4 * t + kq;
                const float fv = fsrc[row * fld];
                sb[t] = col < 12 ? Bf[(NQ + row) * 12 + col] : (col < 78 ? (double)fv : (col == 78 ? nx[NQ + row] : 0.0));
            }
            d4 c = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int t = 0; t < 6; ++t) c = __builtin_amdgcn_mfma_f64_16x16x4f64(ja[t], sb[t], c, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 3; ++r) Mx[(kq + 4 * r) * MC + 16 * tj + r16] = c[r]; // rows 0 .. 11
        }
    }
    __syncthreads();
    // a lane owns columns `lane` and (lanes 0 .. 14) `lane + 64` of the 12 x 79 block, in REGISTERS through the elimination
    double m0[12], m1[12];
    const int col1 = lane + 64 < 79 ? lane + 64 : 78;
#pragma unroll
    for (int r = 0; r < 12; ++r) { m0[r] = Mx[r * MC + lane]; m1[r] = Mx[r * MC + col1]; }
#pragma unroll
    for (int r = 0; r < 12; ++r) {
        if (st && !st[r / 3]) { // foot in the air: the row  df_r = -f_r
            m0[r] = (lane < 12) ? ((lane == r) ? 1.0 : 0.0) : 0.0;
            m1[r] = (col1 == 78) ? u[(size_t)item * NU + b2z1::NJ + r] : 0.0;
        }
    }
    __syncthreads(); // Mx is written again after the elimination
    ROWS_STAMP(2)
    // Gauss-Jordan without pivoting (G is symmetric positive definite) on the register columns: lane p hands column p of G round
    // through LDS and every lane updates its first column; the twelve pivot columns stay in LDS and the second column (lanes 0 .. 14)
    // goes through the same twelve steps afterwards, without barriers (both columns in one unrolled loop cost 338 registers: one
    // wavefront per SIMD)
#pragma unroll
    for (int p = 0; p < 12; ++p) {
        double* cp = colp + 12 * p;
        if (lane == p) {
#pragma unroll
            for (int r = 0; r < 12; ++r) cp[r] = m0[r];
        }
        __syncthreads();
        const double cpp = cp[p];
        double inv = __builtin_amdgcn_rcp(cpp); // hardware reciprocal + two Newton steps: full float64 accuracy at a fifth of the division's instructions
        inv = inv * (2.0 - cpp * inv); inv = inv * (2.0 - cpp * inv);
        const double pv0 = m0[p] * inv;
#pragma unroll
        for (int r = 0; r < 12; ++r) m0[r] = (r == p) ? pv0 : m0[r] - cp[r] * pv0;
    }
    __builtin_amdgcn_sched_barrier(0);
    if (lane + 64 < 79) {
#pragma unroll
        for (int p = 0; p < 12; ++p) {
            const double* cp = colp + 12 * p;
            const double cpp = cp[p];
            double inv = __builtin_amdgcn_rcp(cpp);
            inv = inv * (2.0 - cpp * inv); inv = inv * (2.0 - cpp * inv);
            const double pv1 = m1[p] * inv;
#pragma unroll
            for (int r = 0; r < 12; ++r) m1[r] = (r == p) ? pv1 : m1[r] - cp[r] * pv1;
        }
    }
#pragma unroll
    for (int r = 0; r < 12; ++r) { Mx[r * MC + lane] = m0[r]; if (lane + 64 < 79) Mx[r * MC + lane + 64] = m1[r]; }
    __syncthreads();
    ROWS_STAMP(3)
    // F = -G^-1 [ ... ]: columns 12 .. 78 of the reduced block
    float* F = Fg + (size_t)item * 12 * FCOL;
    for (int e = lane; e < 12 * FCOL; e += 64) {
        const int r = e / FCOL, c = e % FCOL;
        F[e] = c < 67 ? (float)(-Mx[r * MC + 12 + c]) : 0.f;
    }
    ROWS_STAMP(4)
    // the forces leave the dynamics: A <- A + B_f F_x, B_tau <- B_tau + B_f F_tau.  A lane owns a COLUMN (of A: lanes 0 .. 47; of B_tau:
    // 18 lanes in a second round): its 12 entries of the reduced block stay in registers, the rows of B_f are broadcast reads, the
    // row-wise read-modify-write of A / B is coalesced across the lanes
    {
        // [A | B_tau] (48 x 66) -= B_f (48 x 12) X (12 x 66) on the float64 matrix cores: 3 row tiles x 5 column tiles x 3 k-steps, the
        // accumulators start as the current entries (float32 -> float64, fetched in the C layout: four 64-byte row pieces per load)
        double bfa[3][3];
#pragma unroll
        for (int ti = 0; ti < 3; ++ti)
#pragma unroll
            for (int t = 0; t < 3; ++t) bfa[ti][t] = -Bf[(16 * ti + r16) * 12 + 4 * t + kq];
#pragma unroll
        for (int tj = 0; tj < 5; ++tj) {
            const int cfull = 16 * tj + r16;                 // column of [A | B_tau | pad]
            const bool live = cfull < NX + b2z1::NJ;
            float* cbase = cfull < NX ? A + cfull : Bm + (live ? cfull - NX : 0);
            const int ld = cfull < NX ? NX : NUP;
            double xb[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) xb[t] = live ? Mx[(4 * t + kq) * MC + 12 + cfull] : 0.0;
            float cv[3][4];
#pragma unroll
            for (int ti = 0; ti < 3; ++ti)
#pragma unroll
                for (int r = 0; r < 4; ++r) cv[ti][r] = live ? cbase[(16 * ti + kq + 4 * r) * ld] : 0.f;
#pragma unroll
            for (int ti = 0; ti < 3; ++ti) {
                d4 c = {(double)cv[ti][0], (double)cv[ti][1], (double)cv[ti][2], (double)cv[ti][3]};
#pragma unroll
                for (int t = 0; t < 3; ++t) c = __builtin_amdgcn_mfma_f64_16x16x4f64(bfa[ti][t], xb[t], c, 0, 0, 0);
                if (live) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) cbase[(16 * ti + kq + 4 * r) * ld] = (float)c[r];
                }
            }
        }
    }
    if (lane < NX) {
        double acc = 0.0;
        for (int q = 0; q < 12; ++q) acc -= Bf[lane * 12 + q] * Mx[q * MC + 78];
        nx[lane] += acc;
        vec[(size_t)item * VEC + lane] = (float)((double)vec[(size_t)item * VEC + lane] + acc);
    }
    __syncthreads();
    ROWS_STAMP(5)
    for (int e = lane; e < NX * 12; e += 64) Bm[(e / 12) * NUP + b2z1::NJ + e % 12] = 0.f;
    ROWS_STAMP(6)
}

// the foot forces of the solved step, df = F_x dx + F_tau dtau + f0, then what the Riccati kernel does with a finished step
__global__ __launch_bounds__(256) void contact_rows_apply_kernel(const float* Fg, const double* dx, double* du, double* x, double* u, int N, int apply,
                                                                 int* status, const unsigned char* stance)
{
    __shared__ int bad_s;
    const int b = blockIdx.x, tid = threadIdx.x;
    const double* dxb = dx + (size_t)b * (N + 1) * NX;
    double* dub = du + (size_t)b * N * NU;
    if (tid == 0) bad_s = 0;
    __syncthreads();
    int bad = 0;
    for (int e = tid; e < N * 12; e += 256) {
        const int k = e / 12, q = e % 12;
        const float* F = Fg + (((size_t)b * N + k) * 12 + q) * FCOL;
        double acc = (double)F[66];
        for (int j = 0; j < NX; ++j) acc += (double)F[j] * dxb[(size_t)k * NX + j];
        for (int t = 0; t < b2z1::NJ; ++t) acc += (double)F[48 + t] * dub[(size_t)k * NU + t];
        // a foot in the air: exactly -f (its row of the map holds f rounded to float32)
        if (stance && !stance[((size_t)b * N + k) * 4 + q / 3]) acc = -u[((size_t)b * N + k) * NU + b2z1::NJ + q];
        dub[(size_t)k * NU + b2z1::NJ + q] = acc;
        bad |= !isfinite(acc);
    }
    if (bad) bad_s = 1;
    __syncthreads();
    const bool failed = bad_s != 0 || (status && status[b] != 0);
    __syncthreads();
    if (tid == 0 && status) status[b] = failed ? 1 : 0;
    if (apply && !failed) {
        double* xw = x + (size_t)b * (N + 1) * NX;
        double* uw = u + (size_t)b * N * NU;
        for (int i = tid; i < (N + 1) * NX; i += 256) xw[i] += dxb[i];
        for (int i = tid; i < N * NU; i += 256) {
            const int j = i % NU;
            double val = uw[i] + dub[i];
            if (j < b2z1::NJ) { const double lim = b2z1::EFFORT[j]; val = val > lim ? lim : (val < -lim ? -lim : val); }
            uw[i] = val;
        }
    }
}

// ---- iterative refinement of the LQ step (alore_wb_set_refinement) ------------------------------------------------------
// The float32 sweep loses digits where the stage Hessian is stiff (rho J_c' J_c of 1e3 next to velocity weights of 1).  One
// step of iterative refinement: the residuals of the LQ optimality system at the float32 solution (dx, du) are formed in
// float64 -- defect residual A dx_k + B du_k + d_k - dx_{k+1}, stage gradients gx_k + Q dx_k (+ rho J_c' J_c on the
// velocities), gu_k + R du_k, terminal gradient -- and the SAME float32 sweep solves the same problem with these residuals as
// right-hand sides (RicArgs::refine); the correction it returns carries the relative error of the first solution, so the sum
// is accurate to its square.  One workgroup per (problem, stage), thread = row.
__global__ __launch_bounds__(64) void residual_kernel(const float* A32, const float* B32, const double* next, const double* x, const double* u,
                                                      const double* xref, const double* uref, const double* w, const float* pen,
                                                      const double* dx, const double* du, int N, float* res, float* resN)
{
    const int item = blockIdx.x, b = item / N, k = item % N, t = threadIdx.x;
    const double* dxk = dx + ((size_t)b * (N + 1) + k) * NX;
    const double* duk = du + ((size_t)b * N + k) * NU;
    float* out = res + (size_t)item * VEC;
    if (t < NX) {
        const float* Ar = A32 + ((size_t)item * NX + t) * NX;
        const float* Br = B32 + ((size_t)item * NX + t) * NUP;
        double acc = next[(size_t)item * NX + t] - x[((size_t)b * (N + 1) + k + 1) * NX + t];
        for (int j = 0; j < NX; ++j) acc += (double)Ar[j] * dxk[j];
        for (int j = 0; j < NU; ++j) acc += (double)Br[j] * duk[j];
        acc -= dx[((size_t)b * (N + 1) + k + 1) * NX + t];
        out[t] = (float)acc;
        // stage gradient at the first solution
        double gx = w[t] * (x[((size_t)b * (N + 1) + k) * NX + t] - xref[((size_t)b * (N + 1) + k) * NX + t] + dxk[t]);
        if (pen && t >= NQ) {
            const float* J = pen + (size_t)item * PEN; // sqrt(rho) J_c [12][24], then rho J_c' (J_c v) [24]
            double h = (double)J[288 + t - NQ];
            for (int r = 0; r < 12; ++r) {
                double jv = 0.0;
                for (int c = 0; c < NV; ++c) jv += (double)J[r * 24 + c] * dxk[NQ + c];
                h += (double)J[r * 24 + t - NQ] * jv;
            }
            gx += h;
        }
        out[48 + t] = (float)gx;
        if (k == N - 1) {
            const double xe = x[((size_t)b * (N + 1) + N) * NX + t] - xref[((size_t)b * (N + 1) + N) * NX + t] + dx[((size_t)b * (N + 1) + N) * NX + t];
            resN[(size_t)b * NX + t] = (float)(w[NX + NU + t] * xe);
        }
    }
    if (t < 32) {
        out[96 + t] = 0.f;
        out[128 + t] = t < NU ? (float)(w[NX + t] * (u[((size_t)b * N + k) * NU + t] - uref[((size_t)b * N + k) * NU + t] + duk[t])) : 0.f;
    }
}

// dx <- dx + ex, du <- du + eu, then what the Riccati kernel does with a finished step: refuse a step that is not finite,
// add it to the iterate, clip the applied torques to the URDF effort limits
__global__ __launch_bounds__(256) void refine_apply_kernel(double* dx, double* du, const double* ex, const double* eu, double* x, double* u, int N,
                                                           int apply, int* status)
{
    __shared__ int bad_s;
    const int b = blockIdx.x, tid = threadIdx.x;
    double* dxb = dx + (size_t)b * (N + 1) * NX;
    double* dub = du + (size_t)b * N * NU;
    const double* exb = ex + (size_t)b * (N + 1) * NX;
    const double* eub = eu + (size_t)b * N * NU;
    if (tid == 0) bad_s = 0;
    __syncthreads();
    int bad = 0;
    for (int i = tid; i < (N + 1) * NX; i += 256) { const double v = dxb[i] + exb[i]; dxb[i] = v; bad |= !isfinite(v); }
    for (int i = tid; i < N * NU; i += 256) { const double v = dub[i] + eub[i]; dub[i] = v; bad |= !isfinite(v); }
    if (bad) bad_s = 1;
    __syncthreads();
    const bool failed = bad_s != 0 || (status && status[b] != 0);
    __syncthreads();
    if (tid == 0 && status) status[b] = failed ? 1 : 0;
    if (apply && !failed) {
        double* xw = x + (size_t)b * (N + 1) * NX;
        double* uw = u + (size_t)b * N * NU;
        for (int i = tid; i < (N + 1) * NX; i += 256) xw[i] += dxb[i];
        for (int i = tid; i < N * NU; i += 256) {
            const int j = i % NU;
            double val = uw[i] + dub[i];
            if (j < b2z1::NJ) { const double lim = b2z1::EFFORT[j]; val = val > lim ? lim : (val < -lim ? -lim : val); }
            uw[i] = val;
        }
    }
}

// ---- exact working-set mode (alore_wb_set_constraint_mode(h, 1)) --------------------------------------------------------
// The inequality constraints of the LQ problem of a real-time iteration -- torque boxes, no force on a foot in the air, the
// friction pyramid |fx|, |fy| <= mu fz (which contains fz >= 0) of a stance foot -- solved EXACTLY by a working-set iteration
// around the unconstrained MFMA sweep, the scheme of the planar kernel (nmpc_block_kernel.hip) carried over:
//   reduce   the active rows of the current working set are ELIMINATED from the stage data: an input held at a bound is a
//            constant (its column of B moves into the defect), a tangential force tied to a face of the pyramid follows the
//            normal force (fx = +-mu fz: its column is added to the column of fz, its weight mu^2 R to the weight of fz);
//   sweep    riccati_kernel on the reduced data, nothing clamped inside;
//   expand + multipliers + update (ws_step_kernel, float64, one workgroup per problem): the eliminated inputs follow from the
//            solution; the gradient of the Lagrangian with respect to every ORIGINAL input, g_u = R (u + du - uref) + B' lambda
//            with the costates lambda_k = Q (x_k + dx_k - xref_k) + A_k' lambda_{k+1}, is the multiplier of a held input
//            (sign-checked: released when it pulls inward) and ~0 on the free ones; free inputs that left their bounds
//            are held.  All changes of a sweep at once, until the set repeats (or ws_max sweeps).
// Working-set codes per input:
enum : unsigned char { WS_FREE = 0, WS_LOWER = 1, WS_UPPER = 2, WS_TIE_POS = 3, WS_TIE_NEG = 4, WS_ZERO = 5 };

__global__ __launch_bounds__(64) void ws_reduce_kernel(const float* B32, const float* vec, const double* next, const unsigned char* ws, const double* u,
                                                        const double* uref, const double* w, float mu, float* Bw, float* vecw, double* nextw, float* wrs)
{
    __shared__ float val[32];   // constant part of an eliminated input: du_i = val_i (+ tie_i * du_fz)
    __shared__ float tie[32];   // +-mu for a tied tangential component, else 0
    __shared__ int code[32];
    const int item = blockIdx.x, t = threadIdx.x;
    const double* uk = u + (size_t)item * NU;
    const double* Rd = w + NX;
    if (t < 32) {
        const int c = t < NU ? ws[(size_t)item * 32 + t] : WS_FREE;
        float v = 0.f, ti = 0.f;
        if (t < b2z1::NJ) {
            const float eff = (float)b2z1::EFFORT[t];
            if (c == WS_LOWER) v = -eff - (float)uk[t];
            else if (c == WS_UPPER) v = eff - (float)uk[t];
        } else if (t < NU) {
            const int foot = (t - b2z1::NJ) / 3, iz = b2z1::NJ + 3 * foot + 2;
            if (c == WS_ZERO) v = -(float)uk[t];
            else if (c == WS_TIE_POS) { ti = mu; v = mu * (float)uk[iz] - (float)uk[t]; }
            else if (c == WS_TIE_NEG) { ti = -mu; v = -mu * (float)uk[iz] - (float)uk[t]; }
        }
        code[t] = c; val[t] = v; tie[t] = ti;
    }
    __syncthreads();
    if (t < NX) { // row t of B: held inputs into the defect, tied ones onto their normal force
        const float* br = B32 + ((size_t)item * NX + t) * NUP;
        float* bo = Bw + ((size_t)item * NX + t) * NUP;
        float row[NUP];
#pragma unroll
        for (int i = 0; i < NUP; ++i) row[i] = br[i];
        double dadd = 0.0;
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            if (code[i] != WS_FREE) dadd += (double)row[i] * (double)val[i];
        }
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const int ix = b2z1::NJ + 3 * f;
            row[ix + 2] += tie[ix] * row[ix] + tie[ix + 1] * row[ix + 1];
        }
#pragma unroll
        for (int i = 0; i < NU; ++i) bo[i] = code[i] != WS_FREE ? 0.f : row[i];
        bo[30] = 0.f; bo[31] = 0.f;
        vecw[(size_t)item * VEC + t] = (float)((double)vec[(size_t)item * VEC + t] + dadd);
        nextw[(size_t)item * NX + t] = next[(size_t)item * NX + t] + dadd;
    }
    // input gradients and weights of the reduced problem; the other pieces of vec travel as they are
    if (t < 32) {
        float gu = 0.f, wr = t < NU ? (float)Rd[t] : 1.f;
        if (t < NU && code[t] == WS_FREE) {
            gu = wr * vec[(size_t)item * VEC + 128 + t];
            if (t >= b2z1::NJ && (t - b2z1::NJ) % 3 == 2) { // a normal force: what follows it
                for (int a = 1; a <= 2; ++a) {
                    const int ia = t - a; // fy, fx of the same foot
                    if (tie[ia] != 0.f) {
                        const float ra = (float)Rd[ia];
                        wr += tie[ia] * tie[ia] * ra;
                        gu += tie[ia] * ra * ((float)(uk[ia] - uref[(size_t)item * NU + ia]) + val[ia]);
                    }
                }
            }
        }
        vecw[(size_t)item * VEC + 128 + t] = gu;
        wrs[(size_t)item * 32 + t] = wr;
        vecw[(size_t)item * VEC + 96 + t] = vec[(size_t)item * VEC + 96 + t];
    }
    if (t < NX) vecw[(size_t)item * VEC + 48 + t] = vec[(size_t)item * VEC + 48 + t];
}

// The point the iteration stands at: z, a FEASIBLE input step of the problem (z = 0 at the start: the iterate itself is feasible).
// One workgroup per problem, float64: the state step of z by the linearised dynamics (dxz), the cost f(z), the costates and the
// gradient g_u of the cost with respect to every input, and from (z, g_u) the BINDING set -- the rows active at z whose multiplier
// has the right sign -- as the working set of the next sweep (projected Newton, Bertsekas 1982: the binding set comes from the
// gradient at a feasible point, never from the multipliers of an infeasible subspace minimiser, so the iteration cannot run in circles).
__device__ __forceinline__ double wb_stage_cost(const double* Qd, const double* Rd, const double* xk, const double* dxk, const double* xrk,
                                                const double* uk, const double* zk, const double* urk, int t)
{
    double c = 0.0;
    if (t < NX) { const double e = xk[t] + dxk[t] - xrk[t]; c += 0.5 * Qd[t] * e * e; }
    if (t < NU) { const double e = uk[t] + zk[t] - urk[t]; c += 0.5 * Rd[t] * e * e; }
    return c;
}
__device__ __forceinline__ double wb_wave_sum(double v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(64) void ws_point_kernel(const float* A32, const float* B32, const double* next, const double* x, const double* u, const double* x0,
                                                       const double* xref, const double* uref, const double* w, const double* z, double* dxz, unsigned char* ws,
                                                       const unsigned char* stance, float mu, int N, int limits, int cones, double* fval, int* set_changed,
                                                       int* total_changed, double* gu_out)
{
    __shared__ double dxa[NX], dxb_[NX], lam[NX], lamn[NX], gu[32], gsc[32], zn[32];
    __shared__ int nchg;
    const int b = blockIdx.x, t = threadIdx.x;
    const double *Qd = w, *Rd = w + NX, *QNd = w + NX + NU;
    const double* xb = x + (size_t)b * (N + 1) * NX;
    const double* xr = xref + (size_t)b * (N + 1) * NX;
    double* dxo = dxz + (size_t)b * (N + 1) * NX;
    if (t == 0) nchg = 0;
    if (t < NX) { dxa[t] = x0[(size_t)b * NX + t] - xb[t]; dxo[t] = dxa[t]; }
    __syncthreads();
    double cost = 0.0;
    double *cur = dxa, *nxt = dxb_;
    for (int k = 0; k < N; ++k) { // forward: the state step of z, the cost
        const size_t item = (size_t)b * N + k;
        cost += wb_stage_cost(Qd, Rd, xb + (size_t)k * NX, cur, xr + (size_t)k * NX, u + item * NU, z + item * NU, uref + item * NU, t);
        if (t < NX) {
            const float* Ar = A32 + (item * NX + t) * NX;
            const float* Br = B32 + (item * NX + t) * NUP;
            double acc = next[item * NX + t] - xb[(size_t)(k + 1) * NX + t];
            for (int j = 0; j < NX; ++j) acc += (double)Ar[j] * cur[j];
            for (int j = 0; j < NU; ++j) acc += (double)Br[j] * z[item * NU + j];
            nxt[t] = acc;
            dxo[(size_t)(k + 1) * NX + t] = acc;
        }
        __syncthreads();
        double* sw = cur; cur = nxt; nxt = sw;
    }
    if (t < NX) {
        const double e = xb[(size_t)N * NX + t] + cur[t] - xr[(size_t)N * NX + t];
        cost += 0.5 * QNd[t] * e * e;
        lamn[t] = QNd[t] * e;
    }
    cost = wb_wave_sum(cost);
    if (t == 0) fval[b] = cost;
    __syncthreads();
    for (int k = N - 1; k >= 0; --k) { // backward: costates, input gradients, the binding set of every stage
        const size_t item = (size_t)b * N + k;
        const float* A = A32 + item * NX * NX;
        const float* Bm = B32 + item * NX * NUP;
        const double* dxk = dxo + (size_t)k * NX;
        if (t < NU) {
            double acc = Rd[t] * (u[item * NU + t] + z[item * NU + t] - uref[item * NU + t]);
            double mag = fabs(acc);
            for (int j = 0; j < NX; ++j) { const double term = (double)Bm[j * NUP + t] * lamn[j]; acc += term; mag += fabs(term); }
            gu[t] = acc; gsc[t] = mag;
            zn[t] = u[item * NU + t] + z[item * NU + t];
            if (gu_out) gu_out[item * NU + t] = acc;
        }
        if (t < NX) {
            double acc = Qd[t] * (xb[(size_t)k * NX + t] + dxk[t] - xr[(size_t)k * NX + t]);
            for (int j = 0; j < NX; ++j) acc += (double)A[j * NX + t] * lamn[j];
            lam[t] = acc;
        }
        __syncthreads();
        int moved = 0;
        if (t < b2z1::NJ && limits) {
            const double eff = b2z1::EFFORT[t], tolv = 1e-7 * eff, tolg = 1e-9 * gsc[t];
            const unsigned char c = ws[item * 32 + t];
            unsigned char n = WS_FREE;
            if (zn[t] <= -eff + tolv && gu[t] > tolg) n = WS_LOWER;       // at the lower bound and the cost rises inward
            else if (zn[t] >= eff - tolv && gu[t] < -tolg) n = WS_UPPER;
            if (n != c) { ws[item * 32 + t] = n; moved = 1; }
        } else if (t >= b2z1::NJ && t < b2z1::NJ + 4 && cones) {
            const int f = t - b2z1::NJ, ix = b2z1::NJ + 3 * f, iy = ix + 1, iz = ix + 2;
            const bool st = stance ? stance[item * 4 + f] != 0 : true;
            const unsigned char cx = ws[item * 32 + ix], cy = ws[item * 32 + iy], cz = ws[item * 32 + iz];
            unsigned char nx_ = WS_FREE, ny_ = WS_FREE, nz_ = WS_FREE;
            const double m = (double)mu, fz = zn[iz], tolv = 1e-7 * (1.0 + fabs(fz)), tolg = 1e-9 * (gsc[ix] + gsc[iy] + gsc[iz]);
            if (!st) { nx_ = ny_ = nz_ = WS_ZERO; }
            else if (fz <= tolv) {
                // at the apex: binding while -g lies in the polar cone of the pyramid, g_z - mu (|g_x| + |g_y|) >= 0
                if (gu[iz] - m * (fabs(gu[ix]) + fabs(gu[iy])) >= -tolg) { nx_ = ny_ = nz_ = WS_ZERO; }
            } else {
                // on a face: multiplier of  fx - mu fz <= 0  is -g_x, of  -fx - mu fz <= 0  is g_x
                if (zn[ix] >= m * fz - tolv && gu[ix] < -tolg) nx_ = WS_TIE_POS;
                else if (zn[ix] <= -m * fz + tolv && gu[ix] > tolg) nx_ = WS_TIE_NEG;
                if (zn[iy] >= m * fz - tolv && gu[iy] < -tolg) ny_ = WS_TIE_POS;
                else if (zn[iy] <= -m * fz + tolv && gu[iy] > tolg) ny_ = WS_TIE_NEG;
            }
            if (nx_ != cx || ny_ != cy || nz_ != cz) {
                ws[item * 32 + ix] = nx_; ws[item * 32 + iy] = ny_; ws[item * 32 + iz] = nz_;
                moved = (nx_ != cx) + (ny_ != cy) + (nz_ != cz);
            }
        }
        if (moved) atomicAdd(&nchg, moved);
        if (t < NX) lamn[t] = lam[t];
        __syncthreads();
    }
    if (t == 0) {
        set_changed[b] = nchg;
        if (nchg) atomicAdd(total_changed, nchg);
    }
}

// The step of the primal active-set method (Nocedal & Wright, Alg. 16.3), one workgroup per problem, float64.  The sweep's solution of the
// reduced problem, expanded to all inputs (zs, with the state step dxs of the sweep's own closed-loop forward pass), is the minimiser on
// the face of the working set; z moves towards it along the straight line -- both ends satisfy the linearised dynamics, so does every
// point between them, and the state step is the same combination: nothing is rolled out open-loop through the (unstable) dynamics -- as
// far as the first row outside the working set allows (ratio test); that row joins the working set.  full[b] = 1: the whole step was
// taken, z is the face minimiser and ws_release_kernel may look at its multipliers.
__global__ __launch_bounds__(256) void ws_ratio_kernel(const double* u, double* z, double* zs, double* dxz, const double* dxs, unsigned char* ws,
                                                        const unsigned char* stance, float mu, int N, int limits, int cones, int* full, int* blocked_total)
{
    __shared__ double amin[256];
    __shared__ int aidx[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const double m = (double)mu;
    for (int e = tid; e < N * NU; e += 256) { // expand the eliminated inputs of the sweep's solution
        const int k = e / NU, i = e % NU;
        const size_t item = (size_t)b * N + k;
        const int code = ws[item * 32 + i];
        if (code == WS_FREE) continue;
        const double ui = u[item * NU + i];
        double v;
        if (i < b2z1::NJ) v = (code == WS_LOWER ? -b2z1::EFFORT[i] : b2z1::EFFORT[i]) - ui;
        else if (code == WS_ZERO) v = -ui;
        else {
            const int iz = b2z1::NJ + 3 * ((i - b2z1::NJ) / 3) + 2;
            v = (code == WS_TIE_POS ? 1.0 : -1.0) * m * (u[item * NU + iz] + zs[item * NU + iz]) - ui;
        }
        zs[item * NU + i] = v;
    }
    __syncthreads();
    // ratio test over the rows outside the working set: candidate id = (stage * 64 + row) * 2 + side
    double best = 1.0;
    int bid = -1;
    auto consider = [&](double c, double dir, int id) { // row value c <= 0 at z, changes by dir per unit step
        if (dir > 1e-14 * (1.0 + fabs(c))) {
            const double a = (c >= 0.0 ? 0.0 : -c / dir);
            if (a < best || (a == best && bid >= 0 && id < bid)) { best = a; bid = id; }
        }
    };
    for (int e = tid; e < N * 26; e += 256) {
        const int k = e / 26, r = e % 26;
        const size_t item = (size_t)b * N + k;
        if (r < b2z1::NJ) {
            if (!limits || ws[item * 32 + r] != WS_FREE) continue;
            const double eff = b2z1::EFFORT[r], v = u[item * NU + r] + z[item * NU + r], pv = zs[item * NU + r] - z[item * NU + r];
            consider(v - eff, pv, (k * 64 + r) * 2 + 1);
            consider(-v - eff, -pv, (k * 64 + r) * 2);
        } else if (cones) {
            const int f = (r - b2z1::NJ) >> 1, ax = (r - b2z1::NJ) & 1, ix = b2z1::NJ + 3 * f, ia = ix + ax, iz = ix + 2;
            const bool st = stance ? stance[item * 4 + f] != 0 : true;
            if (!st || ws[item * 32 + iz] == WS_ZERO) continue;
            const int ca = ws[item * 32 + ia];
            const double fa = u[item * NU + ia] + z[item * NU + ia], fz = u[item * NU + iz] + z[item * NU + iz];
            const double pa = zs[item * NU + ia] - z[item * NU + ia], pz = zs[item * NU + iz] - z[item * NU + iz];
            if (ca != WS_TIE_POS) consider(fa - m * fz, pa - m * pz, (k * 64 + 32 + 2 * f + ax) * 2 + 1);
            if (ca != WS_TIE_NEG) consider(-fa - m * fz, -pa - m * pz, (k * 64 + 32 + 2 * f + ax) * 2);
        }
    }
    amin[tid] = best; aidx[tid] = bid;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            const double a2 = amin[tid + o]; const int i2 = aidx[tid + o];
            if (i2 >= 0 && (aidx[tid] < 0 || a2 < amin[tid] || (a2 == amin[tid] && i2 < aidx[tid]))) { amin[tid] = a2; aidx[tid] = i2; }
        }
        __syncthreads();
    }
    const double alpha = aidx[0] >= 0 ? fmin(amin[0], 1.0) : 1.0;
    const int blk = (aidx[0] >= 0 && amin[0] < 1.0) ? aidx[0] : -1;
    __syncthreads();
    for (int e = tid; e < N * NU; e += 256) { const size_t i = (size_t)b * N * NU + e; z[i] += alpha * (zs[i] - z[i]); }
    for (int e = tid; e < (N + 1) * NX; e += 256) { const size_t i = (size_t)b * (N + 1) * NX + e; dxz[i] += alpha * (dxs[i] - dxz[i]); }
    if (tid == 0) {
        full[b] = blk < 0 ? 1 : 0;
        if (blk >= 0) {
            atomicAdd(blocked_total, 1);
            const int side = blk & 1, row = (blk >> 1) & 63, k = blk >> 7;
            const size_t item = (size_t)b * N + k;
            if (row < 32) ws[item * 32 + row] = side ? WS_UPPER : WS_LOWER;
            else {
                const int f = (row - 32) >> 1, ax = (row - 32) & 1, ix = b2z1::NJ + 3 * f, ia = ix + ax;
                const int old = ws[item * 32 + ia];
                if (old == WS_TIE_POS || old == WS_TIE_NEG) { ws[item * 32 + ix] = ws[item * 32 + ix + 1] = ws[item * 32 + ix + 2] = WS_ZERO; } // both faces of an axis: the apex
                else ws[item * 32 + ia] = side ? WS_TIE_POS : WS_TIE_NEG;
            }
        }
    }
}

// At a face minimiser (full[b] = 1): costates and the gradient of the cost with respect to every input from (dxz, z); the multiplier of
// every held row is sign-checked and the WORST offender (largest violation relative to the size of the terms of its gradient) leaves the
// working set; none: the problem is solved (done[b] = 1).  One workgroup per problem.
__global__ __launch_bounds__(64) void ws_release_kernel(const float* A32, const float* B32, const double* x, const double* u, const double* xref, const double* uref,
                                                         const double* w, const double* z, const double* dxz, unsigned char* ws, const unsigned char* stance, float mu,
                                                         int N, const int* full, int* done, int* open_total, double* gu_out)
{
    __shared__ double lam[NX], lamn[NX], gu[32], gsc[32];
    __shared__ double wv[64];
    __shared__ int wi[64];
    const int b = blockIdx.x, t = threadIdx.x;
    if (!full[b]) { if (t == 0) { done[b] = 1; atomicAdd(open_total, 1); } return; } // done[b]: 1 while the problem is still open
    const double *Qd = w, *Rd = w + NX, *QNd = w + NX + NU;
    const double* xb = x + (size_t)b * (N + 1) * NX;
    const double* xr = xref + (size_t)b * (N + 1) * NX;
    const double* dxo = dxz + (size_t)b * (N + 1) * NX;
    const double m = (double)mu;
    if (t < NX) lamn[t] = QNd[t] * (xb[(size_t)N * NX + t] + dxo[(size_t)N * NX + t] - xr[(size_t)N * NX + t]);
    double worst = 0.0; // most negative normalised multiplier seen by this thread
    int wid = -1;       // (stage * 32 + input), for a foot at the apex its fz
    __syncthreads();
    for (int k = N - 1; k >= 0; --k) {
        const size_t item = (size_t)b * N + k;
        const float* A = A32 + item * NX * NX;
        const float* Bm = B32 + item * NX * NUP;
        if (t < NU) {
            double acc = Rd[t] * (u[item * NU + t] + z[item * NU + t] - uref[item * NU + t]);
            double mag = fabs(acc);
            for (int j = 0; j < NX; ++j) { const double term = (double)Bm[j * NUP + t] * lamn[j]; acc += term; mag += fabs(term); }
            gu[t] = acc; gsc[t] = mag + 1e-300;
            if (gu_out) gu_out[item * NU + t] = acc;
        }
        if (t < NX) {
            double acc = Qd[t] * (xb[(size_t)k * NX + t] + dxo[(size_t)k * NX + t] - xr[(size_t)k * NX + t]);
            for (int j = 0; j < NX; ++j) acc += (double)A[j * NX + t] * lamn[j];
            lam[t] = acc;
        }
        __syncthreads();
        if (t < NU) {
            const int c = ws[item * 32 + t];
            double lm = 0.0, sc = gsc[t]; // multiplier (>= 0 wanted)
            bool held = false;
            if (t < b2z1::NJ) {
                if (c == WS_LOWER) { lm = gu[t]; held = true; } else if (c == WS_UPPER) { lm = -gu[t]; held = true; }
            } else {
                const int f = (t - b2z1::NJ) / 3, ax = (t - b2z1::NJ) % 3, ix = b2z1::NJ + 3 * f;
                const bool st = stance ? stance[item * 4 + f] != 0 : true;
                if (st && ax < 2 && c == WS_TIE_POS) { lm = -gu[t]; held = true; }
                else if (st && ax < 2 && c == WS_TIE_NEG) { lm = gu[t]; held = true; }
                else if (st && ax == 2 && c == WS_ZERO) { // apex: stay while g_z - mu (|g_x| + |g_y|) >= 0
                    lm = gu[t] - m * (fabs(gu[ix]) + fabs(gu[ix + 1])); sc = gsc[ix] + gsc[ix + 1] + gsc[t]; held = true;
                }
            }
            if (held) {
                const double rel = lm / sc;
                if (rel < -3e-4 && rel < worst) { worst = rel; wid = k * 32 + t; } // the sweep is float32: a multiplier that misses zero by less is zero
            }
        }
        if (t < NX) lamn[t] = lam[t];
        __syncthreads();
    }
    wv[t] = worst; wi[t] = wid;
    __syncthreads();
    if (t == 0) {
        double bw = 0.0; int bi = -1;
        for (int i = 0; i < 64; ++i)
            if (wi[i] >= 0 && (bi < 0 || wv[i] < bw || (wv[i] == bw && wi[i] < bi))) { bw = wv[i]; bi = wi[i]; }
        if (bi < 0) done[b] = 0;
        else {
            done[b] = 1;
            atomicAdd(open_total, 1);
            const int k = bi / 32, i = bi % 32;
            const size_t item = (size_t)b * N + k;
            if (ws[item * 32 + i] == WS_ZERO) { const int ix = i - 2; ws[item * 32 + ix] = ws[item * 32 + ix + 1] = ws[item * 32 + i] = WS_FREE; } // off the apex
            else ws[item * 32 + i] = WS_FREE;
        }
    }
}

// the finished step of the exact mode: refuse a step that is not finite, add it to the iterate; the applied inputs satisfy the
// constraints to the last bit (torques clipped, forces projected: what float32 rounding of the sweep may have left outside)
__global__ __launch_bounds__(256) void ws_apply_kernel(const double* dx, const double* du, double* x, double* u, int N, int* status, const unsigned char* stance, float mu)
{
    __shared__ int bad_s;
    const int b = blockIdx.x, tid = threadIdx.x;
    const double* dxb = dx + (size_t)b * (N + 1) * NX;
    const double* dub = du + (size_t)b * N * NU;
    if (tid == 0) bad_s = 0;
    __syncthreads();
    int bad = 0;
    for (int i = tid; i < (N + 1) * NX; i += 256) bad |= !isfinite(dxb[i]);
    for (int i = tid; i < N * NU; i += 256) bad |= !isfinite(dub[i]);
    if (bad) bad_s = 1;
    __syncthreads();
    const bool failed = bad_s != 0 || (status && status[b] != 0);
    __syncthreads();
    if (tid == 0 && status) status[b] = failed ? 1 : 0;
    if (failed) return;
    double* xw = x + (size_t)b * (N + 1) * NX;
    double* uw = u + (size_t)b * N * NU;
    for (int i = tid; i < (N + 1) * NX; i += 256) xw[i] += dxb[i];
    for (int i = tid; i < N * b2z1::NJ; i += 256) {
        const int k = i / b2z1::NJ, j = i % b2z1::NJ;
        const double lim = b2z1::EFFORT[j];
        double val = uw[(size_t)k * NU + j] + dub[(size_t)k * NU + j];
        uw[(size_t)k * NU + j] = val > lim ? lim : (val < -lim ? -lim : val);
    }
    for (int it = tid; it < N * 4; it += 256) {
        const int kk = it >> 2, foot = it & 3, i0 = kk * NU + b2z1::NJ + 3 * foot;
        const bool st = stance ? stance[((size_t)b * N + kk) * 4 + foot] != 0 : true;
        const double fzr = uw[i0 + 2] + dub[i0 + 2], fz = st ? (fzr > 0.0 ? fzr : 0.0) : 0.0, lim = (double)mu * fz;
        double fx = uw[i0] + dub[i0], fy = uw[i0 + 1] + dub[i0 + 1];
        fx = fx > lim ? lim : (fx < -lim ? -lim : fx);
        fy = fy > lim ? lim : (fy < -lim ? -lim : fy);
        uw[i0] = fx; uw[i0 + 1] = fy; uw[i0 + 2] = fz;
    }
}

// one lane per evaluation point (alore_wb_aba): the articulated-body algorithm of wb_aba.h
__global__ __launch_bounds__(64) void aba_kernel(int n, const double* q, const double* v, const double* u, double grav, double* acc)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    Eval e{q + (size_t)t * NQ, v + (size_t)t * NV, v + (size_t)t * NV, u + (size_t)t * NU + 18, 1.0, 0.0, 1.0, -1, -1, -1, -1, 0.0, 0.0, 0.0, 0.0, grav};
    aba(e, u + (size_t)t * NU, Sink{acc + (size_t)t * NV, 1, 1.0, 0});
}

// receding horizon: stage k <- stage k + 1 for the iterate (the last stage is repeated), as MpcWrapper::update's shift
__global__ void shift_kernel(int B, int N, double* x, double* u)
{
    const int b = blockIdx.x, t = threadIdx.x;
    double* xb = x + (size_t)b * (N + 1) * NX;
    double* ub = u + (size_t)b * N * NU;
    for (int k = 0; k < N; ++k) { // sequential in k, parallel inside a stage: reads of stage k + 1 precede its overwrite
        if (t < NX) xb[(size_t)k * NX + t] = xb[(size_t)(k + 1) * NX + t];
        if (t < NU && k + 1 < N) ub[(size_t)k * NU + t] = ub[(size_t)(k + 1) * NU + t];
        __syncthreads();
    }
}

} // namespace wb

// =====================================================================================================================
// C ABI
// =====================================================================================================================
struct alore_wb_solver {
    alore_wb_config cfg{};
    std::string err;
    double *d_x = nullptr, *d_u = nullptr, *d_x0 = nullptr, *d_xref = nullptr, *d_uref = nullptr, *d_w = nullptr;
    float *d_A = nullptr, *d_B = nullptr;
    double* d_next = nullptr;
    float* d_vec = nullptr; // [B][N][160]
    double *d_dx = nullptr, *d_du = nullptr;
    float *d_K = nullptr, *d_kff = nullptr;
    int* d_status = nullptr;
    long long* d_stamps = nullptr; // [64] when ALORE_WB_STAMPS=1
    int limits = 1;                // alore_wb_set_torque_limits
    int cones = 0;                 // alore_wb_set_contact_constraints
    float mu = 0.7f;
    unsigned char* d_stance = nullptr; // [max_problems][N][4], null = all stance
    float* d_pen = nullptr;            // [max_problems][N][PEN], allocated by alore_wb_set_contact_penalty
    double rho = 0.0;
    int rows = 0;                      // alore_wb_set_contact_rows
    float* d_F = nullptr;              // [B][N][12][FCOL] the foot forces as a map of (dx, dtau)
    int refine = 0;                    // alore_wb_set_refinement
    float *d_res = nullptr, *d_resN = nullptr; // [B][N][VEC], [B][48] residuals of the refinement pass
    double *d_ex = nullptr, *d_eu = nullptr;   // its correction
    // exact working-set mode (alore_wb_set_constraint_mode)
    int exact = 0, ws_max = 8, ws_sweeps = 0;
    unsigned char* d_ws = nullptr;             // [B][N][32] working-set codes, kept from one real-time iteration to the next
    float *d_Bw = nullptr, *d_vecw = nullptr, *d_wrs = nullptr; // the reduced stage data
    double *d_nextw = nullptr, *d_gu = nullptr;                 // ... and the input gradients (multipliers) at the point the iteration stands at
    double *d_z = nullptr, *d_dxz = nullptr, *d_fval = nullptr; // that point (a feasible input step), its state step, its cost
    int *d_changed = nullptr, *d_total = nullptr, *d_nonfull = nullptr, *d_moved = nullptr; // d_total: [0] set changes, [1] non-full steps, [2] problems that moved
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    float ms_lin = -1.f, ms_ric = -1.f;
    bool timed = false;
};

namespace {
int fail(alore_wb_handle h, int code, const char* what, hipError_t e = hipSuccess)
{
    if (h) {
        h->err = what;
        if (e != hipSuccess) { h->err += ": "; h->err += hipGetErrorString(e); }
    }
    return code;
}
#define WB_TRY(h, call)                                                        \
    do {                                                                       \
        hipError_t e_ = (call);                                                \
        if (e_ != hipSuccess) return fail(h, ALORE_WB_E_HIP, #call, e_);       \
    } while (0)

template <class T>
hipError_t zalloc(T** p, size_t n)
{
    hipError_t e = hipMalloc((void**)p, n * sizeof(T));
    if (e == hipSuccess) e = hipMemset(*p, 0, n * sizeof(T));
    return e;
}

struct Tmp { // scratch device buffers of the synchronous dynamics entry points
    std::vector<void*> p;
    ~Tmp() { for (void* x : p) (void)hipFree(x); }
    template <class T>
    T* get(size_t n)
    {
        T* d = nullptr;
        if (zalloc(&d, n) != hipSuccess) return nullptr;
        p.push_back(d);
        return d;
    }
};
} // namespace

extern "C" {

void alore_wb_default_config(alore_wb_config* c)
{
    if (!c) return;
    c->horizon = 20; c->dt = 0.01; c->device = 0; c->max_problems = 4096;
}

int alore_wb_kernel_info(int* stage_lds_bytes, int* riccati_lds_bytes)
{
    if (stage_lds_bytes) *stage_lds_bytes = (int)sizeof(wb::StageLds);
    if (riccati_lds_bytes) *riccati_lds_bytes = (int)sizeof(wb::RicLds);
    return ALORE_WB_OK;
}

int alore_wb_model_info(double* masses, double* lower, double* upper, double* effort)
{
    if (masses) for (int i = 0; i < b2z1::NB; ++i) masses[i] = b2z1::MASS[i];
    for (int i = 0; i < b2z1::NJ; ++i) {
        if (lower) lower[i] = b2z1::Q_LOWER[i];
        if (upper) upper[i] = b2z1::Q_UPPER[i];
        if (effort) effort[i] = b2z1::EFFORT[i];
    }
    return ALORE_WB_OK;
}

int alore_wb_create(const alore_wb_config* cfg, alore_wb_handle* out)
{
    if (!cfg || !out || cfg->horizon < 1 || cfg->horizon > 32 || cfg->max_problems < 1 || !(cfg->dt > 0)) return ALORE_WB_E_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= cfg->device) { (void)hipGetLastError(); return ALORE_WB_E_NO_DEVICE; }
    if (hipSetDevice(cfg->device) != hipSuccess) return ALORE_WB_E_NO_DEVICE;
    alore_wb_solver* h = new (std::nothrow) alore_wb_solver;
    if (!h) return ALORE_WB_E_NOMEM;
    h->cfg = *cfg;
    const size_t B = cfg->max_problems, N = cfg->horizon;
    bool ok = zalloc(&h->d_x, B * (N + 1) * wb::NX) == hipSuccess && zalloc(&h->d_u, B * N * wb::NU) == hipSuccess &&
              zalloc(&h->d_x0, B * wb::NX) == hipSuccess && zalloc(&h->d_xref, B * (N + 1) * wb::NX) == hipSuccess &&
              zalloc(&h->d_uref, B * N * wb::NU) == hipSuccess && zalloc(&h->d_w, (size_t)(2 * wb::NX + wb::NU)) == hipSuccess &&
              zalloc(&h->d_A, B * N * wb::NX * wb::NX) == hipSuccess && zalloc(&h->d_B, B * N * wb::NX * wb::NUP) == hipSuccess &&
              zalloc(&h->d_next, B * N * wb::NX) == hipSuccess && zalloc(&h->d_dx, B * (N + 1) * wb::NX) == hipSuccess &&
              zalloc(&h->d_du, B * N * wb::NU) == hipSuccess && zalloc(&h->d_K, B * N * 32 * 48) == hipSuccess &&
              zalloc(&h->d_kff, B * N * 32) == hipSuccess && zalloc(&h->d_status, B) == hipSuccess &&
              zalloc(&h->d_vec, B * N * wb::VEC) == hipSuccess;
    for (int i = 0; i < 3 && ok; ++i) ok = hipEventCreate(&h->ev[i]) == hipSuccess;
    if (ok && std::getenv("ALORE_WB_STAMPS")) ok = zalloc(&h->d_stamps, (size_t)64) == hipSuccess;
    if (!ok) { alore_wb_destroy(h); return ALORE_WB_E_NOMEM; }
    *out = h;
    return ALORE_WB_OK;
}

int alore_wb_destroy(alore_wb_handle h)
{
    if (!h) return ALORE_WB_E_INVALID;
    if (h->d_stamps) { // diagnostic: contact rows kernel, workgroup 0 of the last launch
        long long rs[8];
        if (hipMemcpyFromSymbol(rs, HIP_SYMBOL(wb::g_rows_stamps), sizeof(rs)) == hipSuccess && rs[6] > rs[0]) {
            std::fprintf(stderr, "[alore_wb stamps] contact rows kernel, workgroup 0, cycles: stage-in %lld, J x [..] %lld, elimination %lld, map out %lld, A / B update %lld, rest %lld\n",
                         rs[1] - rs[0], rs[2] - rs[1], rs[3] - rs[2], rs[4] - rs[3], rs[5] - rs[4], rs[6] - rs[5]);
        }
    }
    if (h->d_stamps) { // diagnostic: cycles of workgroup 0 per phase, summed over all launches
        long long st[64];
        if (hipMemcpy(st, h->d_stamps, sizeof(st), hipMemcpyDeviceToHost) == hipSuccess) {
            std::fprintf(stderr, "[alore_wb stamps] stage kernel, workgroup 0, cycles per phase:");
            for (int i = 0; i < 8; ++i) std::fprintf(stderr, " %lld", st[i]);
            std::fprintf(stderr, "\n[alore_wb stamps] riccati kernel, workgroup 0, cycles per phase (all stages):");
            for (int i = 0; i < 12; ++i) std::fprintf(stderr, " %lld", st[32 + i]);
            std::fprintf(stderr, "\n");
        }
    }
    void* ptrs[] = {h->d_x, h->d_u, h->d_x0, h->d_xref, h->d_uref, h->d_w, h->d_A, h->d_B, h->d_next, h->d_dx, h->d_du, h->d_K, h->d_kff, h->d_stamps, h->d_status, h->d_vec, h->d_stance, h->d_pen, h->d_res, h->d_resN, h->d_ex, h->d_eu, h->d_F,
                    h->d_ws, h->d_Bw, h->d_vecw, h->d_wrs, h->d_nextw, h->d_gu, h->d_changed, h->d_total, h->d_nonfull, h->d_moved, h->d_z, h->d_dxz, h->d_fval};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (int i = 0; i < 3; ++i) if (h->ev[i]) (void)hipEventDestroy(h->ev[i]);
    delete h;
    return ALORE_WB_OK;
}

const char* alore_wb_last_error(alore_wb_handle h) { return h ? h->err.c_str() : "null handle"; }

int alore_wb_rnea(alore_wb_handle h, int n, const double* q, const double* v, const double* a, const double* f, int gravity, double* tau)
{
    if (!h || n <= 0 || !q || !v || !a || !tau) return fail(h, ALORE_WB_E_INVALID, "rnea: bad argument");
    WB_TRY(h, hipSetDevice(h->cfg.device));
    Tmp t;
    double *dq = t.get<double>((size_t)n * 24), *dv = t.get<double>((size_t)n * 24), *da = t.get<double>((size_t)n * 24);
    double *df = f ? t.get<double>((size_t)n * 12) : nullptr, *dt = t.get<double>((size_t)n * 24);
    if (!dq || !dv || !da || !dt || (f && !df)) return fail(h, ALORE_WB_E_NOMEM, "rnea: hipMalloc");
    WB_TRY(h, hipMemcpy(dq, q, sizeof(double) * n * 24, hipMemcpyHostToDevice));
    WB_TRY(h, hipMemcpy(dv, v, sizeof(double) * n * 24, hipMemcpyHostToDevice));
    WB_TRY(h, hipMemcpy(da, a, sizeof(double) * n * 24, hipMemcpyHostToDevice));
    if (f) WB_TRY(h, hipMemcpy(df, f, sizeof(double) * n * 12, hipMemcpyHostToDevice));
    wb::rnea_kernel<<<(n + 63) / 64, 64>>>(n, dq, dv, da, df, gravity ? b2z1::GRAVITY : 0.0, dt);
    WB_TRY(h, hipGetLastError());
    WB_TRY(h, hipMemcpy(tau, dt, sizeof(double) * n * 24, hipMemcpyDeviceToHost));
    return ALORE_WB_OK;
}

int alore_wb_aba(alore_wb_handle h, int n, const double* q, const double* v, const double* u, double* a)
{
    if (!h || n <= 0 || !q || !v || !u || !a) return fail(h, ALORE_WB_E_INVALID, "aba: bad argument");
    WB_TRY(h, hipSetDevice(h->cfg.device));
    Tmp t;
    double *dq = t.get<double>((size_t)n * 24), *dv = t.get<double>((size_t)n * 24), *du = t.get<double>((size_t)n * wb::NU), *da = t.get<double>((size_t)n * 24);
    if (!dq || !dv || !du || !da) return fail(h, ALORE_WB_E_NOMEM, "aba: hipMalloc");
    WB_TRY(h, hipMemcpy(dq, q, sizeof(double) * n * 24, hipMemcpyHostToDevice));
    WB_TRY(h, hipMemcpy(dv, v, sizeof(double) * n * 24, hipMemcpyHostToDevice));
    WB_TRY(h, hipMemcpy(du, u, sizeof(double) * n * wb::NU, hipMemcpyHostToDevice));
    wb::aba_kernel<<<(n + 63) / 64, 64>>>(n, dq, dv, du, b2z1::GRAVITY, da);
    WB_TRY(h, hipGetLastError());
    WB_TRY(h, hipMemcpy(a, da, sizeof(double) * n * 24, hipMemcpyDeviceToHost));
    return ALORE_WB_OK;
}

int alore_wb_forward_dynamics(alore_wb_handle h, int n, const double* q, const double* v, const double* u, double* M, double* a)
{
    if (!h || n <= 0 || !q || !v || !u) return fail(h, ALORE_WB_E_INVALID, "forward_dynamics: bad argument");
    WB_TRY(h, hipSetDevice(h->cfg.device));
    Tmp t;
    // n one-stage problems: x [n][2][48] (second slot unused), u [n][1][30]
    std::vector<double> hx((size_t)n * 2 * wb::NX, 0.0);
    for (int i = 0; i < n; ++i) {
        std::memcpy(&hx[(size_t)i * 2 * wb::NX], q + (size_t)i * 24, sizeof(double) * 24);
        std::memcpy(&hx[(size_t)i * 2 * wb::NX + 24], v + (size_t)i * 24, sizeof(double) * 24);
    }
    double *dx = t.get<double>(hx.size()), *du = t.get<double>((size_t)n * wb::NU), *dn = t.get<double>((size_t)n * wb::NX);
    float *dA = t.get<float>((size_t)n * wb::NX * wb::NX), *dB = t.get<float>((size_t)n * wb::NX * wb::NUP);
    double *dM = M ? t.get<double>((size_t)n * 576) : nullptr, *da = a ? t.get<double>((size_t)n * 24) : nullptr;
    if (!dx || !du || !dn || !dA || !dB || (M && !dM) || (a && !da)) return fail(h, ALORE_WB_E_NOMEM, "forward_dynamics: hipMalloc");
    WB_TRY(h, hipMemcpy(dx, hx.data(), sizeof(double) * hx.size(), hipMemcpyHostToDevice));
    WB_TRY(h, hipMemcpy(du, u, sizeof(double) * n * wb::NU, hipMemcpyHostToDevice));
    wb::StageArgs g{dx, du, 1, n, h->cfg.dt, dA, dB, dn, nullptr, nullptr, dM, da, nullptr};
    wb::stage_kernel<false><<<n, 64>>>(g);
    WB_TRY(h, hipGetLastError());
    if (M) WB_TRY(h, hipMemcpy(M, dM, sizeof(double) * n * 576, hipMemcpyDeviceToHost));
    if (a) WB_TRY(h, hipMemcpy(a, da, sizeof(double) * n * 24, hipMemcpyDeviceToHost));
    WB_TRY(h, hipDeviceSynchronize());
    return ALORE_WB_OK;
}

int alore_wb_set_weights(alore_wb_handle h, const double* Q, const double* R, const double* QN)
{
    if (!h || !Q || !R || !QN) return fail(h, ALORE_WB_E_INVALID, "set_weights: bad argument");
    WB_TRY(h, hipSetDevice(h->cfg.device));
    WB_TRY(h, hipMemcpy(h->d_w, Q, sizeof(double) * wb::NX, hipMemcpyHostToDevice));
    WB_TRY(h, hipMemcpy(h->d_w + wb::NX, R, sizeof(double) * wb::NU, hipMemcpyHostToDevice));
    WB_TRY(h, hipMemcpy(h->d_w + wb::NX + wb::NU, QN, sizeof(double) * wb::NX, hipMemcpyHostToDevice));
    return ALORE_WB_OK;
}

int alore_wb_set_torque_limits(alore_wb_handle h, int enable)
{
    if (!h) return ALORE_WB_E_INVALID;
    h->limits = enable ? 1 : 0;
    return ALORE_WB_OK;
}

int alore_wb_set_contact_constraints(alore_wb_handle h, int enable, double mu)
{
    if (!h || (enable && !(mu > 0.0))) return fail(h, ALORE_WB_E_INVALID, "set_contact_constraints: bad argument");
    h->cones = enable ? 1 : 0;
    if (enable) h->mu = (float)mu;
    return ALORE_WB_OK;
}

int alore_wb_set_contact_penalty(alore_wb_handle h, double rho)
{
    if (!h || !(rho >= 0.0) || !(rho < 1e12)) return fail(h, ALORE_WB_E_INVALID, "set_contact_penalty: rho must be finite and >= 0");
    WB_TRY(h, hipSetDevice(h->cfg.device));
    WB_TRY(h, hipDeviceSynchronize());
    if (rho > 0.0 && !h->d_pen) {
        if (zalloc(&h->d_pen, (size_t)h->cfg.max_problems * h->cfg.horizon * wb::PEN) != hipSuccess) return fail(h, ALORE_WB_E_NOMEM, "set_contact_penalty: device memory");
    }
    h->rho = rho;
    return ALORE_WB_OK;
}

int alore_wb_set_contact_rows(alore_wb_handle h, int enable)
{
    if (!h) return ALORE_WB_E_INVALID;
    WB_TRY(h, hipSetDevice(h->cfg.device));
    const size_t n = (size_t)h->cfg.max_problems * h->cfg.horizon;
    if (enable && !h->d_pen && zalloc(&h->d_pen, n * wb::PEN) != hipSuccess) return fail(h, ALORE_WB_E_NOMEM, "set_contact_rows: device memory");
    if (enable && !h->d_F && zalloc(&h->d_F, n * 12 * wb::FCOL) != hipSuccess) return fail(h, ALORE_WB_E_NOMEM, "set_contact_rows: device memory");
    h->rows = enable ? 1 : 0;
    return ALORE_WB_OK;
}

int alore_wb_set_refinement(alore_wb_handle h, int enable)
{
    if (!h || enable < 0 || enable > 4) return fail(h, ALORE_WB_E_INVALID, "set_refinement: 0 .. 4 steps");
    WB_TRY(h, hipSetDevice(h->cfg.device));
    if (enable && !h->d_res) {
        const size_t B = h->cfg.max_problems, N = h->cfg.horizon;
        if (zalloc(&h->d_res, B * N * wb::VEC) != hipSuccess || zalloc(&h->d_resN, B * wb::NX) != hipSuccess ||
            zalloc(&h->d_ex, B * (N + 1) * wb::NX) != hipSuccess || zalloc(&h->d_eu, B * N * wb::NU) != hipSuccess)
            return fail(h, ALORE_WB_E_NOMEM, "set_refinement: device memory");
    }
    h->refine = enable;
    return ALORE_WB_OK;
}

int alore_wb_set_constraint_mode(alore_wb_handle h, int mode, int max_sweeps)
{
    if (!h || mode < 0 || mode > 1 || (mode == 1 && (max_sweeps < 1 || max_sweeps > 4096)))
        return fail(h, ALORE_WB_E_INVALID, "set_constraint_mode: mode 0 / 1, 1 .. 4096 sweeps");
    WB_TRY(h, hipSetDevice(h->cfg.device));
    WB_TRY(h, hipDeviceSynchronize());
    if (mode == 1 && !h->d_ws) {
        const size_t B = h->cfg.max_problems, n = B * h->cfg.horizon;
        if (zalloc(&h->d_ws, n * 32) != hipSuccess || zalloc(&h->d_Bw, n * wb::NX * wb::NUP) != hipSuccess || zalloc(&h->d_vecw, n * wb::VEC) != hipSuccess ||
            zalloc(&h->d_wrs, n * 32) != hipSuccess || zalloc(&h->d_nextw, n * wb::NX) != hipSuccess || zalloc(&h->d_gu, n * wb::NU) != hipSuccess ||
            zalloc(&h->d_changed, B) != hipSuccess || zalloc(&h->d_total, (size_t)4) != hipSuccess || zalloc(&h->d_nonfull, B) != hipSuccess ||
            zalloc(&h->d_moved, B) != hipSuccess || zalloc(&h->d_z, n * wb::NU) != hipSuccess || zalloc(&h->d_dxz, B * (h->cfg.horizon + 1) * wb::NX) != hipSuccess ||
            zalloc(&h->d_fval, B) != hipSuccess)
            return fail(h, ALORE_WB_E_NOMEM, "set_constraint_mode: device memory");
    }
    if (mode == 1) WB_TRY(h, hipMemset(h->d_ws, 0, (size_t)h->cfg.max_problems * h->cfg.horizon * 32)); // every input free
    h->exact = mode;
    if (mode == 1) h->ws_max = max_sweeps;
    return ALORE_WB_OK;
}

int alore_wb_working_set_info(alore_wb_handle h, int B, int* sweeps, int* changed, unsigned char* ws, double* input_gradients)
{
    if (!h || B <= 0 || B > h->cfg.max_problems || !h->d_ws) return fail(h, ALORE_WB_E_INVALID, "working_set_info: the exact mode has not been enabled");
    WB_TRY(h, hipSetDevice(h->cfg.device));
    WB_TRY(h, hipDeviceSynchronize());
    const size_t n = (size_t)B * h->cfg.horizon;
    if (sweeps) *sweeps = h->ws_sweeps;
    if (changed) WB_TRY(h, hipMemcpy(changed, h->d_moved, sizeof(int) * B, hipMemcpyDeviceToHost)); // 1: still open when the sweeps ran out
    if (ws) WB_TRY(h, hipMemcpy(ws, h->d_ws, n * 32, hipMemcpyDeviceToHost));
    if (input_gradients) WB_TRY(h, hipMemcpy(input_gradients, h->d_gu, sizeof(double) * n * wb::NU, hipMemcpyDeviceToHost));
    return ALORE_WB_OK;
}

int alore_wb_set_contact_schedule(alore_wb_handle h, int B, const unsigned char* stance)
{
    if (!h || B <= 0 || B > h->cfg.max_problems) return fail(h, ALORE_WB_E_INVALID, "set_contact_schedule: bad argument");
    WB_TRY(h, hipSetDevice(h->cfg.device));
    WB_TRY(h, hipDeviceSynchronize());
    if (!stance) { // back to "every foot in contact at every stage"
        if (h->d_stance) { (void)hipFree(h->d_stance); h->d_stance = nullptr; }
        return ALORE_WB_OK;
    }
    const size_t n = (size_t)h->cfg.max_problems * h->cfg.horizon * 4;
    if (!h->d_stance) {
        WB_TRY(h, hipMalloc((void**)&h->d_stance, n));
        WB_TRY(h, hipMemset(h->d_stance, 1, n));
    }
    WB_TRY(h, hipMemcpy(h->d_stance, stance, (size_t)B * h->cfg.horizon * 4, hipMemcpyHostToDevice));
    return ALORE_WB_OK;
}

int alore_wb_set_problem(alore_wb_handle h, int B, const double* x0, const double* xref, const double* uref)
{
    if (!h || B <= 0 || B > h->cfg.max_problems || !x0 || !xref || !uref) return fail(h, ALORE_WB_E_INVALID, "set_problem: bad argument");
    WB_TRY(h, hipSetDevice(h->cfg.device));
    const size_t N = h->cfg.horizon;
    WB_TRY(h, hipDeviceSynchronize()); // work enqueued on the caller's (possibly non-blocking) stream must be done
    WB_TRY(h, hipMemcpy(h->d_x0, x0, sizeof(double) * B * wb::NX, hipMemcpyHostToDevice));
    WB_TRY(h, hipMemcpy(h->d_xref, xref, sizeof(double) * B * (N + 1) * wb::NX, hipMemcpyHostToDevice));
    WB_TRY(h, hipMemcpy(h->d_uref, uref, sizeof(double) * B * N * wb::NU, hipMemcpyHostToDevice));
    return ALORE_WB_OK;
}

int alore_wb_set_x0(alore_wb_handle h, int B, const double* x0)
{
    if (!h || B <= 0 || B > h->cfg.max_problems || !x0) return fail(h, ALORE_WB_E_INVALID, "set_x0: bad argument");
    WB_TRY(h, hipSetDevice(h->cfg.device));
    WB_TRY(h, hipDeviceSynchronize()); // work enqueued on the caller's (possibly non-blocking) stream must be done
    WB_TRY(h, hipMemcpy(h->d_x0, x0, sizeof(double) * B * wb::NX, hipMemcpyHostToDevice));
    return ALORE_WB_OK;
}

int alore_wb_shift_iterate(alore_wb_handle h, int B, void* stream)
{
    if (!h || B <= 0 || B > h->cfg.max_problems) return fail(h, ALORE_WB_E_INVALID, "shift_iterate: bad argument");
    WB_TRY(h, hipSetDevice(h->cfg.device));
    wb::shift_kernel<<<B, 64, 0, (hipStream_t)stream>>>(B, h->cfg.horizon, h->d_x, h->d_u);
    if (h->d_ws) WB_TRY(h, hipMemsetAsync(h->d_ws, 0, (size_t)h->cfg.max_problems * h->cfg.horizon * 32, (hipStream_t)stream));
    WB_TRY(h, hipGetLastError());
    return ALORE_WB_OK;
}

int alore_wb_get_first_input(alore_wb_handle h, int B, double* u0)
{
    if (!h || B <= 0 || B > h->cfg.max_problems || !u0) return fail(h, ALORE_WB_E_INVALID, "get_first_input: bad argument");
    WB_TRY(h, hipSetDevice(h->cfg.device));
    const size_t N = h->cfg.horizon;
    WB_TRY(h, hipDeviceSynchronize()); // work enqueued on the caller's (possibly non-blocking) stream must be done
    WB_TRY(h, hipMemcpy2D(u0, sizeof(double) * wb::NU, h->d_u, sizeof(double) * N * wb::NU, sizeof(double) * wb::NU, B, hipMemcpyDeviceToHost));
    return ALORE_WB_OK;
}

int alore_wb_set_iterate(alore_wb_handle h, int B, const double* x, const double* u)
{
    if (!h || B <= 0 || B > h->cfg.max_problems || !x || !u) return fail(h, ALORE_WB_E_INVALID, "set_iterate: bad argument");
    WB_TRY(h, hipSetDevice(h->cfg.device));
    const size_t N = h->cfg.horizon;
    WB_TRY(h, hipDeviceSynchronize()); // work enqueued on the caller's (possibly non-blocking) stream must be done
    WB_TRY(h, hipMemcpy(h->d_x, x, sizeof(double) * B * (N + 1) * wb::NX, hipMemcpyHostToDevice));
    WB_TRY(h, hipMemcpy(h->d_u, u, sizeof(double) * B * N * wb::NU, hipMemcpyHostToDevice));
    if (h->d_ws) WB_TRY(h, hipMemset(h->d_ws, 0, (size_t)h->cfg.max_problems * N * 32)); // a new iterate: the working set starts free
    return ALORE_WB_OK;
}

int alore_wb_get_iterate(alore_wb_handle h, int B, double* x, double* u)
{
    if (!h || B <= 0 || B > h->cfg.max_problems) return fail(h, ALORE_WB_E_INVALID, "get_iterate: bad argument");
    WB_TRY(h, hipSetDevice(h->cfg.device));
    const size_t N = h->cfg.horizon;
    WB_TRY(h, hipDeviceSynchronize());
    if (x) WB_TRY(h, hipMemcpy(x, h->d_x, sizeof(double) * B * (N + 1) * wb::NX, hipMemcpyDeviceToHost));
    if (u) WB_TRY(h, hipMemcpy(u, h->d_u, sizeof(double) * B * N * wb::NU, hipMemcpyDeviceToHost));
    return ALORE_WB_OK;
}

int alore_wb_linearize(alore_wb_handle h, int B, double* A, double* Bm, double* next)
{
    if (!h || B <= 0 || B > h->cfg.max_problems) return fail(h, ALORE_WB_E_INVALID, "linearize: bad argument");
    WB_TRY(h, hipSetDevice(h->cfg.device));
    const int N = h->cfg.horizon;
    const size_t n = (size_t)B * N;
    Tmp t;
    double *dA = A ? t.get<double>(n * wb::NX * wb::NX) : nullptr, *dB = Bm ? t.get<double>(n * wb::NX * wb::NU) : nullptr;
    if ((A && !dA) || (Bm && !dB)) return fail(h, ALORE_WB_E_NOMEM, "linearize: hipMalloc");
    wb::StageArgs g{h->d_x, h->d_u, N, (int)n, h->cfg.dt, h->d_A, h->d_B, h->d_next, dA, dB, nullptr, nullptr, nullptr};
    WB_TRY(h, hipDeviceSynchronize()); // the iterate may still be written by work on the caller's stream
    wb::stage_kernel<false><<<(unsigned)n, 64>>>(g);
    WB_TRY(h, hipGetLastError());
    if (A) WB_TRY(h, hipMemcpy(A, dA, sizeof(double) * n * wb::NX * wb::NX, hipMemcpyDeviceToHost));
    if (Bm) WB_TRY(h, hipMemcpy(Bm, dB, sizeof(double) * n * wb::NX * wb::NU, hipMemcpyDeviceToHost));
    if (next) WB_TRY(h, hipMemcpy(next, h->d_next, sizeof(double) * n * wb::NX, hipMemcpyDeviceToHost));
    WB_TRY(h, hipDeviceSynchronize());
    return ALORE_WB_OK;
}

int alore_wb_rti(alore_wb_handle h, int B, int n_iter, void* stream)
{
    if (!h || B <= 0 || B > h->cfg.max_problems || n_iter < 1) return fail(h, ALORE_WB_E_INVALID, "rti: bad argument");
    WB_TRY(h, hipSetDevice(h->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    const int N = h->cfg.horizon;
    const size_t n = (size_t)B * N;
    static bool lds_set[16] = {false};
    if (!lds_set[h->cfg.device & 15]) {
        WB_TRY(h, hipFuncSetAttribute((const void*)wb::riccati_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(wb::RicLds)));
        WB_TRY(h, hipFuncSetAttribute((const void*)wb::riccati_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(wb::RicLds)));
        lds_set[h->cfg.device & 15] = true;
    }
    if (h->exact && (h->rows || h->rho > 0.0 || h->refine))
        return fail(h, ALORE_WB_E_INVALID, "rti: the exact working-set mode runs without contact rows, contact penalty and refinement");
    for (int it = 0; it < n_iter; ++it) {
        const bool last = it == n_iter - 1;
        if (last) WB_TRY(h, hipEventRecord(h->ev[0], s));
        wb::StageArgs g{h->d_x, h->d_u, N, (int)n, h->cfg.dt, h->d_A, h->d_B, h->d_next, nullptr, nullptr, nullptr, nullptr, h->d_stamps, h->d_xref, h->d_uref, h->d_vec};
        const bool rows = h->rows && h->d_pen && h->d_F;   // hard contact rows: the stage kernel writes J_c (rho = 1), no penalty
        const bool pen = !rows && h->rho > 0.0 && h->d_pen;
        if (pen) { g.pen = h->d_pen; g.rho = h->rho; g.stance = h->d_stance; }
        if (rows) { g.pen = h->d_pen; g.rho = 1.0; g.stance = h->d_stance; }
        if (pen || rows) wb::stage_kernel<true><<<(unsigned)n, 64, 0, s>>>(g);
        else wb::stage_kernel<false><<<(unsigned)n, 64, 0, s>>>(g);
        if (rows) wb::contact_rows_kernel<<<(unsigned)n, 64, 0, s>>>(h->d_A, h->d_B, h->d_next, h->d_vec, h->d_pen, h->d_stance, h->d_u, N, h->d_F);
        if (last) WB_TRY(h, hipEventRecord(h->ev[1], s));
        // one step of iterative refinement (alore_wb_set_refinement) where nothing is clamped inside the sweep: the float32
        // solution is not applied, its float64 residuals go through the same sweep, the sum is applied
        const bool refine = h->refine && !h->limits && !h->cones && !rows && h->d_res;
        wb::RicArgs r{h->d_A, h->d_B, h->d_next, h->d_x, h->d_u, h->d_x0, h->d_xref, h->d_uref, h->d_w, h->d_K, h->d_kff, h->d_dx, h->d_du, N, (refine || rows) ? 0 : 1, h->d_status, h->limits, h->d_stamps ? h->d_stamps + 32 : nullptr, h->d_vec,
                       rows ? 0 : h->cones, h->mu, h->d_stance};
        if (pen) r.pen = h->d_pen;
        if (h->exact) {
            // exact working-set iteration around the unconstrained sweep (see ws_reduce_kernel); the host reads one counter per sweep
            wb::RicArgs q = r;
            q.B32 = h->d_Bw; q.vec = h->d_vecw; q.next = h->d_nextw; q.wrs = h->d_wrs; q.gu_direct = 1; q.limits = 0; q.cones = 0; q.apply = 0;
            const float mu_c = h->cones ? h->mu : 0.f;
            WB_TRY(h, hipMemsetAsync(h->d_z, 0, sizeof(double) * n * wb::NU, s)); // the iterate is feasible: the iteration starts at the zero step
            WB_TRY(h, hipMemsetAsync(h->d_total, 0, 4 * sizeof(int), s));
            // starting working set: the rows active at the iterate whose multiplier there has the right sign (and the zero-force rows)
            wb::ws_point_kernel<<<B, 64, 0, s>>>(h->d_A, h->d_B, h->d_next, h->d_x, h->d_u, h->d_x0, h->d_xref, h->d_uref, h->d_w, h->d_z, h->d_dxz, h->d_ws, h->d_stance,
                                                 mu_c, N, h->limits, h->cones, h->d_fval, h->d_changed, h->d_total, h->d_gu);
            int sweeps = 0, open = 0;
            for (;;) {
                WB_TRY(h, hipMemsetAsync(h->d_total, 0, 4 * sizeof(int), s));
                wb::ws_reduce_kernel<<<(unsigned)n, 64, 0, s>>>(h->d_B, h->d_vec, h->d_next, h->d_ws, h->d_u, h->d_uref, h->d_w, mu_c, h->d_Bw, h->d_vecw, h->d_nextw, h->d_wrs);
                wb::riccati_kernel<false><<<B, wb::RIC_THREADS, sizeof(wb::RicLds), s>>>(q);
                wb::ws_ratio_kernel<<<B, 256, 0, s>>>(h->d_u, h->d_z, h->d_du, h->d_dxz, h->d_dx, h->d_ws, h->d_stance, mu_c, N, h->limits, h->cones, h->d_nonfull, h->d_total + 1);
                wb::ws_release_kernel<<<B, 64, 0, s>>>(h->d_A, h->d_B, h->d_x, h->d_u, h->d_xref, h->d_uref, h->d_w, h->d_z, h->d_dxz, h->d_ws, h->d_stance, mu_c, N,
                                                       h->d_nonfull, h->d_moved, h->d_total + 2, h->d_gu);
                ++sweeps;
                WB_TRY(h, hipMemcpyAsync(&open, h->d_total + 2, sizeof(int), hipMemcpyDeviceToHost, s));
                WB_TRY(h, hipStreamSynchronize(s));
                if (open == 0 || sweeps >= h->ws_max) break;
            }
            h->ws_sweeps = sweeps;
            // the step of the iteration: (dxz, z) of the last point
            WB_TRY(h, hipMemcpyAsync(h->d_dx, h->d_dxz, sizeof(double) * B * (N + 1) * wb::NX, hipMemcpyDeviceToDevice, s));
            WB_TRY(h, hipMemcpyAsync(h->d_du, h->d_z, sizeof(double) * n * wb::NU, hipMemcpyDeviceToDevice, s));
            WB_TRY(h, hipMemsetAsync(h->d_status, 0, sizeof(int) * B, s));
            wb::ws_apply_kernel<<<B, 256, 0, s>>>(h->d_dx, h->d_du, h->d_x, h->d_u, N, h->d_status, h->d_stance, h->cones ? h->mu : 3.0e38f);
            if (last) WB_TRY(h, hipEventRecord(h->ev[2], s));
            continue;
        }
        if (pen) wb::riccati_kernel<true><<<B, wb::RIC_THREADS, sizeof(wb::RicLds), s>>>(r);
        else wb::riccati_kernel<false><<<B, wb::RIC_THREADS, sizeof(wb::RicLds), s>>>(r);
        if (rows) wb::contact_rows_apply_kernel<<<B, 256, 0, s>>>(h->d_F, h->d_dx, h->d_du, h->d_x, h->d_u, N, 1, h->d_status, h->d_stance);
        for (int rs = 0; refine && rs < h->refine; ++rs) {
            wb::residual_kernel<<<(unsigned)n, 64, 0, s>>>(h->d_A, h->d_B, h->d_next, h->d_x, h->d_u, h->d_xref, h->d_uref, h->d_w, pen ? h->d_pen : nullptr,
                                                           h->d_dx, h->d_du, N, h->d_res, h->d_resN);
            wb::RicArgs c = r;
            c.vec = h->d_res; c.resN = h->d_resN; c.refine = 1; c.dx = h->d_ex; c.du = h->d_eu; c.apply = 0; c.status = nullptr;
            if (pen) wb::riccati_kernel<true><<<B, wb::RIC_THREADS, sizeof(wb::RicLds), s>>>(c);
            else wb::riccati_kernel<false><<<B, wb::RIC_THREADS, sizeof(wb::RicLds), s>>>(c);
            wb::refine_apply_kernel<<<B, 256, 0, s>>>(h->d_dx, h->d_du, h->d_ex, h->d_eu, h->d_x, h->d_u, N, rs == h->refine - 1 ? 1 : 0, h->d_status);
        }
        if (last) WB_TRY(h, hipEventRecord(h->ev[2], s));
    }
    WB_TRY(h, hipGetLastError());
    h->timed = true;
    return ALORE_WB_OK;
}

int alore_wb_last_step(alore_wb_handle h, int B, double* dx, double* du)
{
    if (!h || B <= 0 || B > h->cfg.max_problems) return fail(h, ALORE_WB_E_INVALID, "last_step: bad argument");
    WB_TRY(h, hipSetDevice(h->cfg.device));
    const size_t N = h->cfg.horizon;
    WB_TRY(h, hipDeviceSynchronize());
    if (dx) WB_TRY(h, hipMemcpy(dx, h->d_dx, sizeof(double) * B * (N + 1) * wb::NX, hipMemcpyDeviceToHost));
    if (du) WB_TRY(h, hipMemcpy(du, h->d_du, sizeof(double) * B * N * wb::NU, hipMemcpyDeviceToHost));
    return ALORE_WB_OK;
}

int alore_wb_status(alore_wb_handle h, int B, int* status)
{
    if (!h || B <= 0 || B > h->cfg.max_problems || !status) return fail(h, ALORE_WB_E_INVALID, "status: bad argument");
    WB_TRY(h, hipSetDevice(h->cfg.device));
    WB_TRY(h, hipDeviceSynchronize());
    WB_TRY(h, hipMemcpy(status, h->d_status, sizeof(int) * B, hipMemcpyDeviceToHost));
    return ALORE_WB_OK;
}

int alore_wb_last_times(alore_wb_handle h, float* linearize_ms, float* riccati_ms)
{
    if (!h) return ALORE_WB_E_INVALID;
    if (h->timed) {
        WB_TRY(h, hipEventSynchronize(h->ev[2]));
        WB_TRY(h, hipEventElapsedTime(&h->ms_lin, h->ev[0], h->ev[1]));
        WB_TRY(h, hipEventElapsedTime(&h->ms_ric, h->ev[1], h->ev[2]));
        h->timed = false;
    }
    if (linearize_ms) *linearize_ms = h->ms_lin;
    if (riccati_ms) *riccati_ms = h->ms_ric;
    return ALORE_WB_OK;
}

} // extern "C"
