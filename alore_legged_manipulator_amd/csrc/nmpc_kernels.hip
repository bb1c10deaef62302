// nmpc_kernels.hip -- gfx950 kernels of the batched NMPC real-time iteration.
//
// Mapping.  A workgroup holds G problems; each problem is worked on by a group
// of L consecutive lanes (L = 4..32, a power of two, so a group never straddles
// a 64-lane wavefront).  All per-stage data of a problem live in one LDS row
// laid out field-major ([field][stage]) so that
//   * the stage-parallel phases (linearise, cost, expand) run with lane j of a
//     group on stages j, j+L, ... and touch consecutive LDS banks, and
//   * the two sequential sweeps of the QP (Riccati backward, rollout forward)
//     are executed by every lane of the group on the same addresses (LDS
//     broadcast reads, no conflicts; lane 0 of the group stores).
// Row strides are padded to L mod 32 so that the groups of one 32-lane half
// hit disjoint banks.  Global traffic per tick is the algorithmic minimum:
// every input member is read once, x/u/dual are written once.
//
// Reference for what is computed: see nmpc_core.h.  One launch runs n_sqp
// real-time iterations (acado_preparationStep + acado_feedbackStep,
// CG/acado_solver.c:1057-1077) for every problem of the batch.
#include "nmpc_kernels.h"

#include "nmpc_core.h"

namespace nmpc {

// LDS fields of one problem row; every field has NS = N + 1 slots
enum Field : int {
    F_X0, F_X1, F_X2,   // iterate x      (N+1 nodes)
    F_U0, F_U1,         // iterate u      (N)
    F_Y0, F_Y1,         // QP dual        (N)
    F_A, F_B,           // Gx = I + e(a,b)
    F_B00, F_B01, F_B10, F_B11, F_B20, F_B21, // Gu
    F_D0, F_D1, F_D2,   // shooting defect
    F_Q00, F_Q01, F_Q02, F_Q11, F_Q12, F_Q22, // state Hessian (slot N = terminal)
    F_QV0, F_QV1, F_QV2, // state gradient  (slot N = terminal)
    F_R00, F_R01, F_R11, // control Hessian
    F_RV0, F_RV1,        // control gradient
    F_LB0, F_LB1, F_UB0, F_UB1, // QP bounds on du
    F_ST0, F_ST1,        // working-set status (int bits)
    F_C00, F_C01, F_C02, F_F0, F_C10, F_C11, F_C12, F_E1, F_F1, // policy records
    F_DU0, F_DU1, F_MU0, F_MU1, // step and multipliers
    F_DX0, F_DX1, F_DX2, // state step (N+1 nodes)
    NFIELDS
};

int rti_row_floats(int N) { return NFIELDS * (N + 1); }

bool rti_geometry(int B, int N, int forced_L, int lds_limit_bytes, int n_cu, LaunchGeom* g)
{
    if (B <= 0 || N <= 0) return false;
    int L = forced_L;
    if (L == 0) {
        // aim at >= one 256-thread workgroup of lanes per CU, i.e. B * L >= 256 * n_cu
        long want = (256L * n_cu + B - 1) / B;
        L = 4;
        while (L < 32 && L < want) L *= 2;
    }
    if (L != 4 && L != 8 && L != 16 && L != 32) return false;
    for (; L <= 32; L *= 2) {
        const int row = rti_row_floats(N);
        const int RS = ((row + 31) / 32) * 32 + (L < 32 ? L : 0);
        const long row_bytes = 4L * RS;
        int Gmax = (int)(lds_limit_bytes / row_bytes);
        // threads = G * L must be a multiple of 64 and <= 256
        int threads = 256;
        while (threads >= 64 && threads / L > Gmax) threads -= 64;
        if (threads < 64) {
            if (forced_L) return false;
            continue; // try more lanes per problem (fewer problems per block)
        }
        g->L = L;
        g->G = threads / L;
        g->threads = threads;
        g->grid = (B + g->G - 1) / g->G;
        g->RS = RS;
        g->lds_bytes = (size_t)row_bytes * g->G;
        return true;
    }
    return false;
}

// ---------------------------------------------------------------------------

#define ROW(f, k) row[(f) * NS + (k)]

template <int L>
__global__ __launch_bounds__(256) void rti_kernel(const RtiParams p)
{
    extern __shared__ float lds[];
    const int N = p.N, NS = N + 1;
    const int G = blockDim.x / L;
    const int grp = threadIdx.x / L;
    const int j = threadIdx.x % L;
    const bool writer = (j == 0);
    int prob = blockIdx.x * G + grp;
    const bool valid = prob < p.B;
    if (!valid) prob = p.B - 1; // padding groups shadow the last problem, never store
    float* row = lds + (size_t)grp * p.RS;

    IrkConst K;
    K.h = p.h; K.hh = p.hh; K.c1h = p.c1h; K.c2h = p.c2h;

    const float* gx = p.b.x + (size_t)prob * NS * 3;
    const float* gu = p.b.u + (size_t)prob * N * 2;
    const float* gdual = p.b.dual + (size_t)prob * N * 2;
    const float* god = p.b.od + (size_t)prob * NS * 3;
    const float* gy = p.b.y + (size_t)prob * N * 5;
    const float* gyN = p.b.yN + (size_t)prob * 3;
    const float* gW = p.b.W + (size_t)prob * N * 25;
    const float* gWN = p.b.WN + (size_t)prob * 9;
    const float* glb = p.b.lbValues + (size_t)prob * N * 2;
    const float* gub = p.b.ubValues + (size_t)prob * N * 2;

    // ---- phase 0: iterate -> LDS
    for (int k = j; k <= N; k += L) {
        ROW(F_X0, k) = gx[k * 3];
        ROW(F_X1, k) = gx[k * 3 + 1];
        ROW(F_X2, k) = gx[k * 3 + 2];
        if (k < N) {
            ROW(F_U0, k) = gu[k * 2];
            ROW(F_U1, k) = gu[k * 2 + 1];
            ROW(F_Y0, k) = gdual[k * 2];
            ROW(F_Y1, k) = gdual[k * 2 + 1];
        }
    }
    const float x00 = p.b.x0[(size_t)prob * 3], x01 = p.b.x0[(size_t)prob * 3 + 1],
                x02 = p.b.x0[(size_t)prob * 3 + 2];
    __syncthreads();

    int status = RET_OK, n_iter = 0;
    float kkt = 0.0f, obj = 0.0f;

    for (int sqp = 0; sqp < p.n_sqp; ++sqp) {
        // ---- phase A (stage-parallel): linearise, Gauss-Newton cost, bounds, working-set guess
        bool infeasible = false;
        for (int k = j; k <= N; k += L) {
            const float px = ROW(F_X0, k), py = ROW(F_X1, k), ps = ROW(F_X2, k);
            if (k < N) {
                const float vr = ROW(F_U0, k), vl = ROW(F_U1, k);
                StageLin lin;
                ddr_linearize(K, px, py, ps, vr, vl, god[k * 3], god[k * 3 + 1], god[k * 3 + 2], lin);
                ROW(F_A, k) = lin.a;
                ROW(F_B, k) = lin.b;
                ROW(F_B00, k) = lin.B00; ROW(F_B01, k) = lin.B01;
                ROW(F_B10, k) = lin.B10; ROW(F_B11, k) = lin.B11;
                ROW(F_B20, k) = lin.B20; ROW(F_B21, k) = lin.B21;
                ROW(F_D0, k) = lin.phi0 - ROW(F_X0, k + 1);
                ROW(F_D1, k) = lin.phi1 - ROW(F_X1, k + 1);
                ROW(F_D2, k) = lin.phi2 - ROW(F_X2, k + 1);
                // Dy = h(x,u) - y ; gradient = W[rows] * Dy ; Hessian blocks of W
                const float* yk = gy + k * 5;
                const float e0 = px - yk[0], e1 = py - yk[1], e2 = ps - yk[2], e3 = vr - yk[3], e4 = vl - yk[4];
                const float* Wk = gW + k * 25;
                float w[25];
#pragma unroll
                for (int i = 0; i < 25; ++i) w[i] = Wk[i];
                ROW(F_QV0, k) = w[0] * e0 + w[1] * e1 + w[2] * e2 + w[3] * e3 + w[4] * e4;
                ROW(F_QV1, k) = w[5] * e0 + w[6] * e1 + w[7] * e2 + w[8] * e3 + w[9] * e4;
                ROW(F_QV2, k) = w[10] * e0 + w[11] * e1 + w[12] * e2 + w[13] * e3 + w[14] * e4;
                ROW(F_RV0, k) = w[15] * e0 + w[16] * e1 + w[17] * e2 + w[18] * e3 + w[19] * e4;
                ROW(F_RV1, k) = w[20] * e0 + w[21] * e1 + w[22] * e2 + w[23] * e3 + w[24] * e4;
                ROW(F_Q00, k) = w[0]; ROW(F_Q01, k) = w[1]; ROW(F_Q02, k) = w[2];
                ROW(F_Q11, k) = w[6]; ROW(F_Q12, k) = w[7]; ROW(F_Q22, k) = w[12];
                ROW(F_R00, k) = w[18]; ROW(F_R01, k) = w[19]; ROW(F_R11, k) = w[24];
                const float lb0 = glb[k * 2] - vr, lb1 = glb[k * 2 + 1] - vl;
                const float ub0 = gub[k * 2] - vr, ub1 = gub[k * 2 + 1] - vl;
                ROW(F_LB0, k) = lb0; ROW(F_LB1, k) = lb1;
                ROW(F_UB0, k) = ub0; ROW(F_UB1, k) = ub1;
                infeasible = infeasible || (lb0 > ub0 + 1e-6f) || (lb1 > ub1 + 1e-6f);
                ROW(F_ST0, k) = __int_as_float(status_from_dual(ROW(F_Y0, k), lb0, ub0));
                ROW(F_ST1, k) = __int_as_float(status_from_dual(ROW(F_Y1, k), lb1, ub1));
            } else {
                const float e0 = px - gyN[0], e1 = py - gyN[1], e2 = ps - gyN[2];
                float w[9];
#pragma unroll
                for (int i = 0; i < 9; ++i) w[i] = gWN[i];
                ROW(F_QV0, k) = w[0] * e0 + w[1] * e1 + w[2] * e2;
                ROW(F_QV1, k) = w[3] * e0 + w[4] * e1 + w[5] * e2;
                ROW(F_QV2, k) = w[6] * e0 + w[7] * e1 + w[8] * e2;
                ROW(F_Q00, k) = w[0]; ROW(F_Q01, k) = w[1]; ROW(F_Q02, k) = w[2];
                ROW(F_Q11, k) = w[4]; ROW(F_Q12, k) = w[5]; ROW(F_Q22, k) = w[8];
            }
        }
        { // a lane only saw its own stages: OR the flag over the group
            int inf = infeasible ? 1 : 0;
#pragma unroll
            for (int off = L / 2; off > 0; off >>= 1) inf |= __shfl_xor(inf, off, L);
            infeasible = inf != 0;
        }
        __syncthreads();

        const float Dx0 = x00 - ROW(F_X0, 0), Dx1 = x01 - ROW(F_X1, 0), Dx2 = x02 - ROW(F_X2, 0);

        // ---- phase B: working-set iterations, every lane of the group runs both sweeps
        bool pd_fail = false, changed = false;
        int it = 0;
        n_iter = 0;
        for (;;) {
            // backward Riccati sweep
            Value V;
            V.P.m00 = ROW(F_Q00, N); V.P.m01 = ROW(F_Q01, N); V.P.m02 = ROW(F_Q02, N);
            V.P.m11 = ROW(F_Q11, N); V.P.m12 = ROW(F_Q12, N); V.P.m22 = ROW(F_Q22, N);
            V.p0 = ROW(F_QV0, N); V.p1 = ROW(F_QV1, N); V.p2 = ROW(F_QV2, N);
            bool ok = true;
            for (int k = N - 1; k >= 0; --k) {
                StageQP s;
                s.a = ROW(F_A, k); s.b = ROW(F_B, k);
                s.B00 = ROW(F_B00, k); s.B01 = ROW(F_B01, k); s.B10 = ROW(F_B10, k);
                s.B11 = ROW(F_B11, k); s.B20 = ROW(F_B20, k); s.B21 = ROW(F_B21, k);
                s.d0 = ROW(F_D0, k); s.d1 = ROW(F_D1, k); s.d2 = ROW(F_D2, k);
                s.Q.m00 = ROW(F_Q00, k); s.Q.m01 = ROW(F_Q01, k); s.Q.m02 = ROW(F_Q02, k);
                s.Q.m11 = ROW(F_Q11, k); s.Q.m12 = ROW(F_Q12, k); s.Q.m22 = ROW(F_Q22, k);
                s.q0 = ROW(F_QV0, k); s.q1 = ROW(F_QV1, k); s.q2 = ROW(F_QV2, k);
                s.R00 = ROW(F_R00, k); s.R01 = ROW(F_R01, k); s.R11 = ROW(F_R11, k);
                s.r0 = ROW(F_RV0, k); s.r1 = ROW(F_RV1, k);
                s.st0 = __float_as_int(ROW(F_ST0, k));
                s.st1 = __float_as_int(ROW(F_ST1, k));
                s.v0 = (s.st0 == ST_UPPER) ? ROW(F_UB0, k) : ROW(F_LB0, k);
                s.v1 = (s.st1 == ST_UPPER) ? ROW(F_UB1, k) : ROW(F_LB1, k);
                Policy pol;
                ok = riccati_step(s, V, pol, k > 0) && ok;
                if (writer) {
                    ROW(F_C00, k) = pol.c00; ROW(F_C01, k) = pol.c01; ROW(F_C02, k) = pol.c02;
                    ROW(F_F0, k) = pol.f0;
                    ROW(F_C10, k) = pol.c10; ROW(F_C11, k) = pol.c11; ROW(F_C12, k) = pol.c12;
                    ROW(F_E1, k) = pol.e1; ROW(F_F1, k) = pol.f1;
                }
            }
            pd_fail = pd_fail || !ok;
            __syncthreads();

            // forward sweep: step, multipliers, new working set, KKT terms
            float dx0 = Dx0, dx1 = Dx1, dx2 = Dx2; // full step  dx_k
            float sb0 = Dx0, sb1 = Dx1, sb2 = Dx2; // free response sbar_k (du = 0)
            float gd = 0.0f, comp = 0.0f;
            changed = false;
            if (writer) { ROW(F_DX0, 0) = dx0; ROW(F_DX1, 0) = dx1; ROW(F_DX2, 0) = dx2; }
            for (int k = 0; k < N; ++k) {
                Policy pol;
                pol.c00 = ROW(F_C00, k); pol.c01 = ROW(F_C01, k); pol.c02 = ROW(F_C02, k);
                pol.f0 = ROW(F_F0, k);
                pol.c10 = ROW(F_C10, k); pol.c11 = ROW(F_C11, k); pol.c12 = ROW(F_C12, k);
                pol.e1 = ROW(F_E1, k); pol.f1 = ROW(F_F1, k);
                const int st0 = __float_as_int(ROW(F_ST0, k)), st1 = __float_as_int(ROW(F_ST1, k));
                const float lb0 = ROW(F_LB0, k), lb1 = ROW(F_LB1, k), ub0 = ROW(F_UB0, k), ub1 = ROW(F_UB1, k);
                StageStep o;
                forward_step(pol, st0, st1, dx0, dx1, dx2, lb0, ub0, lb1, ub1, o);
                changed = changed || (o.nst0 != st0) || (o.nst1 != st1);
                if (k > 0) {
                    // (Q_k sbar_k + q_k)' (dx_k - sbar_k)
                    const float t0 = dx0 - sb0, t1 = dx1 - sb1, t2 = dx2 - sb2;
                    const float Q00 = ROW(F_Q00, k), Q01 = ROW(F_Q01, k), Q02 = ROW(F_Q02, k),
                                Q11 = ROW(F_Q11, k), Q12 = ROW(F_Q12, k), Q22 = ROW(F_Q22, k);
                    gd += (Q00 * sb0 + Q01 * sb1 + Q02 * sb2 + ROW(F_QV0, k)) * t0 +
                          (Q01 * sb0 + Q11 * sb1 + Q12 * sb2 + ROW(F_QV1, k)) * t1 +
                          (Q02 * sb0 + Q12 * sb1 + Q22 * sb2 + ROW(F_QV2, k)) * t2;
                }
                gd += ROW(F_RV0, k) * o.du0 + ROW(F_RV1, k) * o.du1;
                comp += (o.mu0 > 1e-12f) ? fabsf(lb0 * o.mu0) : ((o.mu0 < -1e-12f) ? fabsf(ub0 * o.mu0) : 0.0f);
                comp += (o.mu1 > 1e-12f) ? fabsf(lb1 * o.mu1) : ((o.mu1 < -1e-12f) ? fabsf(ub1 * o.mu1) : 0.0f);
                const float a = ROW(F_A, k), b = ROW(F_B, k);
                const float B00 = ROW(F_B00, k), B01 = ROW(F_B01, k), B10 = ROW(F_B10, k),
                            B11 = ROW(F_B11, k), B20 = ROW(F_B20, k), B21 = ROW(F_B21, k);
                const float d0 = ROW(F_D0, k), d1 = ROW(F_D1, k), d2 = ROW(F_D2, k);
                const float n0 = dx0 + a * dx2 + B00 * o.du0 + B01 * o.du1 + d0;
                const float n1 = dx1 + b * dx2 + B10 * o.du0 + B11 * o.du1 + d1;
                const float n2 = dx2 + B20 * o.du0 + B21 * o.du1 + d2;
                const float m0 = sb0 + a * sb2 + d0, m1 = sb1 + b * sb2 + d1, m2 = sb2 + d2;
                dx0 = n0; dx1 = n1; dx2 = n2;
                sb0 = m0; sb1 = m1; sb2 = m2;
                if (writer) {
                    ROW(F_DU0, k) = o.du0; ROW(F_DU1, k) = o.du1;
                    ROW(F_MU0, k) = o.mu0; ROW(F_MU1, k) = o.mu1;
                    ROW(F_DX0, k + 1) = dx0; ROW(F_DX1, k + 1) = dx1; ROW(F_DX2, k + 1) = dx2;
                    ROW(F_ST0, k) = __int_as_float(o.nst0);
                    ROW(F_ST1, k) = __int_as_float(o.nst1);
                }
            }
            { // terminal term of g' du
                const float t0 = dx0 - sb0, t1 = dx1 - sb1, t2 = dx2 - sb2;
                const float Q00 = ROW(F_Q00, N), Q01 = ROW(F_Q01, N), Q02 = ROW(F_Q02, N), Q11 = ROW(F_Q11, N),
                            Q12 = ROW(F_Q12, N), Q22 = ROW(F_Q22, N);
                gd += (Q00 * sb0 + Q01 * sb1 + Q02 * sb2 + ROW(F_QV0, N)) * t0 +
                      (Q01 * sb0 + Q11 * sb1 + Q12 * sb2 + ROW(F_QV1, N)) * t1 +
                      (Q02 * sb0 + Q12 * sb1 + Q22 * sb2 + ROW(F_QV2, N)) * t2;
            }
            kkt = fabsf(gd) + comp;
            ++it;
            if (changed) n_iter = it;
            // block-uniform continuation: any problem of the block still changing?
            const int more = __syncthreads_or((changed && it < p.max_as_iter) ? 1 : 0);
            if (!more) break;
        }
        if (n_iter == 0) n_iter = 1;
        else if (!changed) n_iter += 1; // the confirming sweep
        status = infeasible ? RET_INIT_FAILED_INFEASIBILITY
                            : (pd_fail ? RET_INIT_FAILED_CHOLESKY : (changed ? RET_MAX_NWSR_REACHED : RET_OK));

        // ---- phase C (stage-parallel): expand (acado_expand) and carry the dual
        for (int k = j; k <= N; k += L) {
            ROW(F_X0, k) += ROW(F_DX0, k);
            ROW(F_X1, k) += ROW(F_DX1, k);
            ROW(F_X2, k) += ROW(F_DX2, k);
            if (k < N) {
                const float lb0 = ROW(F_LB0, k), lb1 = ROW(F_LB1, k), ub0 = ROW(F_UB0, k), ub1 = ROW(F_UB1, k);
                // a free control may sit up to TOL_PRIMAL outside its box: keep the iterate feasible
                ROW(F_U0, k) += (lb0 <= ub0) ? clampf(ROW(F_DU0, k), lb0, ub0) : ROW(F_DU0, k);
                ROW(F_U1, k) += (lb1 <= ub1) ? clampf(ROW(F_DU1, k), lb1, ub1) : ROW(F_DU1, k);
                ROW(F_Y0, k) = ROW(F_MU0, k);
                ROW(F_Y1, k) = ROW(F_MU1, k);
            }
        }
        __syncthreads();
    }

    // ---- objective at the returned iterate (acado_getObjective) + write-back
    float part = 0.0f;
    for (int k = j; k <= N; k += L) {
        const float px = ROW(F_X0, k), py = ROW(F_X1, k), ps = ROW(F_X2, k);
        if (k < N) {
            const float vr = ROW(F_U0, k), vl = ROW(F_U1, k);
            const float* yk = gy + k * 5;
            const float* Wk = gW + k * 25;
            float e[5] = {px - yk[0], py - yk[1], ps - yk[2], vr - yk[3], vl - yk[4]};
            float acc = 0.0f;
#pragma unroll
            for (int c = 0; c < 5; ++c) {
                const float t = e[0] * Wk[c] + e[1] * Wk[5 + c] + e[2] * Wk[10 + c] + e[3] * Wk[15 + c] +
                                e[4] * Wk[20 + c];
                acc += e[c] * t;
            }
            part += acc;
        } else { // the reference uses only the diagonal of WN here (acado_solver.c:1442-1444)
            const float e0 = px - gyN[0], e1 = py - gyN[1], e2 = ps - gyN[2];
            part += e0 * e0 * gWN[0] + e1 * e1 * gWN[4] + e2 * e2 * gWN[8];
        }
        if (valid) {
            float* ox = p.b.x + (size_t)prob * NS * 3;
            ox[k * 3] = px; ox[k * 3 + 1] = py; ox[k * 3 + 2] = ps;
            if (k < N) {
                float* ou = p.b.u + (size_t)prob * N * 2;
                float* od = p.b.dual + (size_t)prob * N * 2;
                ou[k * 2] = ROW(F_U0, k); ou[k * 2 + 1] = ROW(F_U1, k);
                od[k * 2] = ROW(F_Y0, k); od[k * 2 + 1] = ROW(F_Y1, k);
            }
        }
    }
#pragma unroll
    for (int off = L / 2; off > 0; off >>= 1) part += __shfl_xor(part, off, L);
    obj = 0.5f * part;
    if (valid && writer) {
        p.b.status[prob] = status;
        p.b.n_iter[prob] = n_iter;
        p.b.kkt[prob] = kkt;
        p.b.obj[prob] = obj;
    }
}

hipError_t launch_rti(const RtiParams& p, const LaunchGeom& g, hipStream_t s)
{
    dim3 grid(g.grid), block(g.threads);
    hipError_t e = hipSuccess;
    switch (g.L) {
#define CASE(LL)                                                                                             \
    case LL: {                                                                                               \
        static size_t configured = 0; /* raise the dynamic-LDS cap once per size, not per launch */          \
        if (g.lds_bytes > configured) {                                                                      \
            e = hipFuncSetAttribute((const void*)rti_kernel<LL>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    (int)g.lds_bytes);                                                       \
            if (e != hipSuccess) return e;                                                                   \
            configured = g.lds_bytes;                                                                        \
        }                                                                                                    \
        hipLaunchKernelGGL(rti_kernel<LL>, grid, block, g.lds_bytes, s, p);                                  \
        break;                                                                                               \
    }
        CASE(4)
        CASE(8)
        CASE(16)
        CASE(32)
#undef CASE
    default:
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// small stage-parallel kernels around the hot path (one thread per (problem, stage))

__global__ void linearize_kernel(alore_nmpc_batch b, int B, int N, IrkConst K, alore_nmpc_lin_out o)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)B * N) return;
    const int prob = (int)(t / N), k = (int)(t % N);
    const float* x = b.x + ((size_t)prob * (N + 1) + k) * 3;
    const float* u = b.u + ((size_t)prob * N + k) * 2;
    const float* od = b.od + ((size_t)prob * (N + 1) + k) * 3;
    StageLin lin;
    ddr_linearize(K, x[0], x[1], x[2], u[0], u[1], od[0], od[1], od[2], lin);
    if (o.d) {
        float* d = o.d + (size_t)t * 3;
        d[0] = lin.phi0 - x[3]; d[1] = lin.phi1 - x[4]; d[2] = lin.phi2 - x[5];
    }
    if (o.evGx) {
        float* g = o.evGx + (size_t)t * 9;
        g[0] = 1.f; g[1] = 0.f; g[2] = lin.a;
        g[3] = 0.f; g[4] = 1.f; g[5] = lin.b;
        g[6] = 0.f; g[7] = 0.f; g[8] = 1.f;
    }
    if (o.evGu) {
        float* g = o.evGu + (size_t)t * 6;
        g[0] = lin.B00; g[1] = lin.B01; g[2] = lin.B10; g[3] = lin.B11; g[4] = lin.B20; g[5] = lin.B21;
    }
}

hipError_t launch_linearize(const alore_nmpc_batch& b, int B, int N, float dt, const alore_nmpc_lin_out& o,
                            hipStream_t s)
{
    const long total = (long)B * N;
    const int threads = 256;
    hipLaunchKernelGGL(linearize_kernel, dim3((unsigned)((total + threads - 1) / threads)), dim3(threads), 0, s, b,
                       B, N, make_irk(dt), o);
    return hipGetLastError();
}

// sequential in k, one thread per problem (acado_initializeNodesByForwardSimulation)
__global__ void forward_sim_kernel(alore_nmpc_batch b, int B, int N, IrkConst K)
{
    const int prob = blockIdx.x * blockDim.x + threadIdx.x;
    if (prob >= B) return;
    float* x = b.x + (size_t)prob * (N + 1) * 3;
    const float* u = b.u + (size_t)prob * N * 2;
    const float* od = b.od + (size_t)prob * (N + 1) * 3;
    float px = x[0], py = x[1], ps = x[2];
    for (int k = 0; k < N; ++k) {
        StageLin lin;
        ddr_linearize(K, px, py, ps, u[k * 2], u[k * 2 + 1], od[k * 3], od[k * 3 + 1], od[k * 3 + 2], lin);
        px = lin.phi0; py = lin.phi1; ps = lin.phi2;
        x[(k + 1) * 3] = px; x[(k + 1) * 3 + 1] = py; x[(k + 1) * 3 + 2] = ps;
    }
}

hipError_t launch_forward_simulate(const alore_nmpc_batch& b, int B, int N, float dt, hipStream_t s)
{
    hipLaunchKernelGGL(forward_sim_kernel, dim3((B + 63) / 64), dim3(64), 0, s, b, B, N, make_irk(dt));
    return hipGetLastError();
}

// acado_shiftStates + acado_shiftControls, one thread per problem
__global__ void shift_kernel(alore_nmpc_batch b, int B, int N, IrkConst K, int strategy, const float* xEnd,
                             const float* uEnd)
{
    const int prob = blockIdx.x * blockDim.x + threadIdx.x;
    if (prob >= B) return;
    float* x = b.x + (size_t)prob * (N + 1) * 3;
    float* u = b.u + (size_t)prob * N * 2;
    const float* od = b.od + (size_t)prob * (N + 1) * 3;
    for (int i = 0; i < 3 * N; ++i) x[i] = x[i + 3];
    if (strategy == 1 && xEnd) {
        for (int i = 0; i < 3; ++i) x[3 * N + i] = xEnd[(size_t)prob * 3 + i];
    } else if (strategy == 2) {
        const float vr = uEnd ? uEnd[(size_t)prob * 2] : u[(N - 1) * 2];
        const float vl = uEnd ? uEnd[(size_t)prob * 2 + 1] : u[(N - 1) * 2 + 1];
        StageLin lin;
        ddr_linearize(K, x[3 * N], x[3 * N + 1], x[3 * N + 2], vr, vl, od[3 * N], od[3 * N + 1], od[3 * N + 2], lin);
        x[3 * N] = lin.phi0; x[3 * N + 1] = lin.phi1; x[3 * N + 2] = lin.phi2;
    }
    for (int i = 0; i < 2 * (N - 1); ++i) u[i] = u[i + 2];
    if (uEnd) {
        u[2 * (N - 1)] = uEnd[(size_t)prob * 2];
        u[2 * (N - 1) + 1] = uEnd[(size_t)prob * 2 + 1];
    }
}

hipError_t launch_shift(const alore_nmpc_batch& b, int B, int N, float dt, int strategy, const float* xEnd,
                        const float* uEnd, hipStream_t s)
{
    hipLaunchKernelGGL(shift_kernel, dim3((B + 63) / 64), dim3(64), 0, s, b, B, N, make_irk(dt), strategy, xEnd, uEnd);
    return hipGetLastError();
}

__global__ void fill_kernel(float* p, float v, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}

hipError_t launch_fill(float* p, float v, size_t n, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    size_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, v, n);
    return hipGetLastError();
}

} // namespace nmpc
