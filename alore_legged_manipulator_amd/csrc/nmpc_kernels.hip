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

// LDS layout.  One problem = N+1 stage records of SR floats (19 float4 slots;
// the odd slot count spreads the records of consecutive stages over all banks
// for 16-byte accesses), record N being the terminal node.
//   slot 0  a b B00 B01        slot 7  st0 st1 du0 du1      slot 13 dx0 dx1 dx2 p2
//   slot 1  B10 B11 B20 d0     slot 8  c00 c01 c02 f0       slot 14 x0 x1 x2 -
//   slot 2  d1 d2 q0 q1        slot 9  c10 c11 c12 e1       slot 15 u0 u1 y0 y1
//   slot 3  q2 r0 r1 R00       slot 10 mu0 mu1 f1 -         slot 16 sb0 sb1 sb2 -
//   slot 4  R01 R11 Q00 Q01    slot 11 P00 P01 P02 P11      slot 17-18 spare
//   slot 5  Q02 Q11 Q12 Q22    slot 12 P12 P22 p0 p1
//   slot 6  lb0 ub0 lb1 ub1
// (P, p) of record k is the cost-to-go at node k under the working set of the
// last backward sweep; it lets a later sweep restart below the highest stage
// whose working set changed instead of at the horizon end.
constexpr int SLOTS = 19;
constexpr int SR = SLOTS * 4;
enum Slot : int {
    S_LIN0 = 0, S_LIN1 = 1, S_DQ = 2, S_QR = 3, S_RQ = 4, S_QQ = 5, S_BND = 6, S_STDU = 7,
    S_POL0 = 8, S_POL1 = 9, S_MU = 10, S_V0 = 11, S_V1 = 12, S_DX = 13, S_X = 14, S_UY = 15, S_SB = 16
};

int rti_row_floats(int N) { return SR * (N + 1); }

bool rti_geometry(int B, int N, int forced_L, int lds_limit_bytes, int n_cu, LaunchGeom* g)
{
    if (B <= 0 || N <= 0) return false;
    int L = forced_L;
    if (L == 0) {
        // aim at >= one wavefront per SIMD over the whole chip: B * L >= 256 * n_cu lanes
        long want = (256L * n_cu + B - 1) / B;
        L = 4;
        while (L < 64 && L < want) L *= 2;
        // the forward scan works on chunks of L stages: with lanes to spare, cover the horizon in one chunk
        while (L < 64 && L < N && (long)B * L * 2 <= 512L * n_cu) L *= 2;
    }
    if (L != 4 && L != 8 && L != 16 && L != 32 && L != 64) return false;
    for (; L <= 64; L *= 2) {
        // pad the row so that RS = 4 (mod 64): the G rows of one wavefront start one 16-byte slot apart
        const int base = rti_row_floats(N);
        const int RS = base + ((4 - base % 64) + 64) % 64;
        const long row_bytes = 4L * RS;
        const int G = 64 / L; // one wavefront per workgroup
        if ((long)G * row_bytes > lds_limit_bytes) {
            if (forced_L) return false;
            continue; // more lanes per problem = fewer problems per wavefront
        }
        g->L = L;
        g->G = G;
        g->threads = 64;
        g->grid = (B + G - 1) / G;
        g->RS = RS;
        g->lds_bytes = (size_t)row_bytes * G;
        return true;
    }
    return false;
}

// ---------------------------------------------------------------------------

struct StageRegs { // slots 0..6 of a stage record
    float4 l0, l1, dq, qr, rq, qq, bnd;
};

__device__ __forceinline__ float4 lds4(const float* rec, int slot)
{
    return *reinterpret_cast<const float4*>(rec + slot * 4);
}
__device__ __forceinline__ void st4(float* rec, int slot, float a, float b, float c, float d)
{
    *reinterpret_cast<float4*>(rec + slot * 4) = make_float4(a, b, c, d);
}

__device__ __forceinline__ void load_stage(const float* rec, StageRegs& r)
{
    r.l0 = lds4(rec, S_LIN0); r.l1 = lds4(rec, S_LIN1); r.dq = lds4(rec, S_DQ); r.qr = lds4(rec, S_QR);
    r.rq = lds4(rec, S_RQ); r.qq = lds4(rec, S_QQ); r.bnd = lds4(rec, S_BND);
}

template <int L>
__device__ __forceinline__ float group_sum(float v)
{
#pragma unroll
    for (int off = L / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, L);
    return v;
}
template <int L>
__device__ __forceinline__ int group_or(int v)
{
#pragma unroll
    for (int off = L / 2; off > 0; off >>= 1) v |= __shfl_xor(v, off, L);
    return v;
}
__device__ __forceinline__ int wave_max(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
    return v;
}

// one wavefront per workgroup, G = 64 / L problems per wavefront
template <int L, bool STAMP>
__global__ __launch_bounds__(64) void rti_kernel(const RtiParams p)
{
    extern __shared__ float4 lds_raw[];
    float* lds = reinterpret_cast<float*>(lds_raw);
    const int N = p.N;
    constexpr int G = 64 / L;
    const int grp = threadIdx.x / L;
    const int j = threadIdx.x % L;
    const bool writer = (j == 0);
    int prob = blockIdx.x * G + grp;
    const bool valid = prob < p.B;
    if (!valid) prob = p.B - 1; // padding groups shadow the last problem, never store
    float* row = lds + (size_t)grp * p.RS;

    long long t_stamp[6];
    if (STAMP) t_stamp[0] = __builtin_amdgcn_s_memtime();

    IrkConst K;
    K.h = p.h; K.hh = p.hh; K.c1h = p.c1h; K.c2h = p.c2h;

    const float* gx = p.b.x + (size_t)prob * (N + 1) * 3;
    const float* gu = p.b.u + (size_t)prob * N * 2;
    const float* gdual = p.b.dual + (size_t)prob * N * 2;
    const float* god = p.b.od + (size_t)prob * (N + 1) * 3;
    const float* gy = p.b.y + (size_t)prob * N * 5;
    const float* gyN = p.b.yN + (size_t)prob * 3;
    const float* gW = p.b.W + (size_t)prob * N * 25;
    const float* gWN = p.b.WN + (size_t)prob * 9;
    const float* glb = p.b.lbValues + (size_t)prob * N * 2;
    const float* gub = p.b.ubValues + (size_t)prob * N * 2;

    // ---- phase 0: iterate -> LDS
    for (int k = j; k <= N; k += L) {
        float* rec = row + k * SR;
        st4(rec, S_X, gx[k * 3], gx[k * 3 + 1], gx[k * 3 + 2], 0.0f);
        if (k < N) st4(rec, S_UY, gu[k * 2], gu[k * 2 + 1], gdual[k * 2], gdual[k * 2 + 1]);
    }
    const float x00 = p.b.x0[(size_t)prob * 3], x01 = p.b.x0[(size_t)prob * 3 + 1],
                x02 = p.b.x0[(size_t)prob * 3 + 2];
    __syncthreads();

    int status = RET_OK, n_iter = 0;
    float kkt = 0.0f;

    for (int sqp = 0; sqp < p.n_sqp; ++sqp) {
        // ---- phase A (stage-parallel): linearise, Gauss-Newton cost, bounds, working-set guess
        int infeasible = 0;
        for (int k = j; k <= N; k += L) {
            float* rec = row + k * SR;
            const float4 xk = lds4(rec, S_X);
            if (k < N) {
                const float4 uy = lds4(rec, S_UY);
                const float4 xn = lds4(rec + SR, S_X);
                const float vr = uy.x, vl = uy.y;
                StageLin lin;
                ddr_linearize(K, xk.x, xk.y, xk.z, vr, vl, god[k * 3], god[k * 3 + 1], god[k * 3 + 2], lin);
                const float d0 = lin.phi0 - xn.x, d1 = lin.phi1 - xn.y, d2 = lin.phi2 - xn.z;
                // Dy = h(x,u) - y ; gradient = W[rows] * Dy ; Hessian blocks of W
                const float* yk = gy + k * 5;
                const float e0 = xk.x - yk[0], e1 = xk.y - yk[1], e2 = xk.z - yk[2], e3 = vr - yk[3], e4 = vl - yk[4];
                const float* Wk = gW + k * 25;
                float w[25];
#pragma unroll
                for (int i = 0; i < 25; ++i) w[i] = Wk[i];
                const float q0 = w[0] * e0 + w[1] * e1 + w[2] * e2 + w[3] * e3 + w[4] * e4;
                const float q1 = w[5] * e0 + w[6] * e1 + w[7] * e2 + w[8] * e3 + w[9] * e4;
                const float q2 = w[10] * e0 + w[11] * e1 + w[12] * e2 + w[13] * e3 + w[14] * e4;
                const float r0 = w[15] * e0 + w[16] * e1 + w[17] * e2 + w[18] * e3 + w[19] * e4;
                const float r1 = w[20] * e0 + w[21] * e1 + w[22] * e2 + w[23] * e3 + w[24] * e4;
                const float lb0 = glb[k * 2] - vr, lb1 = glb[k * 2 + 1] - vl;
                const float ub0 = gub[k * 2] - vr, ub1 = gub[k * 2 + 1] - vl;
                infeasible |= (lb0 > ub0 + 1e-6f) || (lb1 > ub1 + 1e-6f);
                st4(rec, S_LIN0, lin.a, lin.b, lin.B00, lin.B01);
                st4(rec, S_LIN1, lin.B10, lin.B11, lin.B20, d0);
                st4(rec, S_DQ, d1, d2, q0, q1);
                st4(rec, S_QR, q2, r0, r1, w[18]);
                st4(rec, S_RQ, w[19], w[24], w[0], w[1]);
                st4(rec, S_QQ, w[2], w[6], w[7], w[12]);
                st4(rec, S_BND, lb0, ub0, lb1, ub1);
                st4(rec, S_STDU, __int_as_float(status_from_dual(uy.z, lb0, ub0)),
                    __int_as_float(status_from_dual(uy.w, lb1, ub1)), 0.0f, 0.0f);
            } else {
                const float e0 = xk.x - gyN[0], e1 = xk.y - gyN[1], e2 = xk.z - gyN[2];
                float w[9];
#pragma unroll
                for (int i = 0; i < 9; ++i) w[i] = gWN[i];
                const float q0 = w[0] * e0 + w[1] * e1 + w[2] * e2;
                const float q1 = w[3] * e0 + w[4] * e1 + w[5] * e2;
                const float q2 = w[6] * e0 + w[7] * e1 + w[8] * e2;
                st4(rec, S_DQ, 0.0f, 0.0f, q0, q1);
                st4(rec, S_QR, q2, 0.0f, 0.0f, 0.0f);
                st4(rec, S_RQ, 0.0f, 0.0f, w[0], w[1]);
                st4(rec, S_QQ, w[2], w[4], w[5], w[8]);
                // cost-to-go at the terminal node
                st4(rec, S_V0, w[0], w[1], w[2], w[4]);
                st4(rec, S_V1, w[5], w[8], q0, q1);
                rec[S_DX * 4 + 3] = q2;
            }
        }
        infeasible = group_or<L>(infeasible);
        __syncthreads();
        if (STAMP && sqp == 0) t_stamp[1] = __builtin_amdgcn_s_memtime();

        const float4 xfirst = lds4(row, S_X);
        const float Dx0 = x00 - xfirst.x, Dx1 = x01 - xfirst.y, Dx2 = x02 - xfirst.z;

        // ---- phase B: working-set iterations.  Backward: sequential Riccati sweep, every lane of the
        //      group runs it on the same LDS addresses (broadcast reads), lane 0 of the group stores.
        //      Forward: the closed-loop stage maps dx+ = M_k dx + c_k are affine, so the rollout is a
        //      prefix scan over the lanes (lane j <-> stage j), log2(L) steps instead of N.
        int pd_fail = 0;
        bool changed = true; // "this problem still needs a sweep"
        int khi = N - 1;     // highest stage whose cost-to-go is stale
        int it = 0;
        n_iter = 0;
        long long t_b = 0, t_f = 0;
        for (;;) {
            long long tb0 = 0;
            if (STAMP) tb0 = __builtin_amdgcn_s_memtime();
            // ---- backward Riccati sweep from the highest stale stage of this wavefront
            const int kk = changed ? khi : -1;
            const int kmax = __builtin_amdgcn_readfirstlane(wave_max(kk)); // wavefront-uniform loop bound
            {
                Value V;
                V.P.m00 = V.P.m01 = V.P.m02 = V.P.m11 = V.P.m12 = V.P.m22 = 0.0f;
                V.p0 = V.p1 = V.p2 = 0.0f;
                int ok = 1;
                auto step = [&](int k, const StageRegs& cur, const float4& cst) {
                    if (k == kk) { // this group joins the sweep here: cost-to-go of node k+1
                        const float* rn = row + (k + 1) * SR;
                        const float4 v0 = lds4(rn, S_V0), v1 = lds4(rn, S_V1);
                        V.P.m00 = v0.x; V.P.m01 = v0.y; V.P.m02 = v0.z; V.P.m11 = v0.w;
                        V.P.m12 = v1.x; V.P.m22 = v1.y; V.p0 = v1.z; V.p1 = v1.w;
                        V.p2 = rn[S_DX * 4 + 3];
                    }
                    if (k <= kk) {
                        StageQP s;
                        s.a = cur.l0.x; s.b = cur.l0.y; s.B00 = cur.l0.z; s.B01 = cur.l0.w;
                        s.B10 = cur.l1.x; s.B11 = cur.l1.y; s.B20 = cur.l1.z; s.d0 = cur.l1.w;
                        s.d1 = cur.dq.x; s.d2 = cur.dq.y; s.q0 = cur.dq.z; s.q1 = cur.dq.w;
                        s.q2 = cur.qr.x; s.r0 = cur.qr.y; s.r1 = cur.qr.z; s.R00 = cur.qr.w;
                        s.R01 = cur.rq.x; s.R11 = cur.rq.y; s.Q.m00 = cur.rq.z; s.Q.m01 = cur.rq.w;
                        s.Q.m02 = cur.qq.x; s.Q.m11 = cur.qq.y; s.Q.m12 = cur.qq.z; s.Q.m22 = cur.qq.w;
                        s.st0 = __float_as_int(cst.x);
                        s.st1 = __float_as_int(cst.y);
                        s.v0 = (s.st0 == ST_UPPER) ? cur.bnd.y : cur.bnd.x;
                        s.v1 = (s.st1 == ST_UPPER) ? cur.bnd.w : cur.bnd.z;
                        Policy pol;
                        ok &= riccati_step(s, V, pol, k > 0) ? 1 : 0;
                        if (writer) {
                            float* rec = row + k * SR;
                            st4(rec, S_POL0, pol.c00, pol.c01, pol.c02, pol.f0);
                            st4(rec, S_POL1, pol.c10, pol.c11, pol.c12, pol.e1);
                            rec[S_MU * 4 + 2] = pol.f1;
                            if (k > 0) {
                                st4(rec, S_V0, V.P.m00, V.P.m01, V.P.m02, V.P.m11);
                                st4(rec, S_V1, V.P.m12, V.P.m22, V.p0, V.p1);
                                rec[S_DX * 4 + 3] = V.p2;
                            }
                        }
                    }
                };
                // two register sets, loads of the next record in flight while the current one is used
                StageRegs ra, rb;
                float4 sa = make_float4(0, 0, 0, 0), sb4 = sa;
                int k = kmax;
                if (k >= 0) { load_stage(row + k * SR, ra); sa = lds4(row + k * SR, S_STDU); }
                while (k >= 1) {
                    load_stage(row + (k - 1) * SR, rb); sb4 = lds4(row + (k - 1) * SR, S_STDU);
                    step(k, ra, sa);
                    if (k >= 2) { load_stage(row + (k - 2) * SR, ra); sa = lds4(row + (k - 2) * SR, S_STDU); }
                    step(k - 1, rb, sb4);
                    k -= 2;
                }
                if (k == 0) step(0, ra, sa);
                pd_fail |= (ok == 0);
            }
            __syncthreads();
            long long tf0 = 0;
            if (STAMP) { tf0 = __builtin_amdgcn_s_memtime(); t_b += tf0 - tb0; }

            // ---- forward sweep as a prefix scan, stages in chunks of L (lane j <-> stage base + j)
            const bool first = (it == 0);
            const bool active = changed;
            int new_khi = -1;
            if (__any(active)) {
                float cx0 = Dx0, cx1 = Dx1, cx2 = Dx2; // state step entering the chunk
                float cs0 = Dx0, cs1 = Dx1, cs2 = Dx2; // free response entering the chunk (first sweep)
                for (int base = 0; base < N; base += L) {
                    const int k = base + j;
                    const bool in = k < N;
                    float* rec = row + (in ? k : 0) * SR;
                    const float4 l0 = lds4(rec, S_LIN0), l1 = lds4(rec, S_LIN1), dq = lds4(rec, S_DQ),
                                 bnd = lds4(rec, S_BND), sd = lds4(rec, S_STDU), p0 = lds4(rec, S_POL0),
                                 p1 = lds4(rec, S_POL1), mu = lds4(rec, S_MU);
                    const int st0 = __float_as_int(sd.x), st1 = __float_as_int(sd.y);
                    const bool f0 = (st0 == ST_FREE), f1 = (st1 == ST_FREE);
                    const float b0 = (st0 == ST_UPPER) ? bnd.y : bnd.x, b1 = (st1 == ST_UPPER) ? bnd.w : bnd.z;
                    // du0 = g0.dx + h0 ; du1 = g1.dx + h1  under the current working set
                    const float g00 = f0 ? p0.x : 0.0f, g01 = f0 ? p0.y : 0.0f, g02 = f0 ? p0.z : 0.0f;
                    const float h0 = f0 ? p0.w : b0;
                    const float g10 = f1 ? p1.x + p1.w * g00 : 0.0f, g11 = f1 ? p1.y + p1.w * g01 : 0.0f,
                                g12 = f1 ? p1.z + p1.w * g02 : 0.0f;
                    const float h1 = f1 ? p1.w * h0 + mu.z : b1;
                    const float a = l0.x, b = l0.y, B00 = l0.z, B01 = l0.w, B10 = l1.x, B11 = l1.y, B20 = l1.z;
                    const float d0 = l1.w, d1 = dq.x, d2 = dq.y;
                    // closed-loop stage map  dx+ = M dx + c  (identity for padding lanes)
                    float m00 = 1.0f, m01 = 0.0f, m02 = 0.0f, m10 = 0.0f, m11 = 1.0f, m12 = 0.0f, m20 = 0.0f,
                          m21 = 0.0f, m22 = 1.0f, c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
                    // free-response map: sbar+ = A sbar + d
                    float sa_ = 0.0f, sb_ = 0.0f, sc0 = 0.0f, sc1 = 0.0f, sc2 = 0.0f;
                    if (in) {
                        const float gd0 = g00 - g10, gd1 = g01 - g11, gd2 = g02 - g12; // B2 = B20 * (1, -1)
                        m00 = 1.0f + B00 * g00 + B01 * g10; m01 = B00 * g01 + B01 * g11; m02 = a + B00 * g02 + B01 * g12;
                        m10 = B10 * g00 + B11 * g10; m11 = 1.0f + B10 * g01 + B11 * g11; m12 = b + B10 * g02 + B11 * g12;
                        m20 = B20 * gd0; m21 = B20 * gd1; m22 = 1.0f + B20 * gd2;
                        c0 = B00 * h0 + B01 * h1 + d0;
                        c1 = B10 * h0 + B11 * h1 + d1;
                        c2 = B20 * (h0 - h1) + d2;
                        sa_ = a; sb_ = b; sc0 = d0; sc1 = d1; sc2 = d2;
                    }
                    // inclusive scan of map composition: after it, lane j holds f_k o ... o f_base
#pragma unroll
                    for (int off = 1; off < L; off <<= 1) {
                        const float pm00 = __shfl_up(m00, off, L), pm01 = __shfl_up(m01, off, L), pm02 = __shfl_up(m02, off, L),
                                    pm10 = __shfl_up(m10, off, L), pm11 = __shfl_up(m11, off, L), pm12 = __shfl_up(m12, off, L),
                                    pm20 = __shfl_up(m20, off, L), pm21 = __shfl_up(m21, off, L), pm22 = __shfl_up(m22, off, L),
                                    pc0 = __shfl_up(c0, off, L), pc1 = __shfl_up(c1, off, L), pc2 = __shfl_up(c2, off, L);
                        if (j >= off) {
                            const float n00 = m00 * pm00 + m01 * pm10 + m02 * pm20, n01 = m00 * pm01 + m01 * pm11 + m02 * pm21,
                                        n02 = m00 * pm02 + m01 * pm12 + m02 * pm22;
                            const float n10 = m10 * pm00 + m11 * pm10 + m12 * pm20, n11 = m10 * pm01 + m11 * pm11 + m12 * pm21,
                                        n12 = m10 * pm02 + m11 * pm12 + m12 * pm22;
                            const float n20 = m20 * pm00 + m21 * pm10 + m22 * pm20, n21 = m20 * pm01 + m21 * pm11 + m22 * pm21,
                                        n22 = m20 * pm02 + m21 * pm12 + m22 * pm22;
                            c0 += m00 * pc0 + m01 * pc1 + m02 * pc2;
                            c1 += m10 * pc0 + m11 * pc1 + m12 * pc2;
                            c2 += m20 * pc0 + m21 * pc1 + m22 * pc2;
                            m00 = n00; m01 = n01; m02 = n02; m10 = n10; m11 = n11; m12 = n12; m20 = n20; m21 = n21; m22 = n22;
                        }
                    }
                    // state step leaving stage k (= entering k+1), and the one entering stage k
                    const float o0 = m00 * cx0 + m01 * cx1 + m02 * cx2 + c0;
                    const float o1 = m10 * cx0 + m11 * cx1 + m12 * cx2 + c1;
                    const float o2 = m20 * cx0 + m21 * cx1 + m22 * cx2 + c2;
                    float dx0 = __shfl_up(o0, 1, L), dx1 = __shfl_up(o1, 1, L), dx2 = __shfl_up(o2, 1, L);
                    if (j == 0) { dx0 = cx0; dx1 = cx1; dx2 = cx2; }
                    Policy pol;
                    pol.c00 = p0.x; pol.c01 = p0.y; pol.c02 = p0.z; pol.f0 = p0.w;
                    pol.c10 = p1.x; pol.c11 = p1.y; pol.c12 = p1.z; pol.e1 = p1.w; pol.f1 = mu.z;
                    StageStep o;
                    forward_step(pol, st0, st1, dx0, dx1, dx2, bnd.x, bnd.y, bnd.z, bnd.w, o);
                    if (in && ((o.nst0 != st0) || (o.nst1 != st1))) new_khi = max(new_khi, k);
                    if (in && active) {
                        *reinterpret_cast<float2*>(rec + S_MU * 4) = make_float2(o.mu0, o.mu1);
                        rec[S_DX * 4] = dx0; rec[S_DX * 4 + 1] = dx1; rec[S_DX * 4 + 2] = dx2;
                        st4(rec, S_STDU, __int_as_float(o.nst0), __int_as_float(o.nst1), o.du0, o.du1);
                    }
                    if (first) { // free response (du = 0): A is a shear, so two chained prefix sums do it
                        // sbar2 entering stage k = cs2 + sum_{i<k} d2_i
                        float e2 = sc2;
#pragma unroll
                        for (int off = 1; off < L; off <<= 1) { const float t = __shfl_up(e2, off, L); if (j >= off) e2 += t; }
                        float in2 = __shfl_up(e2, 1, L);
                        in2 = (j == 0) ? cs2 : cs2 + in2;
                        float e0 = sa_ * in2 + sc0, e1 = sb_ * in2 + sc1;
#pragma unroll
                        for (int off = 1; off < L; off <<= 1) {
                            const float t0 = __shfl_up(e0, off, L), t1 = __shfl_up(e1, off, L);
                            if (j >= off) { e0 += t0; e1 += t1; }
                        }
                        float in0 = __shfl_up(e0, 1, L), in1 = __shfl_up(e1, 1, L);
                        in0 = (j == 0) ? cs0 : cs0 + in0;
                        in1 = (j == 0) ? cs1 : cs1 + in1;
                        if (in && active) { rec[S_SB * 4] = in0; rec[S_SB * 4 + 1] = in1; rec[S_SB * 4 + 2] = in2; }
                        // carry to the next chunk: values leaving the last lane
                        const float l0_ = __shfl(e0, L - 1, L), l1_ = __shfl(e1, L - 1, L), l2_ = __shfl(e2, L - 1, L);
                        cs0 += l0_; cs1 += l1_; cs2 += l2_;
                    }
                    cx0 = __shfl(o0, L - 1, L); cx1 = __shfl(o1, L - 1, L); cx2 = __shfl(o2, L - 1, L);
                }
                if (writer && active) { // node N
                    float* rec = row + N * SR;
                    rec[S_DX * 4] = cx0; rec[S_DX * 4 + 1] = cx1; rec[S_DX * 4 + 2] = cx2;
                    if (first) { rec[S_SB * 4] = cs0; rec[S_SB * 4 + 1] = cs1; rec[S_SB * 4 + 2] = cs2; }
                }
                // highest changed stage of the group
#pragma unroll
                for (int off = L / 2; off > 0; off >>= 1) new_khi = max(new_khi, __shfl_xor(new_khi, off, L));
            }
            ++it;
            if (active) {
                changed = (new_khi >= 0);
                khi = new_khi;
                if (changed) n_iter = it;
            }
            if (STAMP) t_f += __builtin_amdgcn_s_memtime() - tf0;
            // wavefront-uniform continuation: any problem of this wavefront still changing?
            const int more = __syncthreads_or((changed && it < p.max_as_iter) ? 1 : 0);
            if (!more) break;
        }
        n_iter = (n_iter == 0) ? 1 : (changed ? n_iter : n_iter + 1); // + the confirming sweep
        status = infeasible ? RET_INIT_FAILED_INFEASIBILITY
                            : (pd_fail ? RET_INIT_FAILED_CHOLESKY : (changed ? RET_MAX_NWSR_REACHED : RET_OK));
        if (STAMP && sqp == 0) { t_stamp[2] = t_b; t_stamp[3] = t_f; t_stamp[4] = __builtin_amdgcn_s_memtime(); }

        // ---- phase C (stage-parallel): KKT value (acado_getKKT), expand (acado_expand), carry the dual
        float gd = 0.0f, comp = 0.0f;
        for (int k = j; k <= N; k += L) {
            float* rec = row + k * SR;
            const float4 dxp = lds4(rec, S_DX), sb = lds4(rec, S_SB), xk = lds4(rec, S_X);
            if (k > 0) { // (Q_k sbar_k + q_k)' (dx_k - sbar_k)
                const float4 dq = lds4(rec, S_DQ), qr = lds4(rec, S_QR), rq = lds4(rec, S_RQ), qq = lds4(rec, S_QQ);
                const float t0 = dxp.x - sb.x, t1 = dxp.y - sb.y, t2 = dxp.z - sb.z;
                gd += (rq.z * sb.x + rq.w * sb.y + qq.x * sb.z + dq.z) * t0 +
                      (rq.w * sb.x + qq.y * sb.y + qq.z * sb.z + dq.w) * t1 +
                      (qq.x * sb.x + qq.z * sb.y + qq.w * sb.z + qr.x) * t2;
            }
            st4(rec, S_X, xk.x + dxp.x, xk.y + dxp.y, xk.z + dxp.z, 0.0f);
            if (k < N) {
                const float4 qr = lds4(rec, S_QR), bnd = lds4(rec, S_BND), sd = lds4(rec, S_STDU), mu = lds4(rec, S_MU),
                             uy = lds4(rec, S_UY);
                gd += qr.y * sd.z + qr.z * sd.w;
                comp += (mu.x > 1e-12f) ? fabsf(bnd.x * mu.x) : ((mu.x < -1e-12f) ? fabsf(bnd.y * mu.x) : 0.0f);
                comp += (mu.y > 1e-12f) ? fabsf(bnd.z * mu.y) : ((mu.y < -1e-12f) ? fabsf(bnd.w * mu.y) : 0.0f);
                // a free control may sit up to TOL_PRIMAL outside its box: keep the iterate feasible
                const float du0 = (bnd.x <= bnd.y) ? clampf(sd.z, bnd.x, bnd.y) : sd.z;
                const float du1 = (bnd.z <= bnd.w) ? clampf(sd.w, bnd.z, bnd.w) : sd.w;
                st4(rec, S_UY, uy.x + du0, uy.y + du1, mu.x, mu.y);
            }
        }
        kkt = fabsf(group_sum<L>(gd)) + group_sum<L>(comp);
        __syncthreads();
    }
    if (STAMP) t_stamp[5] = __builtin_amdgcn_s_memtime();

    // ---- objective at the returned iterate (acado_getObjective) + write-back
    float part = 0.0f;
    for (int k = j; k <= N; k += L) {
        const float* rec = row + k * SR;
        const float4 xk = lds4(rec, S_X);
        if (k < N) {
            const float4 uy = lds4(rec, S_UY);
            const float* yk = gy + k * 5;
            const float* Wk = gW + k * 25;
            float e[5] = {xk.x - yk[0], xk.y - yk[1], xk.z - yk[2], uy.x - yk[3], uy.y - yk[4]};
            float acc = 0.0f;
#pragma unroll
            for (int c = 0; c < 5; ++c) {
                const float t = e[0] * Wk[c] + e[1] * Wk[5 + c] + e[2] * Wk[10 + c] + e[3] * Wk[15 + c] +
                                e[4] * Wk[20 + c];
                acc += e[c] * t;
            }
            part += acc;
            if (valid) {
                float* ou = p.b.u + (size_t)prob * N * 2;
                float* od = p.b.dual + (size_t)prob * N * 2;
                ou[k * 2] = uy.x; ou[k * 2 + 1] = uy.y;
                od[k * 2] = uy.z; od[k * 2 + 1] = uy.w;
            }
        } else { // the reference uses only the diagonal of WN here (acado_solver.c:1442-1444)
            const float e0 = xk.x - gyN[0], e1 = xk.y - gyN[1], e2 = xk.z - gyN[2];
            part += e0 * e0 * gWN[0] + e1 * e1 * gWN[4] + e2 * e2 * gWN[8];
        }
        if (valid) {
            float* ox = p.b.x + (size_t)prob * (N + 1) * 3;
            ox[k * 3] = xk.x; ox[k * 3 + 1] = xk.y; ox[k * 3 + 2] = xk.z;
        }
    }
    const float obj = 0.5f * group_sum<L>(part);
    if (valid && writer) {
        p.b.status[prob] = status;
        p.b.n_iter[prob] = n_iter;
        p.b.kkt[prob] = kkt;
        p.b.obj[prob] = obj;
    }
    if (STAMP && threadIdx.x == 0 && p.stamps) {
        long long* o = p.stamps + (size_t)blockIdx.x * 8;
        const long long t_end = __builtin_amdgcn_s_memtime();
        o[0] = t_stamp[1] - t_stamp[0]; // load + phase A
        o[1] = t_stamp[2];              // backward sweeps
        o[2] = t_stamp[3];              // forward sweeps
        o[3] = t_stamp[5] - t_stamp[4]; // phase C
        o[4] = t_end - t_stamp[5];      // objective + write-back
        o[5] = t_end - t_stamp[0];      // total
    }
}

hipError_t launch_rti(const RtiParams& p, const LaunchGeom& g, hipStream_t s)
{
    dim3 grid(g.grid), block(g.threads);
    hipError_t e = hipSuccess;
    const bool stamp = p.stamps != nullptr;
    switch (g.L) {
#define CASE(LL)                                                                                              \
    case LL: {                                                                                                \
        static size_t configured[2] = {0, 0}; /* raise the dynamic-LDS cap once per size, not per launch */   \
        const void* fn = stamp ? (const void*)rti_kernel<LL, true> : (const void*)rti_kernel<LL, false>;      \
        if (g.lds_bytes > configured[stamp]) {                                                                \
            e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_bytes);        \
            if (e != hipSuccess) return e;                                                                    \
            configured[stamp] = g.lds_bytes;                                                                  \
        }                                                                                                     \
        if (stamp)                                                                                            \
            hipLaunchKernelGGL((rti_kernel<LL, true>), grid, block, g.lds_bytes, s, p);                       \
        else                                                                                                  \
            hipLaunchKernelGGL((rti_kernel<LL, false>), grid, block, g.lds_bytes, s, p);                      \
        break;                                                                                                \
    }
        CASE(4)
        CASE(8)
        CASE(16)
        CASE(32)
        CASE(64)
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
        g[0] = lin.B00; g[1] = lin.B01; g[2] = lin.B10; g[3] = lin.B11; g[4] = lin.B20; g[5] = -lin.B20;
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
