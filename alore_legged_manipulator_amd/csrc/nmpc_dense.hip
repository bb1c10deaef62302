// nmpc_dense.hip -- the condensed (dense) form of the NMPC QP on the GPU, for callers of the reference's dense interface.
//
// The product path never condenses (nmpc_block_kernel.hip solves the stage-wise QP directly).  The reference, however,
// exposes the condensed QP: acado_preparationStep / acado_feedbackStep leave H (2N x 2N), g, lb, ub in acadoWorkspace
// (CG/acado_solver.c:327-363 condensePrep, :365-891 condenseFdb) and acado_solve() (CG/acado_qpoases_interface.cpp:39-60)
// runs qpOASES' QProblemB on exactly those arrays.  Two kernels serve that interface (include/alore_nmpc.h:
// alore_nmpc_condense, alore_nmpc_dense_qp; libalore_acado_compat.so: the workspace members and acado_solve()):
//   * condense_kernel -- one workgroup per problem.  With E(k, j) = d dx_{k+1} / d du_j (E(j, j) = Gu_j, E(k, j) = Gx_k E(k-1, j))
//     and the free response sbar_{k+1} = Gx_k sbar_k + d_k, sbar_0 = x0 - x_0:
//         H(i, j) = [i == j] R_i + sum_{k >= max(i, j)} E(k, i)' Q_{k+1} E(k, j)          (Q_N = QN)
//         g(i)    = r_i + sum_{k >= i} E(k, i)' (Q_{k+1} sbar_{k+1} + q_{k+1})
//     in the reference's layout (row-major, variable 2 i + c = input c of stage i).  All E blocks live in LDS.
//   * dense_qp_kernel -- one workgroup per problem: min 1/2 x'Hx + g'x, lb <= x <= ub for a dense symmetric positive definite H
//     of order n <= 128.  The working-set iteration of the stage-wise solver on the dense form: for a working set (every
//     variable free, at its lower or at its upper bound) the fixed rows / columns of H become an identity, one Cholesky
//     factorisation in LDS solves for the free variables, the multipliers of the fixed ones are (Hx + g); the set is
//     updated by nmpc_core.h: next_status until it reproduces itself; after AS_SWITCH iterations only the most violated
//     change is applied (no cycling).  The minimiser of a strictly convex QP is unique: the result is what qpOASES returns,
//     up to float32 rounding.
#include "nmpc_kernels.h"

#include "nmpc_core.h"

namespace nmpc {
namespace {

constexpr int DT = 256; // threads per workgroup of both kernels

__global__ __launch_bounds__(DT) void condense_kernel(alore_nmpc_batch b, const float* lin_x, const float* lin_u, int B, int N, IrkConst K,
                                                      unsigned shared, float* Hg, float* gg, float* lbg, float* ubg)
{
    extern __shared__ float sm[];
    const int prob = blockIdx.x, tid = threadIdx.x;
    const int nx = 3 * (N + 1), nu = 2 * N, n = 2 * N;
    // LDS map: per stage 24 floats (a b B00 B01 B10 B11 B20 d0 d1 d2 | Q00 Q01 Q02 Q11 Q12 Q22 q0 q1 q2 | R00 R01 R11 r0 r1),
    // terminal 9, sbar 3 (N + 1), E 6 N (N + 1) / 2, t 3 N (t_k = Q_{k+1} sbar_{k+1} + q_{k+1})
    float* st = sm;
    float* term = st + 24 * N;
    float* sbar = term + 12;
    float* tt = sbar + 3 * (N + 1);
    float* E = tt + 3 * N;
    const float* x = b.x + (size_t)prob * nx;
    const float* u = b.u + (size_t)prob * nu;
    const float* lx = lin_x ? lin_x + (size_t)prob * nx : x;
    const float* lu = lin_u ? lin_u + (size_t)prob * nu : u;
    const float* od = b.od + ((shared & ALORE_NMPC_SHARED_OD) ? 0 : (size_t)prob * nx);
    const float* y = b.y + (size_t)prob * 5 * N;
    const float* W = b.W + ((shared & ALORE_NMPC_SHARED_W) ? 0 : (size_t)prob * 25 * N);
    const float* WN = b.WN + ((shared & ALORE_NMPC_SHARED_W) ? 0 : (size_t)prob * 9);
    const float* lbv = b.lbValues + ((shared & ALORE_NMPC_SHARED_BOUNDS) ? 0 : (size_t)prob * nu);
    const float* ubv = b.ubValues + ((shared & ALORE_NMPC_SHARED_BOUNDS) ? 0 : (size_t)prob * nu);
    for (int k = tid; k < N; k += DT) {
        StageLin lin;
        ddr_linearize(K, lx[3 * k], lx[3 * k + 1], lx[3 * k + 2], lu[2 * k], lu[2 * k + 1], od[3 * k], od[3 * k + 1], od[3 * k + 2], lin);
        float* s = st + 24 * k;
        s[0] = lin.a; s[1] = lin.b; s[2] = lin.B00; s[3] = lin.B01; s[4] = lin.B10; s[5] = lin.B11; s[6] = lin.B20;
        s[7] = lin.phi0 - lx[3 * k + 3]; s[8] = lin.phi1 - lx[3 * k + 4]; s[9] = lin.phi2 - lx[3 * k + 5];
        const float* w = W + 25 * k;
        const float e0 = lx[3 * k] - y[5 * k], e1 = lx[3 * k + 1] - y[5 * k + 1], e2 = lx[3 * k + 2] - y[5 * k + 2],
                    e3 = lu[2 * k] - y[5 * k + 3], e4 = lu[2 * k + 1] - y[5 * k + 4];
        s[10] = w[0]; s[11] = w[1]; s[12] = w[2]; s[13] = w[6]; s[14] = w[7]; s[15] = w[12];
        s[16] = w[0] * e0 + w[1] * e1 + w[2] * e2 + w[3] * e3 + w[4] * e4;
        s[17] = w[5] * e0 + w[6] * e1 + w[7] * e2 + w[8] * e3 + w[9] * e4;
        s[18] = w[10] * e0 + w[11] * e1 + w[12] * e2 + w[13] * e3 + w[14] * e4;
        s[19] = w[18]; s[20] = w[19]; s[21] = w[24];
        s[22] = w[15] * e0 + w[16] * e1 + w[17] * e2 + w[18] * e3 + w[19] * e4;
        s[23] = w[20] * e0 + w[21] * e1 + w[22] * e2 + w[23] * e3 + w[24] * e4;
        lbg[(size_t)prob * n + 2 * k] = lbv[2 * k] - u[2 * k]; lbg[(size_t)prob * n + 2 * k + 1] = lbv[2 * k + 1] - u[2 * k + 1];
        ubg[(size_t)prob * n + 2 * k] = ubv[2 * k] - u[2 * k]; ubg[(size_t)prob * n + 2 * k + 1] = ubv[2 * k + 1] - u[2 * k + 1];
    }
    if (tid == 0) {
        const float e0 = lx[3 * N] - b.yN[(size_t)prob * 3], e1 = lx[3 * N + 1] - b.yN[(size_t)prob * 3 + 1], e2 = lx[3 * N + 2] - b.yN[(size_t)prob * 3 + 2];
        term[0] = WN[0]; term[1] = WN[1]; term[2] = WN[2]; term[3] = WN[4]; term[4] = WN[5]; term[5] = WN[8];
        term[6] = WN[0] * e0 + WN[1] * e1 + WN[2] * e2;
        term[7] = WN[3] * e0 + WN[4] * e1 + WN[5] * e2;
        term[8] = WN[6] * e0 + WN[7] * e1 + WN[8] * e2;
    }
    __syncthreads();
    // node k + 1 data: Q (6) and q (3) of stage k + 1, or the terminal ones
    auto Qn = [&](int k1) -> const float* { return k1 < N ? st + 24 * k1 + 10 : term; };
    if (tid == 0) { // free response and t_k = Q_{k+1} sbar_{k+1} + q_{k+1}: a serial walk of N steps
        float s0 = b.x0[(size_t)prob * 3] - x[0], s1 = b.x0[(size_t)prob * 3 + 1] - x[1], s2 = b.x0[(size_t)prob * 3 + 2] - x[2];
        sbar[0] = s0; sbar[1] = s1; sbar[2] = s2;
        for (int k = 0; k < N; ++k) {
            const float* s = st + 24 * k;
            const float n0 = s0 + s[0] * s2 + s[7], n1 = s1 + s[1] * s2 + s[8], n2 = s2 + s[9];
            s0 = n0; s1 = n1; s2 = n2;
            sbar[3 * k + 3] = s0; sbar[3 * k + 4] = s1; sbar[3 * k + 5] = s2;
            const float* q = Qn(k + 1);
            tt[3 * k] = q[0] * s0 + q[1] * s1 + q[2] * s2 + q[6];
            tt[3 * k + 1] = q[1] * s0 + q[3] * s1 + q[4] * s2 + q[7];
            tt[3 * k + 2] = q[2] * s0 + q[4] * s1 + q[5] * s2 + q[8];
        }
    }
    // E(k, j), k >= j, at offset 6 (k (k + 1) / 2 + j): thread j walks k = j .. N - 1 (Gx = I + [0 0 a; 0 0 b; 0 0 0])
    for (int j = tid; j < N; j += DT) {
        const float* s = st + 24 * j;
        float e00 = s[2], e01 = s[3], e10 = s[4], e11 = s[5], e20 = s[6], e21 = -s[6];
        for (int k = j; k < N; ++k) {
            if (k > j) {
                const float* sk = st + 24 * k;
                e00 += sk[0] * e20; e01 += sk[0] * e21; e10 += sk[1] * e20; e11 += sk[1] * e21;
            }
            float* e = E + 6 * (k * (k + 1) / 2 + j);
            e[0] = e00; e[1] = e01; e[2] = e10; e[3] = e11; e[4] = e20; e[5] = e21;
        }
    }
    __syncthreads();
    // H blocks (i <= j), mirrored; g
    float* H = Hg + (size_t)prob * n * n;
    const int npairs = N * (N + 1) / 2;
    for (int pi = tid; pi < npairs; pi += DT) {
        // pi -> (i, j) with i <= j: j = row of the triangular index
        int j = (int)((sqrtf(8.0f * pi + 1.0f) - 1.0f) * 0.5f);
        while ((j + 1) * (j + 2) / 2 <= pi) ++j;
        while (j * (j + 1) / 2 > pi) --j;
        const int i = pi - j * (j + 1) / 2;
        float h00 = 0.f, h01 = 0.f, h10 = 0.f, h11 = 0.f;
        for (int k = j; k < N; ++k) {
            const float* q = Qn(k + 1);
            const float* ei = E + 6 * (k * (k + 1) / 2 + i);
            const float* ej = E + 6 * (k * (k + 1) / 2 + j);
            // m = Q E(k, j)  (3 x 2)
            const float m00 = q[0] * ej[0] + q[1] * ej[2] + q[2] * ej[4], m01 = q[0] * ej[1] + q[1] * ej[3] + q[2] * ej[5];
            const float m10 = q[1] * ej[0] + q[3] * ej[2] + q[4] * ej[4], m11 = q[1] * ej[1] + q[3] * ej[3] + q[4] * ej[5];
            const float m20 = q[2] * ej[0] + q[4] * ej[2] + q[5] * ej[4], m21 = q[2] * ej[1] + q[4] * ej[3] + q[5] * ej[5];
            h00 += ei[0] * m00 + ei[2] * m10 + ei[4] * m20; h01 += ei[0] * m01 + ei[2] * m11 + ei[4] * m21;
            h10 += ei[1] * m00 + ei[3] * m10 + ei[5] * m20; h11 += ei[1] * m01 + ei[3] * m11 + ei[5] * m21;
        }
        if (i == j) {
            const float* s = st + 24 * i;
            h00 += s[19]; h01 += s[20]; h10 += s[20]; h11 += s[21];
        }
        H[(size_t)(2 * i) * n + 2 * j] = h00; H[(size_t)(2 * i) * n + 2 * j + 1] = h01;
        H[(size_t)(2 * i + 1) * n + 2 * j] = h10; H[(size_t)(2 * i + 1) * n + 2 * j + 1] = h11;
        if (i != j) {
            H[(size_t)(2 * j) * n + 2 * i] = h00; H[(size_t)(2 * j + 1) * n + 2 * i] = h01;
            H[(size_t)(2 * j) * n + 2 * i + 1] = h10; H[(size_t)(2 * j + 1) * n + 2 * i + 1] = h11;
        }
    }
    for (int i = tid; i < N; i += DT) {
        const float* s = st + 24 * i;
        float g0 = s[22], g1 = s[23];
        for (int k = i; k < N; ++k) {
            const float* e = E + 6 * (k * (k + 1) / 2 + i);
            const float* t = tt + 3 * k;
            g0 += e[0] * t[0] + e[2] * t[1] + e[4] * t[2];
            g1 += e[1] * t[0] + e[3] * t[1] + e[5] * t[2];
        }
        gg[(size_t)prob * n + 2 * i] = g0; gg[(size_t)prob * n + 2 * i + 1] = g1;
    }
}

// dense box QP, one workgroup per problem.  LDS: Kc (n x n, the working-set system, factorised in place), rhs / solution, flags.
__global__ __launch_bounds__(DT) void dense_qp_kernel(int n, const float* Hg, const float* gg, const float* lbg, const float* ubg, float* xg,
                                                      float* yg, int* statusg, int* niterg, int max_iter)
{
    extern __shared__ float sm[];
    const int prob = blockIdx.x, tid = threadIdx.x;
    const int ld = n + 1; // odd stride: conflict-free column walks
    float* Kc = sm;
    float* rhs = Kc + (size_t)n * ld;
    float* xs = rhs + n;
    float* val = xs + n;
    int* st = reinterpret_cast<int*>(val + n);
    int* flags = st + n; // [0] changed count, [1] pivot failure, [2] worst index (single-change mode), as ints
    const float* H = Hg + (size_t)prob * n * n;
    const float* g = gg + (size_t)prob * n;
    const float* lb = lbg + (size_t)prob * n;
    const float* ub = ubg + (size_t)prob * n;
    int infeasible = 0;
    for (int i = tid; i < n; i += DT) {
        st[i] = status_from_dual(yg[(size_t)prob * n + i], lb[i], ub[i]); // qpOASES' guess from the previous dual (QProblemB.cpp:1010-1036)
        infeasible |= (lb[i] > ub[i] + 1e-6f) ? 1 : 0;
    }
    if (tid == 0) { flags[0] = 0; flags[1] = 0; flags[3] = 0; }
    __syncthreads();
    if (infeasible) flags[3] = 1;
    __syncthreads();
    if (flags[3]) {
        if (tid == 0) { statusg[prob] = RET_INIT_FAILED_INFEASIBILITY; niterg[prob] = 0; }
        return;
    }
    int it = 0, status = RET_MAX_NWSR_REACHED;
    for (; it < max_iter; ++it) {
        // working-set system: fixed rows / columns become an identity; rhs_i = -g_i - sum_{j fixed} H_ij v_j (free), v_i (fixed)
        for (int e = tid; e < n * n; e += DT) {
            const int r = e / n, c = e % n;
            const bool fr = st[r] != ST_FREE, fc = st[c] != ST_FREE;
            Kc[r * ld + c] = (fr || fc) ? ((r == c) ? 1.0f : 0.0f) : H[e];
        }
        for (int i = tid; i < n; i += DT) {
            if (st[i] != ST_FREE) { rhs[i] = (st[i] == ST_UPPER) ? ub[i] : lb[i]; continue; }
            float acc = -g[i];
            for (int j = 0; j < n; ++j)
                if (st[j] != ST_FREE) acc -= H[(size_t)i * n + j] * ((st[j] == ST_UPPER) ? ub[j] : lb[j]);
            rhs[i] = acc;
        }
        __syncthreads();
        // Cholesky Kc = L L' in place (lower), right-looking
        for (int k = 0; k < n; ++k) {
            if (tid == 0) {
                const float d = Kc[k * ld + k];
                if (!(d > 0.0f)) flags[1] = 1;
                Kc[k * ld + k] = sqrtf(fmaxf(d, 1e-30f));
            }
            __syncthreads();
            const float dk = Kc[k * ld + k];
            for (int r = k + 1 + tid; r < n; r += DT) Kc[r * ld + k] /= dk;
            __syncthreads();
            const int m = n - k - 1;
            for (int e = tid; e < m * m; e += DT) {
                const int r = k + 1 + e / m, c = k + 1 + e % m;
                if (c <= r) Kc[r * ld + c] -= Kc[r * ld + k] * Kc[c * ld + k];
            }
            __syncthreads();
        }
        // L z = rhs, L' x = z (column-oriented: thread-parallel updates of the remaining right-hand side)
        for (int k = 0; k < n; ++k) {
            if (tid == 0) rhs[k] /= Kc[k * ld + k];
            __syncthreads();
            const float zk = rhs[k];
            for (int r = k + 1 + tid; r < n; r += DT) rhs[r] -= Kc[r * ld + k] * zk;
            __syncthreads();
        }
        for (int k = n - 1; k >= 0; --k) {
            if (tid == 0) rhs[k] /= Kc[k * ld + k];
            __syncthreads();
            const float xk = rhs[k];
            for (int r = tid; r < k; r += DT) rhs[r] -= Kc[k * ld + r] * xk;
            __syncthreads();
        }
        for (int i = tid; i < n; i += DT) xs[i] = rhs[i];
        __syncthreads();
        // value that decides the next status: the variable itself if free, its multiplier (H x + g)_i if fixed
        for (int i = tid; i < n; i += DT) {
            float v = xs[i];
            if (st[i] != ST_FREE) {
                float acc = g[i];
                for (int j = 0; j < n; ++j) acc += H[(size_t)i * n + j] * xs[j];
                v = acc;
            }
            val[i] = v;
        }
        if (tid == 0) { flags[0] = 0; flags[2] = -1; }
        __syncthreads();
        if (it < AS_SWITCH) {
            for (int i = tid; i < n; i += DT) {
                const int ns = next_status(st[i], val[i], lb[i], ub[i]);
                if (ns != st[i]) { st[i] = ns; atomicAdd(&flags[0], 1); }
            }
        } else if (tid == 0) { // single change: the most violated variable (first index on ties): finite, no cycling
            float worst = 0.0f;
            int wi = -1, wns = 0;
            for (int i = 0; i < n; ++i) {
                const int ns = next_status(st[i], val[i], lb[i], ub[i]);
                if (ns == st[i]) continue;
                const float viol = (st[i] == ST_FREE) ? fmaxf(lb[i] - val[i], val[i] - ub[i]) : fabsf(val[i]);
                if (viol > worst) { worst = viol; wi = i; wns = ns; }
            }
            if (wi >= 0) { st[wi] = wns; flags[0] = 1; }
        }
        __syncthreads();
        if (flags[0] == 0) { status = RET_OK; ++it; break; }
        __syncthreads();
    }
    if (flags[1]) status = RET_INIT_FAILED_CHOLESKY;
    for (int i = tid; i < n; i += DT) {
        xg[(size_t)prob * n + i] = xs[i];
        yg[(size_t)prob * n + i] = (st[i] == ST_FREE) ? 0.0f : val[i]; // > 0 at a lower bound, < 0 at an upper one (qpOASES' sign)
    }
    if (tid == 0) { statusg[prob] = status; niterg[prob] = it; }
}

} // namespace

size_t condense_lds_bytes(int N) { return sizeof(float) * ((size_t)24 * N + 12 + 3 * (N + 1) + 3 * N + (size_t)6 * N * (N + 1) / 2); }
size_t dense_qp_lds_bytes(int n) { return sizeof(float) * ((size_t)n * (n + 1) + 3 * n) + sizeof(int) * ((size_t)n + 8); }

hipError_t launch_condense(const alore_nmpc_batch& b, const float* lin_x, const float* lin_u, int B, int N, float dt, unsigned shared, float* H,
                           float* g, float* lb, float* ub, hipStream_t s)
{
    const size_t lds = condense_lds_bytes(N);
    int dev = 0;
    if (const hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    static size_t raised[16] = {0}; // per device: the attribute belongs to the function on ONE device
    if (lds > 48 * 1024 && lds > raised[dev & 15]) {
        const hipError_t e = hipFuncSetAttribute((const void*)condense_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        raised[dev & 15] = lds;
    }
    hipLaunchKernelGGL(condense_kernel, dim3(B), dim3(DT), lds, s, b, lin_x, lin_u, B, N, make_irk(dt), shared, H, g, lb, ub);
    return hipGetLastError();
}

hipError_t launch_dense_qp(int B, int n, const float* H, const float* g, const float* lb, const float* ub, float* x, float* y, int* status,
                           int* n_iter, int max_iter, hipStream_t s)
{
    const size_t lds = dense_qp_lds_bytes(n);
    int dev = 0;
    if (const hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    static size_t raised[16] = {0}; // per device: the attribute belongs to the function on ONE device
    if (lds > 48 * 1024 && lds > raised[dev & 15]) {
        const hipError_t e = hipFuncSetAttribute((const void*)dense_qp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        raised[dev & 15] = lds;
    }
    hipLaunchKernelGGL(dense_qp_kernel, dim3(B), dim3(DT), lds, s, n, H, g, lb, ub, x, y, status, n_iter, max_iter);
    return hipGetLastError();
}

} // namespace nmpc
