// nmpc_scan.h -- the backward Riccati recursion as an ASSOCIATIVE SCAN over stage elements (float32).
//
// The sweep of nmpc_core.h: riccati_step is N sequential stage steps; on the mappings that spread a problem over 16 or 32 lanes (one
// batch at a time, one robot, the reference's N = 50) that chain is a third to a half of a launch.  Sarkka & Garcia-Fernandez ("Temporal
// parallelization of dynamic programming and linear quadratic control", IEEE TAC 2023) write the recursion as the product of
// per-stage elements under an associative combine rule, so that the cost-to-go entering every lane's block of stages comes out of
// log2(L) combine levels over the lanes.  Element of the stage interval [i, j): (A, b, C, eta, J) of the conditional value function
//     V(x_i, x_j) = max_l  1/2 x_i' J x_i + eta' x_i - 1/2 l' C l - l' (x_j - A x_i - b),
// for one stage x+ = A x + B u + d, cost 1/2 x'Qx + q'x + 1/2 u'Ru + r'u (no x-u cross term: the generated solver of the reference
// has none, acado_solver.c: Q1, R1 only), inputs of the working set held at their bounds:
//     A as it is,  b = d + B (held values - Ri r'),  C = B Ri B',  eta = q,  J = Q,     Ri = inverse of R over the free inputs.
// Combine [i, j) then [j, k):  M = (I + C1 J2)^-1,
//     A = A2 M A1,   b = A2 M (b1 - C1 eta2) + b2,   C = A2 (M C1) A2' + C2,
//     eta = A1' (M' eta2 + Z b1) + eta1,   J = A1' Z A1 + J1,     Z = M' J2  (symmetric, like M C1).
// The terminal node is the element (A = 0, b = 0, C = 0, eta = qN, J = QN); (eta, J) of the suffix [k, N] are (p_k, P_k) of the
// cost-to-go.  An interval behind the terminal node is the identity (A = I, rest 0).  Accuracy on the real stage data -- stress and
// bench distributions, working set of the solution, N = 20 and 50: the step du within 3e-5 of the float64 recursion, like the
// sequential float32 recursion (2e-5): tools/scan_riccati_validate.py.
// Symmetric 3 x 3 matrices travel as 00 01 02 11 12 22 (the order of the cost-to-go in the kernels).
#ifndef ALORE_NMPC_SCAN_H
#define ALORE_NMPC_SCAN_H

#include "nmpc_core.h"

namespace nmpc {

struct ScanElem {
    float A[9], b[3], C[6], h[3], J[6];
};

NMPC_HD float scan_sym(const float (&S)[6], int i, int j)
{
    // 00 01 02 11 12 22
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    return S[lo == 0 ? hi : (lo == 1 ? 2 + hi : 5)];
}

// element of one stage; `bad`: a pivot of R over the free inputs is not positive (the caller falls back to the sequential sweep)
NMPC_HD void scan_stage_element(float a, float b, float B00, float B01, float B10, float B11, float B20, float d0, float d1, float d2, float Q00,
                                float Q01, float Q02, float Q11, float Q12, float Q22, float q0, float q1, float q2, float R00, float R01,
                                float R11, float r0, float r1, int st0, int st1, float v0, float v1, ScanElem& e, bool& bad)
{
    const bool f0 = st0 == ST_FREE, f1 = st1 == ST_FREE;
    const float h0 = f0 ? 0.0f : v0, h1 = f1 ? 0.0f : v1;
    const float i1 = f1 ? pivot_rcp(R11) : 0.0f;
    const float t = R01 * i1;
    const float s0 = R00 - R01 * t;
    const float i0 = f0 ? pivot_rcp(s0) : 0.0f;
    bad = (f1 && !(R11 > 0.0f)) || (f0 && !(s0 > 0.0f));
    const float Ri00 = i0, Ri01 = -i0 * t, Ri11 = i1 + t * t * i0;
    const float rr0 = r0 + R01 * h1, rr1 = r1 + R01 * h0;
    const float u0 = h0 - (Ri00 * rr0 + Ri01 * rr1), u1 = h1 - (Ri01 * rr0 + Ri11 * rr1);
    e.b[0] = d0 + B00 * u0 + B01 * u1;
    e.b[1] = d1 + B10 * u0 + B11 * u1;
    e.b[2] = d2 + B20 * (u0 - u1);
    const float BR00 = B00 * Ri00 + B01 * Ri01, BR01 = B00 * Ri01 + B01 * Ri11;
    const float BR10 = B10 * Ri00 + B11 * Ri01, BR11 = B10 * Ri01 + B11 * Ri11;
    const float BR20 = B20 * (Ri00 - Ri01), BR21 = B20 * (Ri01 - Ri11);
    e.C[0] = BR00 * B00 + BR01 * B01;
    e.C[1] = BR00 * B10 + BR01 * B11;
    e.C[2] = (BR00 - BR01) * B20;
    e.C[3] = BR10 * B10 + BR11 * B11;
    e.C[4] = (BR10 - BR11) * B20;
    e.C[5] = (BR20 - BR21) * B20;
    e.A[0] = 1.0f; e.A[1] = 0.0f; e.A[2] = a;
    e.A[3] = 0.0f; e.A[4] = 1.0f; e.A[5] = b;
    e.A[6] = 0.0f; e.A[7] = 0.0f; e.A[8] = 1.0f;
    e.h[0] = q0; e.h[1] = q1; e.h[2] = q2;
    e.J[0] = Q00; e.J[1] = Q01; e.J[2] = Q02; e.J[3] = Q11; e.J[4] = Q12; e.J[5] = Q22;
}

NMPC_HD void scan_identity(ScanElem& e)
{
#pragma unroll
    for (int i = 0; i < 9; ++i) e.A[i] = (i % 4 == 0) ? 1.0f : 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) { e.b[i] = 0.0f; e.h[i] = 0.0f; }
#pragma unroll
    for (int i = 0; i < 6; ++i) { e.C[i] = 0.0f; e.J[i] = 0.0f; }
}

// M = (I + C1 J2)^-1 by cofactors (the matrix is I + a product of two positive semidefinite ones: eigenvalues >= 1)
NMPC_HD void scan_m(const float (&C1)[6], const float (&J2)[6], float (&M)[9])
{
    float X[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            X[3 * i + j] = (i == j ? 1.0f : 0.0f) + scan_sym(C1, i, 0) * scan_sym(J2, 0, j) + scan_sym(C1, i, 1) * scan_sym(J2, 1, j) +
                           scan_sym(C1, i, 2) * scan_sym(J2, 2, j);
    const float c00 = X[4] * X[8] - X[5] * X[7], c01 = X[2] * X[7] - X[1] * X[8], c02 = X[1] * X[5] - X[2] * X[4];
    const float c10 = X[5] * X[6] - X[3] * X[8], c11 = X[0] * X[8] - X[2] * X[6], c12 = X[2] * X[3] - X[0] * X[5];
    const float c20 = X[3] * X[7] - X[4] * X[6], c21 = X[1] * X[6] - X[0] * X[7], c22 = X[0] * X[4] - X[1] * X[3];
    const float r = pivot_rcp(X[0] * c00 + X[1] * c10 + X[2] * c20);
    M[0] = c00 * r; M[1] = c01 * r; M[2] = c02 * r;
    M[3] = c10 * r; M[4] = c11 * r; M[5] = c12 * r;
    M[6] = c20 * r; M[7] = c21 * r; M[8] = c22 * r;
}

// (eta, J) of [i, j) followed by an interval that reaches the terminal node, of which only (J2, h2) matter: one "block" Riccati step
NMPC_HD void scan_apply(const ScanElem& e1, const float (&J2)[6], const float (&h2)[3], float (&J)[6], float (&h)[3])
{
    float M[9];
    scan_m(e1.C, J2, M);
    float Z[6]; // M' J2, upper triangle
    {
        int n = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = i; j < 3; ++j)
                Z[n++] = M[i] * scan_sym(J2, 0, j) + M[3 + i] * scan_sym(J2, 1, j) + M[6 + i] * scan_sym(J2, 2, j);
    }
    float w[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        w[i] = M[i] * h2[0] + M[3 + i] * h2[1] + M[6 + i] * h2[2] + scan_sym(Z, i, 0) * e1.b[0] + scan_sym(Z, i, 1) * e1.b[1] + scan_sym(Z, i, 2) * e1.b[2];
    float Y[9]; // Z A1
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) Y[3 * i + j] = scan_sym(Z, i, 0) * e1.A[j] + scan_sym(Z, i, 1) * e1.A[3 + j] + scan_sym(Z, i, 2) * e1.A[6 + j];
    float Jn[6], hn[3];
    {
        int n = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = i; j < 3; ++j) Jn[n++] = e1.A[i] * Y[j] + e1.A[3 + i] * Y[3 + j] + e1.A[6 + i] * Y[6 + j] + scan_sym(e1.J, i, j);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) hn[i] = e1.A[i] * w[0] + e1.A[3 + i] * w[1] + e1.A[6 + i] * w[2] + e1.h[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) J[i] = Jn[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) h[i] = hn[i];
}

// out = [i, j) then [j, k)   (out may alias e1 or e2)
NMPC_HD void scan_combine(const ScanElem& e1, const ScanElem& e2, ScanElem& out)
{
    float M[9];
    scan_m(e1.C, e2.J, M);
    ScanElem r;
    // eta, J: as scan_apply
    scan_apply(e1, e2.J, e2.h, r.J, r.h); // (the compiler shares M and Z with the lines below: same expressions)
    float Wc[6]; // M C1, upper triangle
    {
        int n = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = i; j < 3; ++j)
                Wc[n++] = M[3 * i] * scan_sym(e1.C, 0, j) + M[3 * i + 1] * scan_sym(e1.C, 1, j) + M[3 * i + 2] * scan_sym(e1.C, 2, j);
    }
    float N1[9]; // M A1
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) N1[3 * i + j] = M[3 * i] * e1.A[j] + M[3 * i + 1] * e1.A[3 + j] + M[3 * i + 2] * e1.A[6 + j];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) r.A[3 * i + j] = e2.A[3 * i] * N1[j] + e2.A[3 * i + 1] * N1[3 + j] + e2.A[3 * i + 2] * N1[6 + j];
    float v[3], u1[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) v[i] = e1.b[i] - (scan_sym(e1.C, i, 0) * e2.h[0] + scan_sym(e1.C, i, 1) * e2.h[1] + scan_sym(e1.C, i, 2) * e2.h[2]);
#pragma unroll
    for (int i = 0; i < 3; ++i) u1[i] = M[3 * i] * v[0] + M[3 * i + 1] * v[1] + M[3 * i + 2] * v[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) r.b[i] = e2.A[3 * i] * u1[0] + e2.A[3 * i + 1] * u1[1] + e2.A[3 * i + 2] * u1[2] + e2.b[i];
    float T1[9]; // A2 Wc
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) T1[3 * i + j] = e2.A[3 * i] * scan_sym(Wc, 0, j) + e2.A[3 * i + 1] * scan_sym(Wc, 1, j) + e2.A[3 * i + 2] * scan_sym(Wc, 2, j);
    {
        int n = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = i; j < 3; ++j) r.C[n++] = T1[3 * i] * e2.A[3 * j] + T1[3 * i + 1] * e2.A[3 * j + 1] + T1[3 * i + 2] * e2.A[3 * j + 2] + scan_sym(e2.C, i, j);
    }
    out = r;
}

} // namespace nmpc
#endif
