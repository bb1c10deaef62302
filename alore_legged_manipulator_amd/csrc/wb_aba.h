// wb_aba.h -- articulated-body algorithm (forward dynamics in O(n)) of the B2 + Z1 tree for one GPU lane.
//
// Featherstone's three passes (Rigid Body Dynamics Algorithms, table 7.1; floating base as in section 9.4), written for
// this tree the way wb_dynamics.h writes the RNEA: spatial vectors are pairs of 3-vectors (angular; linear) in body
// coordinates at the body origin, a joint transform is one plane rotation plus a translation, the five chains are
// walked down (velocities, bias terms, articulated inertias handed back up) and down again (accelerations); per link
// 16 doubles are kept between the passes.  The 6 x 6 articulated inertia is held as three 3 x 3 blocks
// [[A, B], [B', C]] (A, C symmetric).  Gravity: a free-floating tree with prescribed external forces falls as a
// whole, so the algorithm runs without gravity and the base's linear acceleration gets R0' (0, 0, -g) added.
// Conventions as in wb_dynamics.h / oracle/wb_oracle.py (aba): out = d/dt of [omega_base | v_base | joint rates].
#pragma once
#include "wb_dynamics.h"

namespace wb {

struct M3 { // general 3 x 3, row major
    double a[9];
};
WB_FN V3 mul(const M3& m, V3 v) { return {m.a[0] * v.x + m.a[1] * v.y + m.a[2] * v.z, m.a[3] * v.x + m.a[4] * v.y + m.a[5] * v.z, m.a[6] * v.x + m.a[7] * v.y + m.a[8] * v.z}; }
WB_FN V3 mulT(const M3& m, V3 v) { return {m.a[0] * v.x + m.a[3] * v.y + m.a[6] * v.z, m.a[1] * v.x + m.a[4] * v.y + m.a[7] * v.z, m.a[2] * v.x + m.a[5] * v.y + m.a[8] * v.z}; }
WB_FN M3 add(const M3& x, const M3& y) { M3 r; for (int i = 0; i < 9; ++i) r.a[i] = x.a[i] + y.a[i]; return r; }
WB_FN V3 col(const M3& m, int j) { return {m.a[j], m.a[3 + j], m.a[6 + j]}; }
WB_FN V3 row(const M3& m, int i) { return {m.a[3 * i], m.a[3 * i + 1], m.a[3 * i + 2]}; }
WB_FN M3 from_cols(V3 c0, V3 c1, V3 c2) { return {{c0.x, c1.x, c2.x, c0.y, c1.y, c2.y, c0.z, c1.z, c2.z}}; }
WB_FN M3 from_rows(V3 r0, V3 r1, V3 r2) { return {{r0.x, r0.y, r0.z, r1.x, r1.y, r1.z, r2.x, r2.y, r2.z}}; }
WB_FN M3 outer(V3 u, V3 v) { return {{u.x * v.x, u.x * v.y, u.x * v.z, u.y * v.x, u.y * v.y, u.y * v.z, u.z * v.x, u.z * v.y, u.z * v.z}}; }
WB_FN M3 scale(double s, const M3& m) { M3 r; for (int i = 0; i < 9; ++i) r.a[i] = s * m.a[i]; return r; }
// r x M (skew(r) times M, column by column) and M x r (M times skew(r): row i of the result is -(r x row_i))
WB_FN M3 cross_left(V3 r, const M3& m) { return from_cols(cross(r, col(m, 0)), cross(r, col(m, 1)), cross(r, col(m, 2))); }
WB_FN M3 cross_right(const M3& m, V3 r) { return from_rows(cross(row(m, 0), r), cross(row(m, 1), r), cross(row(m, 2), r)); }
WB_FN M3 transpose(const M3& m) { return {{m.a[0], m.a[3], m.a[6], m.a[1], m.a[4], m.a[7], m.a[2], m.a[5], m.a[8]}}; }
// R M R' for the plane rotation R(AX, c, s) (child -> parent coordinates)
template <int AX>
WB_FN M3 rotate(double c, double s, const M3& m)
{
    const M3 t = from_cols(rot<AX>(c, s, col(m, 0)), rot<AX>(c, s, col(m, 1)), rot<AX>(c, s, col(m, 2))); // R M
    return from_rows(rot<AX>(c, s, row(t, 0)), rot<AX>(c, s, row(t, 1)), rot<AX>(c, s, row(t, 2)));       // (R M) R'
}

struct Art { // articulated inertia [[A, B], [B', C]] and bias force (n; f)
    M3 A, B, C;
    V3 n, f;
};

// rigid-body inertia of body I about its origin and its velocity-product force v x* (I v) (minus nothing: external
// forces are subtracted by the caller)
template <int I>
WB_FN void body_inertia(Art& o)
{
    const V3 c = com<I>();
    const double m = b2z1::MASS[I];
    constexpr double xx = b2z1::INERTIA[6 * I], xy = b2z1::INERTIA[6 * I + 1], xz = b2z1::INERTIA[6 * I + 2];
    constexpr double yy = b2z1::INERTIA[6 * I + 3], yz = b2z1::INERTIA[6 * I + 4], zz = b2z1::INERTIA[6 * I + 5];
    // A = Ic + m (|c|^2 1 - c c'),  B = m c x,  C = m 1
    const double cc = c.x * c.x + c.y * c.y + c.z * c.z;
    o.A = {{xx + m * (cc - c.x * c.x), xy - m * c.x * c.y, xz - m * c.x * c.z, xy - m * c.x * c.y, yy + m * (cc - c.y * c.y), yz - m * c.y * c.z,
            xz - m * c.x * c.z, yz - m * c.y * c.z, zz + m * (cc - c.z * c.z)}};
    o.B = {{0.0, -m * c.z, m * c.y, m * c.z, 0.0, -m * c.x, -m * c.y, m * c.x, 0.0}};
    o.C = {{m, 0.0, 0.0, 0.0, m, 0.0, 0.0, 0.0, m}};
}
WB_FN void velocity_product(const Art& I, V3 w, V3 vl, V3& n, V3& f)
{
    const V3 hn = mul(I.A, w) + mul(I.B, vl), hf = mulT(I.B, w) + mul(I.C, vl); // I v
    n = cross(w, hn) + cross(vl, hf);                                            // v x* (I v)
    f = cross(w, hf);
}

struct LinkSave {
    double c, s, d, u;
    V3 Ua, Ul, ca, cl;
};

template <int FIRST, int D, int LEN, int FOOT>
struct AbaChain {
    // pass 1 + 2 for link FIRST + D and everything below it; hands the articulated inertia and bias force of the
    // subtree back in PARENT coordinates about the parent's origin
    static WB_FN void inward(const Eval& e, const double* tau, V3 w, V3 vl, V3 fe, LinkSave* sv, Art& up)
    {
        constexpr int I = FIRST + D;
        constexpr int AX = b2z1::AXIS[I];
        const V3 p = origin<I>();
        LinkSave& L = sv[D];
        e.SC(5 + I, &L.s, &L.c);
        const double qd = e.V(5 + I);
        const V3 ax = unit<AX>();
        const V3 wi = rotT<AX>(L.c, L.s, w) + qd * ax;
        const V3 vi = rotT<AX>(L.c, L.s, vl + cross(w, p));
        L.ca = qd * cross(wi, ax); // v x (S qd)
        L.cl = qd * cross(vi, ax);
        const V3 fei = rotT<AX>(L.c, L.s, fe);
        Art a;
        body_inertia<I>(a);
        velocity_product(a, wi, vi, a.n, a.f);
        if (D + 1 < LEN) {
            Art ch;
            AbaChain<FIRST, (D + 1 < LEN ? D + 1 : D), LEN, FOOT>::inward(e, tau, wi, vi, fei, sv, ch);
            a.A = add(a.A, ch.A); a.B = add(a.B, ch.B); a.C = add(a.C, ch.C);
            a.n = a.n + ch.n; a.f = a.f + ch.f;
        } else if (FOOT >= 0) {
            const V3 r = {b2z1::FOOT_POINT[3 * (FOOT >= 0 ? FOOT : 0)], b2z1::FOOT_POINT[3 * (FOOT >= 0 ? FOOT : 0) + 1],
                          b2z1::FOOT_POINT[3 * (FOOT >= 0 ? FOOT : 0) + 2]};
            a.n = a.n - cross(r, fei);
            a.f = a.f - fei;
        }
        // U = IA S (S = angular unit vector AX), d = S' U, u = tau - S' pA
        L.Ua = col(a.A, AX);
        L.Ul = row(a.B, AX); // (B' e)_k = B[AX][k]
        L.d = comp<AX>(L.Ua);
        L.u = tau[I - 1] - comp<AX>(a.n);
        const double id = 1.0 / L.d;
        // Ia = IA - U U' / d ;  pa = pA + Ia c + U u / d
        a.A = add(a.A, scale(-id, outer(L.Ua, L.Ua)));
        a.B = add(a.B, scale(-id, outer(L.Ua, L.Ul)));
        a.C = add(a.C, scale(-id, outer(L.Ul, L.Ul)));
        const double ud = L.u * id;
        a.n = a.n + mul(a.A, L.ca) + mul(a.B, L.cl) + ud * L.Ua;
        a.f = a.f + mulT(a.B, L.ca) + mul(a.C, L.cl) + ud * L.Ul;
        // to the parent: rotate, then shift the reference point by p:  X' Ia X with X = [1 0; -p x 1]
        const M3 A = rotate<AX>(L.c, L.s, a.A), B = rotate<AX>(L.c, L.s, a.B), C = rotate<AX>(L.c, L.s, a.C);
        const V3 n = rot<AX>(L.c, L.s, a.n), f = rot<AX>(L.c, L.s, a.f);
        const M3 pC = cross_left(p, C);                  // p x C
        const M3 Bn = add(B, pC);                        // B + p x C
        // A + p x B' - B p x - p x C p x  =  A + p x B' - (B + p x C) p x
        up.A = add(add(A, cross_left(p, transpose(B))), scale(-1.0, cross_right(Bn, p)));
        up.B = Bn;
        up.C = C;
        up.f = f;
        up.n = n + cross(p, f);
    }
    // pass 3: accelerations
    static WB_FN void outward(const LinkSave* sv, V3 aa, V3 al, const Sink& out)
    {
        constexpr int I = FIRST + D;
        constexpr int AX = b2z1::AXIS[I];
        const V3 p = origin<I>();
        const LinkSave& L = sv[D];
        const V3 ax = unit<AX>();
        V3 aai = rotT<AX>(L.c, L.s, aa) + L.ca;
        const V3 ali = rotT<AX>(L.c, L.s, al + cross(aa, p)) + L.cl;
        const double qdd = (L.u - (L.Ua.x * aai.x + L.Ua.y * aai.y + L.Ua.z * aai.z) - (L.Ul.x * ali.x + L.Ul.y * ali.y + L.Ul.z * ali.z)) / L.d;
        out.put(5 + I, qdd);
        aai = aai + qdd * ax;
        if (D + 1 < LEN) AbaChain<FIRST, (D + 1 < LEN ? D + 1 : D), LEN, FOOT>::outward(sv, aai, ali, out);
    }
};

// x <- S^-1 x for the symmetric positive definite 6 x 6 matrix [[A, B], [B', C]] (Cholesky, unrolled)
WB_FN void solve6(const Art& a, double* x)
{
    double m[6][6];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            m[i][j] = a.A.a[3 * i + j]; m[i][3 + j] = a.B.a[3 * i + j]; m[3 + i][j] = a.B.a[3 * j + i]; m[3 + i][3 + j] = a.C.a[3 * i + j];
        }
    for (int k = 0; k < 6; ++k) {
        m[k][k] = sqrt(m[k][k]);
        for (int i = k + 1; i < 6; ++i) m[i][k] /= m[k][k];
        for (int j = k + 1; j < 6; ++j)
            for (int i = j; i < 6; ++i) m[i][j] -= m[i][k] * m[j][k];
    }
    for (int i = 0; i < 6; ++i) { for (int j = 0; j < i; ++j) x[i] -= m[i][j] * x[j]; x[i] /= m[i][i]; }
    for (int i = 5; i >= 0; --i) { for (int j = i + 1; j < 6; ++j) x[i] -= m[j][i] * x[j]; x[i] /= m[i][i]; }
}

// the 24 generalised accelerations for joint torques tau[18] at the evaluation point (e.A is not read)
WB_FN void aba(const Eval& e, const double* tau, const Sink& out)
{
    BaseRot R;
    R.set(e);
    const V3 w0 = {e.V(0), e.V(1), e.V(2)}, v0 = {e.V(3), e.V(4), e.V(5)};
    Art base;
    body_inertia<0>(base);
    velocity_product(base, w0, v0, base.n, base.f);
    LinkSave s0[3], s1[3], s2[3], s3[3], s4[6];
    Art ch;
#define WB_ABA_LEG(FIRST, K, SV)                                                                                        \
    AbaChain<FIRST, 0, 3, K>::inward(e, tau, w0, v0, R.toBase({e.F(3 * K), e.F(3 * K + 1), e.F(3 * K + 2)}), SV, ch);      \
    base.A = add(base.A, ch.A); base.B = add(base.B, ch.B); base.C = add(base.C, ch.C);                                  \
    base.n = base.n + ch.n; base.f = base.f + ch.f;                                                                      \
    WB_FENCE
    WB_ABA_LEG(1, 0, s0)
    WB_ABA_LEG(4, 1, s1)
    WB_ABA_LEG(7, 2, s2)
    WB_ABA_LEG(10, 3, s3)
#undef WB_ABA_LEG
    AbaChain<13, 0, 6, -1>::inward(e, tau, w0, v0, {0.0, 0.0, 0.0}, s4, ch);
    base.A = add(base.A, ch.A); base.B = add(base.B, ch.B); base.C = add(base.C, ch.C);
    base.n = base.n + ch.n; base.f = base.f + ch.f;
    double a0[6] = {-base.n.x, -base.n.y, -base.n.z, -base.f.x, -base.f.y, -base.f.z};
    solve6(base, a0);
    const V3 aa = {a0[0], a0[1], a0[2]}, al = {a0[3], a0[4], a0[5]};
    AbaChain<1, 0, 3, 0>::outward(s0, aa, al, out);
    AbaChain<4, 0, 3, 1>::outward(s1, aa, al, out);
    AbaChain<7, 0, 3, 2>::outward(s2, aa, al, out);
    AbaChain<10, 0, 3, 3>::outward(s3, aa, al, out);
    AbaChain<13, 0, 6, -1>::outward(s4, aa, al, out);
    const V3 gb = R.toBase({0.0, 0.0, e.g});
    out.put(0, aa.x); out.put(1, aa.y); out.put(2, aa.z);
    out.put(3, al.x - gb.x); out.put(4, al.y - gb.y); out.put(5, al.z - gb.z);
}

} // namespace wb
