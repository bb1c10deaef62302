// tools/ws_guess_experiment.cpp -- EXPERIMENT TOOL (host only, not part of the product).
//
// Costs working-set guesses for the cold start of the stage-wise QP solver on the CPU, with the functions of
// csrc/nmpc_core.h, before anything is built into the kernel: for every problem of a batch it runs the primal-dual
// working-set iteration from a guess and records how many sweeps follow the first one and from which stage each restarts.
//   mode 0: no guess (every control free)
//   mode 1: guess made INSIDE the first backward sweep: at stage k the two-control box QP of the stage is solved at an
//           estimate of the state step (dx_est = 0, or the free response), the controls that land on a bound are fixed
//           before the stage is eliminated
//   mode 2: as 1, then the forward sweep's primal-dual update (what the kernel does anyway)
// Driver: tools/ws_guess_experiment.py.
#include <cmath>
#include <cstring>
#include <vector>

#include "../alore_legged_manipulator_amd/csrc/nmpc_core.h"

using namespace nmpc;

namespace {

// the quantities riccati_step forms before the eliminations, for the stage box QP
struct StageHess {
    float H00, H01, H11, hu0, hu1, G00, G01, G02, G10, G11, G12;
};
void stage_hessian(const StageQP& s, const Value& V, StageHess& o)
{
    const Sym3 P = V.P;
    const float B21 = -s.B20;
    const float s0 = P.m00 * s.d0 + P.m01 * s.d1 + P.m02 * s.d2 + V.p0;
    const float s1 = P.m01 * s.d0 + P.m11 * s.d1 + P.m12 * s.d2 + V.p1;
    const float s2 = P.m02 * s.d0 + P.m12 * s.d1 + P.m22 * s.d2 + V.p2;
    const float PB00 = P.m00 * s.B00 + P.m01 * s.B10 + P.m02 * s.B20;
    const float PB10 = P.m01 * s.B00 + P.m11 * s.B10 + P.m12 * s.B20;
    const float PB20 = P.m02 * s.B00 + P.m12 * s.B10 + P.m22 * s.B20;
    const float PB01 = P.m00 * s.B01 + P.m01 * s.B11 + P.m02 * B21;
    const float PB11 = P.m01 * s.B01 + P.m11 * s.B11 + P.m12 * B21;
    const float PB21 = P.m02 * s.B01 + P.m12 * s.B11 + P.m22 * B21;
    o.H00 = s.R00 + s.B00 * PB00 + s.B10 * PB10 + s.B20 * PB20;
    o.H01 = s.R01 + s.B00 * PB01 + s.B10 * PB11 + s.B20 * PB21;
    o.H11 = s.R11 + s.B01 * PB01 + s.B11 * PB11 + B21 * PB21;
    o.G00 = PB00; o.G01 = PB10; o.G02 = s.a * PB00 + s.b * PB10 + PB20;
    o.G10 = PB01; o.G11 = PB11; o.G12 = s.a * PB01 + s.b * PB11 + PB21;
    o.hu0 = s.r0 + s.B00 * s0 + s.B10 * s1 + s.B20 * s2;
    o.hu1 = s.r1 + s.B01 * s0 + s.B11 * s1 + B21 * s2;
}

int at(float v, float lb, float ub) { return (v <= lb) ? ST_LOWER : ((v >= ub) ? ST_UPPER : ST_FREE); }

} // namespace

extern "C" {

// returns the number of sweeps; restarts[i] = stage the (i+2)-th sweep restarted from (khi); up to 32 recorded
int ws_experiment(int N, float dt, const float* x, const float* u, const float* od, const float* y, const float* yN,
                  const float* W, const float* WN, const float* x0, const float* lbV, const float* ubV, int mode,
                  int est, int passes, int* restarts, int* n_active_out)
{
    const IrkConst K = make_irk(dt);
    std::vector<StageQP> st(N);
    std::vector<Policy> pol(N);
    std::vector<Value> Vs(N + 1);
    std::vector<float> lb0(N), lb1(N), ub0(N), ub1(N), sb(3 * (N + 1));
    Sym3 QN;
    float qN[3];
    for (int k = 0; k < N; ++k) {
        StageLin l;
        const float px = x[k * 3], py = x[k * 3 + 1], ps = x[k * 3 + 2], vr = u[k * 2], vl = u[k * 2 + 1];
        ddr_linearize(K, px, py, ps, vr, vl, od[k * 3], od[k * 3 + 1], od[k * 3 + 2], l);
        StageQP& s = st[k];
        s.a = l.a; s.b = l.b; s.B00 = l.B00; s.B01 = l.B01; s.B10 = l.B10; s.B11 = l.B11; s.B20 = l.B20;
        s.d0 = l.phi0 - x[k * 3 + 3]; s.d1 = l.phi1 - x[k * 3 + 4]; s.d2 = l.phi2 - x[k * 3 + 5];
        const float* yk = y + k * 5;
        const float* w = W + k * 25;
        const float e0 = px - yk[0], e1 = py - yk[1], e2 = ps - yk[2], e3 = vr - yk[3], e4 = vl - yk[4];
        s.q0 = w[0] * e0 + w[1] * e1 + w[2] * e2 + w[3] * e3 + w[4] * e4;
        s.q1 = w[5] * e0 + w[6] * e1 + w[7] * e2 + w[8] * e3 + w[9] * e4;
        s.q2 = w[10] * e0 + w[11] * e1 + w[12] * e2 + w[13] * e3 + w[14] * e4;
        s.r0 = w[15] * e0 + w[16] * e1 + w[17] * e2 + w[18] * e3 + w[19] * e4;
        s.r1 = w[20] * e0 + w[21] * e1 + w[22] * e2 + w[23] * e3 + w[24] * e4;
        s.Q.m00 = w[0]; s.Q.m01 = w[1]; s.Q.m02 = w[2]; s.Q.m11 = w[6]; s.Q.m12 = w[7]; s.Q.m22 = w[12];
        s.R00 = w[18]; s.R01 = w[19]; s.R11 = w[24];
        lb0[k] = lbV[k * 2] - vr; lb1[k] = lbV[k * 2 + 1] - vl;
        ub0[k] = ubV[k * 2] - vr; ub1[k] = ubV[k * 2 + 1] - vl;
        s.st0 = ST_FREE; s.st1 = ST_FREE;
    }
    {
        const float e0 = x[N * 3] - yN[0], e1 = x[N * 3 + 1] - yN[1], e2 = x[N * 3 + 2] - yN[2];
        qN[0] = WN[0] * e0 + WN[1] * e1 + WN[2] * e2;
        qN[1] = WN[3] * e0 + WN[4] * e1 + WN[5] * e2;
        qN[2] = WN[6] * e0 + WN[7] * e1 + WN[8] * e2;
        QN.m00 = WN[0]; QN.m01 = WN[1]; QN.m02 = WN[2]; QN.m11 = WN[4]; QN.m12 = WN[5]; QN.m22 = WN[8];
        Vs[N].P = QN; Vs[N].p0 = qN[0]; Vs[N].p1 = qN[1]; Vs[N].p2 = qN[2];
    }
    const float Dx0 = x0[0] - x[0], Dx1 = x0[1] - x[1], Dx2 = x0[2] - x[2];
    { // free response
        float a0 = Dx0, a1 = Dx1, a2 = Dx2;
        for (int k = 0; k <= N; ++k) {
            sb[3 * k] = a0; sb[3 * k + 1] = a1; sb[3 * k + 2] = a2;
            if (k < N) {
                const float m0 = a0 + st[k].a * a2 + st[k].d0, m1 = a1 + st[k].b * a2 + st[k].d1, m2 = a2 + st[k].d2;
                a0 = m0; a1 = m1; a2 = m2;
            }
        }
    }
    if (est >= 100) { // Barzilai-Borwein projected-gradient prediction of the kernel, est - 100 = pg_steps (min), passes = max steps
        const int pg_steps = est - 100, max_steps = passes;
        std::vector<float> w0(N, 0.f), w1(N, 0.f), g0(N), g1(N), pu0(N, 0.f), pu1(N, 0.f), pg0(N), pg1(N), is0(N), is1(N);
        std::vector<int> bits(N);
        auto apply = [&]() {
            std::vector<float> X(3 * (N + 1)), lam(3 * (N + 2), 0.f);
            X[0] = Dx0; X[1] = Dx1; X[2] = Dx2;
            for (int k = 0; k < N; ++k) {
                const StageQP& s = st[k];
                X[3 * k + 3] = X[3 * k] + s.a * X[3 * k + 2] + s.B00 * w0[k] + s.B01 * w1[k] + s.d0;
                X[3 * k + 4] = X[3 * k + 1] + s.b * X[3 * k + 2] + s.B10 * w0[k] + s.B11 * w1[k] + s.d1;
                X[3 * k + 5] = X[3 * k + 2] + s.B20 * (w0[k] - w1[k]) + s.d2;
            }
            for (int k = N; k >= 1; --k) {
                const Sym3& Q = (k < N) ? st[k].Q : QN;
                const float q0 = (k < N) ? st[k].q0 : qN[0], q1 = (k < N) ? st[k].q1 : qN[1], q2 = (k < N) ? st[k].q2 : qN[2];
                const float* xx = &X[3 * k];
                const float a = (k < N) ? st[k].a : 0.f, b = (k < N) ? st[k].b : 0.f;
                const float* ln = &lam[3 * (k + 1)];
                lam[3 * k] = Q.m00 * xx[0] + Q.m01 * xx[1] + Q.m02 * xx[2] + q0 + ln[0];
                lam[3 * k + 1] = Q.m01 * xx[0] + Q.m11 * xx[1] + Q.m12 * xx[2] + q1 + ln[1];
                lam[3 * k + 2] = Q.m02 * xx[0] + Q.m12 * xx[1] + Q.m22 * xx[2] + q2 + ln[2] + a * ln[0] + b * ln[1];
            }
            for (int k = 0; k < N; ++k) {
                const StageQP& s = st[k];
                const float* l = &lam[3 * (k + 1)];
                g0[k] = s.R00 * w0[k] + s.R01 * w1[k] + s.r0 + s.B00 * l[0] + s.B10 * l[1] + s.B20 * l[2];
                g1[k] = s.R01 * w0[k] + s.R11 * w1[k] + s.r1 + s.B01 * l[0] + s.B11 * l[1] - s.B20 * l[2];
            }
        };
        auto atb = [&](int k) { return (w0[k] <= lb0[k] ? 1 : 0) | (w0[k] >= ub0[k] ? 2 : 0) | (w1[k] <= lb1[k] ? 4 : 0) | (w1[k] >= ub1[k] ? 8 : 0); };
        apply();
        bool hits = false;
        for (int k = 0; k < N; ++k) {
            is0[k] = 1.0f / st[k].R00; is1[k] = 1.0f / st[k].R11;
            const float j0 = -is0[k] * g0[k], j1 = -is1[k] * g1[k];
            hits = hits || j0 < lb0[k] || j0 > ub0[k] || j1 < lb1[k] || j1 > ub1[k];
            w0[k] = clampf(j0, lb0[k], ub0[k]); w1[k] = clampf(j1, lb1[k], ub1[k]);
        }
        if (hits && max_steps > 0) {
            for (int k = 0; k < N; ++k) { pg0[k] = g0[k]; pg1[k] = g1[k]; bits[k] = atb(k); }
            float alpha = 1.0f;
            int still = 0;
            for (int t = 1; t < max_steps; ++t) {
                apply();
                float num = 0, den = 0;
                for (int k = 0; k < N; ++k) {
                    const float e0 = w0[k] - pu0[k], e1 = w1[k] - pu1[k];
                    num += st[k].R00 * e0 * e0 + st[k].R11 * e1 * e1;
                    den += e0 * (g0[k] - pg0[k]) + e1 * (g1[k] - pg1[k]);
                }
                alpha = (den > 1e-30f) ? fminf(fmaxf(num / den, 1e-3f), 1.0f) : alpha;
                bool moved = false;
                for (int k = 0; k < N; ++k) {
                    pu0[k] = w0[k]; pu1[k] = w1[k]; pg0[k] = g0[k]; pg1[k] = g1[k];
                    w0[k] = clampf(w0[k] - alpha * is0[k] * g0[k], lb0[k], ub0[k]);
                    w1[k] = clampf(w1[k] - alpha * is1[k] * g1[k], lb1[k], ub1[k]);
                    const int nb = atb(k);
                    moved = moved || nb != bits[k];
                    bits[k] = nb;
                }
                still = moved ? 0 : still + 1;
                if (still >= 2 && t + 1 >= pg_steps) break;
            }
            for (int k = 0; k < N; ++k) { st[k].st0 = at(w0[k], lb0[k], ub0[k]); st[k].st1 = at(w1[k], lb1[k], ub1[k]); }
        }
        est = 0; passes = 0;
    }
    std::vector<float> dxe(3 * (N + 1), 0.0f); // estimate of the state step used by the in-sweep guess
    if (est == 1) dxe = sb;
    int it = 0, khi = N - 1, n_restarts = 0;
    bool changed = true;
    for (;;) {
        Value V = Vs[khi + 1];
        for (int k = khi; k >= 0; --k) {
            StageQP& s = st[k];
            if (mode >= 1 && it < passes) {
                StageHess h;
                stage_hessian(s, V, h);
                const float* e = &dxe[3 * k];
                const float g0 = h.hu0 + h.G00 * e[0] + h.G01 * e[1] + h.G02 * e[2];
                const float g1 = h.hu1 + h.G10 * e[0] + h.G11 * e[1] + h.G12 * e[2];
                // two-control box QP: min 1/2 v'Hv + g'v, v in the box.  Unconstrained first, then the faces.
                const float det = h.H00 * h.H11 - h.H01 * h.H01;
                float v0 = -(h.H11 * g0 - h.H01 * g1) / det, v1 = -(h.H00 * g1 - h.H01 * g0) / det;
                int a0 = ST_FREE, a1 = ST_FREE;
                const bool out0 = v0 < lb0[k] || v0 > ub0[k], out1 = v1 < lb1[k] || v1 > ub1[k];
                if (out0 || out1) {
                    // candidate: control 0 clipped, control 1 re-solved (and the other way round); pick the lower cost
                    float best = 3e38f;
                    for (int c = 0; c < 3; ++c) {
                        float w0, w1;
                        if (c == 0) { w0 = clampf(v0, lb0[k], ub0[k]); w1 = clampf(-(g1 + h.H01 * w0) / h.H11, lb1[k], ub1[k]); }
                        else if (c == 1) { w1 = clampf(v1, lb1[k], ub1[k]); w0 = clampf(-(g0 + h.H01 * w1) / h.H00, lb0[k], ub0[k]); }
                        else { w0 = clampf(v0, lb0[k], ub0[k]); w1 = clampf(v1, lb1[k], ub1[k]); }
                        const float cost = 0.5f * (h.H00 * w0 * w0 + 2 * h.H01 * w0 * w1 + h.H11 * w1 * w1) + g0 * w0 + g1 * w1;
                        if (cost < best) { best = cost; a0 = at(w0, lb0[k], ub0[k]); a1 = at(w1, lb1[k], ub1[k]); }
                    }
                }
                s.st0 = a0; s.st1 = a1;
            }
            s.v0 = (s.st0 == ST_UPPER) ? ub0[k] : lb0[k];
            s.v1 = (s.st1 == ST_UPPER) ? ub1[k] : lb1[k];
            riccati_step(s, V, pol[k], k > 0);
            if (k > 0) Vs[k] = V;
        }
        float dx0 = Dx0, dx1 = Dx1, dx2 = Dx2;
        int new_khi = -1;
        for (int k = 0; k < N; ++k) {
            StageQP& s = st[k];
            StageStep o;
            forward_step(pol[k], s.st0, s.st1, dx0, dx1, dx2, lb0[k], ub0[k], lb1[k], ub1[k], o);
            if (o.nst0 != s.st0 || o.nst1 != s.st1) new_khi = k;
            if (mode == 3 && new_khi >= 0) { // this pass is no longer the answer: roll on with the controls a box allows
                if (s.st0 == ST_FREE) o.du0 = clampf(o.du0, lb0[k], ub0[k]);
                if (s.st1 == ST_FREE) {
                    // control 1 given the clipped control 0
                    const float val1 = pol[k].c10 * dx0 + pol[k].c11 * dx1 + pol[k].c12 * dx2 + pol[k].e1 * o.du0 + pol[k].f1;
                    o.du1 = clampf(val1, lb1[k], ub1[k]);
                    o.nst1 = next_status(s.st1, val1, lb1[k], ub1[k]);
                }
            }
            dxe[3 * k] = dx0; dxe[3 * k + 1] = dx1; dxe[3 * k + 2] = dx2;
            const float n0 = dx0 + s.a * dx2 + s.B00 * o.du0 + s.B01 * o.du1 + s.d0;
            const float n1 = dx1 + s.b * dx2 + s.B10 * o.du0 + s.B11 * o.du1 + s.d1;
            const float n2 = dx2 + s.B20 * (o.du0 - o.du1) + s.d2;
            dx0 = n0; dx1 = n1; dx2 = n2;
            s.st0 = o.nst0; s.st1 = o.nst1;
        }
        ++it;
        changed = new_khi >= 0;
        if (!changed || it >= 60) break;
        khi = (mode >= 1 && it < passes) ? N - 1 : new_khi; // a guessing pass is a full sweep
        if (n_restarts < 32) restarts[n_restarts] = khi;
        ++n_restarts;
    }
    int na = 0;
    for (int k = 0; k < N; ++k) na += (st[k].st0 != ST_FREE) + (st[k].st1 != ST_FREE);
    *n_active_out = na;
    return it;
}
}
