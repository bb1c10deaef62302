// CPU harness for csrc/minco_spline.h (the knot-state spline the kernels use), built by tests/test_minco_spline_cpu.py.
#include <vector>

#include "../../alore_legged_manipulator_amd/csrc/minco_spline.h"

extern "C" {
// coef out: (6 i + q) * 2 + d, like oracle/backend_oracle.c be_spline
void harness_spline(int M, const double* T, const double* inner, const double* head /* [d][p v a] */, const double* tail,
                    double* coef)
{
    std::vector<double> p(M + 1), y(2 * (M > 1 ? M - 1 : 1)), cf(6 * M);
    std::vector<minco::Sym2> sinv(M);
    for (int d = 0; d < 2; ++d) {
        p[0] = head[d * 3];
        for (int k = 1; k < M; ++k) p[k] = inner[(k - 1) * 2 + d];
        p[M] = tail[d * 3];
        minco::spline_1d(M, T, p.data(), head[d * 3 + 1], head[d * 3 + 2], tail[d * 3 + 1], tail[d * 3 + 2], sinv.data(), y.data(), cf.data());
        for (int i = 0; i < 6 * M; ++i) coef[i * 2 + d] = cf[i];
    }
}
// d coef / d T of piece i with the knot states fixed, and the adjoint of the Hermite map
void harness_hermite(double T, const double* z0, const double* z1, double* c, double* dc)
{
    const minco::InvT q(T);
    minco::hermite(T, q, z0[0], z0[1], z0[2], z1[0], z1[1], z1[2], c, dc);
}
void harness_hermite_adjoint(double T, const double* G, double* g0, double* g1)
{
    const minco::InvT q(T);
    minco::hermite_adjoint(q, G, g0, g1);
}
}
