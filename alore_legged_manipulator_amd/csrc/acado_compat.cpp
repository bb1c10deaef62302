// acado_compat.cpp -- the reference's single-instance ACADO symbols on top of the batched GPU engine
// (B = 1).  See include/alore_acado_compat.h.  The split preparation / feedback semantics of the
// reference are kept: acado_preparationStep() remembers the iterate and the weights it linearised,
// acado_feedbackStep() solves with that linearisation and expands whatever is in acadoVariables.x/u
// at that moment (alore_nmpc_set_linearization_point), like acado_solver.c:1057-1077.
#include "../../include/alore_acado_compat.h"

#include <cstdlib>

#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdio>
#include <cstring>

#include "../../include/alore_nmpc.h"

namespace {
constexpr int N = ACADO_N;
alore_nmpc_handle g_h = nullptr;
alore_nmpc_batch g_dev{};
float *g_lin_x = nullptr, *g_lin_u = nullptr; // device copies of the prepared iterate
float *g_d = nullptr, *g_gx = nullptr, *g_gu = nullptr;
float *g_H = nullptr, *g_g = nullptr, *g_lb = nullptr, *g_ub = nullptr, *g_qx = nullptr, *g_qy = nullptr; // the condensed QP on the device
int* g_qst = nullptr; // [2]: status, factorisations
int g_device = 0;
int g_nwsr = 0;
float g_kkt = 0.0f, g_obj = 0.0f;
// what the last preparation step saw (the reference evaluates these at preparation time)
ACADOvariables g_prep;
bool g_have_prep = false;

bool ok(int rc) { return rc == ALORE_NMPC_OK; }

bool ensure()
{
    if (g_h) return true;
    alore_nmpc_config cfg{N, 0.01f, g_device, 300, 0, -1};
    if (!ok(alore_nmpc_create(&cfg, &g_h))) {
        std::fprintf(stderr, "[alore_acado_compat] alore_nmpc_create failed: no usable GPU (there is no CPU path)\n");
        g_h = nullptr;
        return false;
    }
    if (!ok(alore_nmpc_batch_alloc(g_h, 1, &g_dev))) return false;
    if (hipMalloc((void**)&g_lin_x, sizeof(float) * 3 * (N + 1)) != hipSuccess) return false;
    if (hipMalloc((void**)&g_lin_u, sizeof(float) * 2 * N) != hipSuccess) return false;
    if (hipMalloc((void**)&g_d, sizeof(float) * 3 * N) != hipSuccess) return false;
    if (hipMalloc((void**)&g_gx, sizeof(float) * 9 * N) != hipSuccess) return false;
    if (hipMalloc((void**)&g_gu, sizeof(float) * 6 * N) != hipSuccess) return false;
    if (hipMalloc((void**)&g_H, sizeof(float) * ACADO_QP_NV * ACADO_QP_NV) != hipSuccess) return false;
    if (hipMalloc((void**)&g_g, sizeof(float) * ACADO_QP_NV * 5) != hipSuccess) return false; // g | lb | ub | x | y
    g_lb = g_g + ACADO_QP_NV; g_ub = g_lb + ACADO_QP_NV; g_qx = g_ub + ACADO_QP_NV; g_qy = g_qx + ACADO_QP_NV;
    if (hipMalloc((void**)&g_qst, sizeof(int) * 2) != hipSuccess) return false;
    return true;
}

void upload_variables(const ACADOvariables& v, bool iterate, bool dual)
{
    alore_nmpc_batch host{};
    if (iterate) { host.x = const_cast<float*>(v.x); host.u = const_cast<float*>(v.u); }
    host.od = v.od; host.y = v.y; host.yN = v.yN; host.W = v.W; host.WN = v.WN; host.x0 = v.x0;
    host.lbValues = v.lbValues; host.ubValues = v.ubValues;
    if (dual) host.dual = acadoWorkspace.y;
    alore_nmpc_batch_upload(g_h, &g_dev, &host, 1, nullptr);
}
// the condensed QP of the batch on the device -> acadoWorkspace.H, g (what condensePrep / condenseFdb leave there)
int g_dense = -1; // -1: not decided yet (environment), 0 / 1
// false: the condensed QP could not be formed -- acadoWorkspace.H / g would be stale, the step reports it (29, as for a lost device)
bool condense_to_workspace(bool with_g)
{
    if (g_dense < 0) { const char* e = std::getenv("ALORE_ACADO_DENSE_WORKSPACE"); g_dense = (e && e[0] == '0') ? 0 : 1; }
    if (!g_dense) return true;
    alore_nmpc_dense_qp_data q{g_H, g_g, g_lb, g_ub};
    if (!ok(alore_nmpc_condense(g_h, &g_dev, 1, &q, nullptr))) {
        std::fprintf(stderr, "[alore_acado_compat] acadoWorkspace.H / g not refreshed: %s\n", alore_nmpc_last_error(g_h));
        return false;
    }
    bool copied = hipMemcpyAsync(acadoWorkspace.H, g_H, sizeof(float) * ACADO_QP_NV * ACADO_QP_NV, hipMemcpyDeviceToHost, nullptr) == hipSuccess;
    if (with_g) copied = copied && hipMemcpyAsync(acadoWorkspace.g, g_g, sizeof(float) * ACADO_QP_NV, hipMemcpyDeviceToHost, nullptr) == hipSuccess;
    return copied;
}
} // namespace

extern "C" {

void alore_acado_set_device(int device) { g_device = device; }
void alore_acado_dense_workspace(int enable) { g_dense = enable ? 1 : 0; }

void alore_acado_shutdown(void)
{
    if (!g_h) return;
    alore_nmpc_batch_free(g_h, &g_dev);
    (void)hipFree(g_lin_x); (void)hipFree(g_lin_u); (void)hipFree(g_d); (void)hipFree(g_gx); (void)hipFree(g_gu);
    (void)hipFree(g_H); (void)hipFree(g_g); (void)hipFree(g_qst);
    alore_nmpc_destroy(g_h);
    g_h = nullptr;
    g_have_prep = false;
}

int acado_initializeSolver(void)
{
    std::memset(&acadoWorkspace, 0, sizeof(acadoWorkspace));
    for (int i = 0; i < ACADO_QP_NV; ++i) { // the bounds baked by the generator (UAV_CAR_model.cpp:97-101)
        acadoVariables.lbValues[i] = -3.0f;
        acadoVariables.ubValues[i] = 3.0f;
    }
    g_have_prep = false;
    return ensure() ? 0 : 29; // RET_INIT_FAILED
}

void acado_initializeNodesByForwardSimulation(void)
{
    if (!ensure()) return;
    upload_variables(acadoVariables, true, false);
    alore_nmpc_forward_simulate(g_h, &g_dev, 1, nullptr);
    alore_nmpc_batch out{};
    out.x = acadoVariables.x;
    alore_nmpc_batch_download(g_h, &g_dev, &out, 1, nullptr);
    (void)hipStreamSynchronize(nullptr);
}

int acado_preparationStep(void)
{
    if (!ensure()) return 2;
    g_prep = acadoVariables; // x, u, od, W, WN as the reference's preparation sees them
    g_have_prep = true;
    // fill the preparation-side workspace members the reference exposes
    upload_variables(acadoVariables, true, false);
    alore_nmpc_lin_out lo{g_d, g_gx, g_gu};
    alore_nmpc_linearize(g_h, &g_dev, 1, &lo, nullptr);
    (void)hipMemcpyAsync(acadoWorkspace.d, g_d, sizeof(float) * 3 * N, hipMemcpyDeviceToHost, nullptr);
    (void)hipMemcpyAsync(acadoWorkspace.evGx, g_gx, sizeof(float) * 9 * N, hipMemcpyDeviceToHost, nullptr);
    (void)hipMemcpyAsync(acadoWorkspace.evGu, g_gu, sizeof(float) * 6 * N, hipMemcpyDeviceToHost, nullptr);
    const bool dense_ok = condense_to_workspace(false); // acado_condensePrep: the Hessian of the condensed QP belongs to the preparation
    (void)hipStreamSynchronize(nullptr);
    return dense_ok ? 0 : 29;
}

int acado_feedbackStep(void)
{
    if (!ensure()) return 29;
    if (!g_have_prep) { g_prep = acadoVariables; g_have_prep = true; }
    // linearisation data (od, W, WN) from preparation time; y, yN, x0, bounds and the iterate from now
    ACADOvariables v = acadoVariables;
    std::memcpy(v.od, g_prep.od, sizeof(v.od));
    std::memcpy(v.W, g_prep.W, sizeof(v.W));
    std::memcpy(v.WN, g_prep.WN, sizeof(v.WN));
    upload_variables(v, true, true);
    (void)hipMemcpyAsync(g_lin_x, g_prep.x, sizeof(g_prep.x), hipMemcpyHostToDevice, nullptr);
    (void)hipMemcpyAsync(g_lin_u, g_prep.u, sizeof(g_prep.u), hipMemcpyHostToDevice, nullptr);
    alore_nmpc_set_linearization_point(g_h, g_lin_x, g_lin_u);
    float u_old[ACADO_QP_NV];
    std::memcpy(u_old, acadoVariables.u, sizeof(u_old));
    for (int i = 0; i < 3; ++i) acadoWorkspace.Dx0[i] = acadoVariables.x0[i] - acadoVariables.x[i];
    for (int i = 0; i < ACADO_QP_NV; ++i) {
        acadoWorkspace.lb[i] = acadoVariables.lbValues[i] - acadoVariables.u[i];
        acadoWorkspace.ub[i] = acadoVariables.ubValues[i] - acadoVariables.u[i];
    }
    const bool dense_ok = condense_to_workspace(true); // acado_condenseFdb: gradient of the condensed QP for the measured state (H again: same linearisation)
    const int rc = alore_nmpc_rti(g_h, &g_dev, 1, 1, nullptr);
    alore_nmpc_set_linearization_point(g_h, nullptr, nullptr);
    int status = 0;
    alore_nmpc_batch out{};
    out.x = acadoVariables.x; out.u = acadoVariables.u; out.dual = acadoWorkspace.y;
    out.status = &status; out.n_iter = &g_nwsr; out.kkt = &g_kkt; out.obj = &g_obj;
    alore_nmpc_batch_download(g_h, &g_dev, &out, 1, nullptr);
    (void)hipStreamSynchronize(nullptr);
    for (int i = 0; i < ACADO_QP_NV; ++i) acadoWorkspace.x[i] = acadoVariables.u[i] - u_old[i];
    g_have_prep = false;
    return (ok(rc) && dense_ok) ? status : 29;
}

void acado_shiftStates(int strategy, real_t* const xEnd, real_t* const uEnd)
{
    // the states shift alone; acado_shiftControls is a separate call in the reference ABI
    for (int i = 0; i < 3 * N; ++i) acadoVariables.x[i] = acadoVariables.x[i + 3];
    if (strategy == 1 && xEnd) {
        for (int i = 0; i < 3; ++i) acadoVariables.x[3 * N + i] = xEnd[i];
    } else if (strategy == 2) {
        real_t eta[23] = {0};
        for (int i = 0; i < 3; ++i) eta[i] = acadoVariables.x[3 * N + i];
        eta[18] = uEnd ? uEnd[0] : acadoVariables.u[2 * (N - 1)];
        eta[19] = uEnd ? uEnd[1] : acadoVariables.u[2 * (N - 1) + 1];
        for (int i = 0; i < 3; ++i) eta[20 + i] = acadoVariables.od[3 * N + i];
        acado_integrate(eta, 1);
        for (int i = 0; i < 3; ++i) acadoVariables.x[3 * N + i] = eta[i];
    }
}

void acado_shiftControls(real_t* const uEnd)
{
    for (int i = 0; i < 2 * (N - 1); ++i) acadoVariables.u[i] = acadoVariables.u[i + 2];
    if (uEnd) {
        acadoVariables.u[2 * (N - 1)] = uEnd[0];
        acadoVariables.u[2 * (N - 1) + 1] = uEnd[1];
    }
}

real_t acado_getKKT(void) { return g_kkt; }
real_t acado_getObjective(void) { return g_obj; } // at the iterate returned by the last feedback step
int acado_getNWSR(void) { return g_nwsr; }

// one interval through the linearisation kernel: rk_eta = [x(3) | Gx(9) | Gu(6) | u(2) | od(3)]
int acado_integrate(real_t* const eta, int)
{
    if (!ensure()) return 2;
    static float *dx = nullptr, *du = nullptr, *dod = nullptr;
    if (!dx) {
        if (hipMalloc((void**)&dx, sizeof(float) * 3 * (N + 1)) != hipSuccess) return 2;
        if (hipMalloc((void**)&du, sizeof(float) * 2 * N) != hipSuccess) return 2;
        if (hipMalloc((void**)&dod, sizeof(float) * 3 * (N + 1)) != hipSuccess) return 2;
    }
    float hx[6] = {eta[0], eta[1], eta[2], 0, 0, 0}; // x_1 = 0: the defect is then the new state
    (void)hipMemcpyAsync(dx, hx, sizeof(hx), hipMemcpyHostToDevice, nullptr);
    (void)hipMemcpyAsync(du, eta + 18, sizeof(float) * 2, hipMemcpyHostToDevice, nullptr);
    (void)hipMemcpyAsync(dod, eta + 20, sizeof(float) * 3, hipMemcpyHostToDevice, nullptr);
    alore_nmpc_batch b{};
    b.x = dx; b.u = du; b.od = dod;
    alore_nmpc_lin_out lo{g_d, g_gx, g_gu};
    if (!ok(alore_nmpc_linearize(g_h, &b, 1, &lo, nullptr))) return 2;
    float d[3], gx[9], gu[6];
    (void)hipMemcpyAsync(d, g_d, sizeof(d), hipMemcpyDeviceToHost, nullptr);
    (void)hipMemcpyAsync(gx, g_gx, sizeof(gx), hipMemcpyDeviceToHost, nullptr);
    (void)hipMemcpyAsync(gu, g_gu, sizeof(gu), hipMemcpyDeviceToHost, nullptr);
    (void)hipStreamSynchronize(nullptr);
    for (int i = 0; i < 3; ++i) eta[i] = d[i];
    for (int i = 0; i < 9; ++i) eta[3 + i] = gx[i];
    for (int i = 0; i < 6; ++i) eta[12 + i] = gu[i];
    return 0;
}

// The ICR differential-drive model (host; the kernels carry their own copy in nmpc_kernels.hip):
//   v = (u_r yl - u_l yr) / (yl - yr),  w = (u_r - u_l) / (yl - yr),
//   xdot = v cos(th) + xv w sin(th),  ydot = v sin(th) - xv w cos(th),  thdot = w
void acado_rhs(const real_t* in, real_t* out)
{
    const real_t th = in[2], ur = in[3], ul = in[4], xv = in[5], yr = in[6], yl = in[7];
    const real_t span = yl - yr, c = (real_t)std::cos((double)th), s = (real_t)std::sin((double)th); // double libm, as the generated C
    const real_t v = (ur * yl - ul * yr) / span, lat = ((ur - ul) * xv) / span;
    out[0] = v * c + lat * s;
    out[1] = v * s - lat * c;
    out[2] = (ur - ul) / span;
}

void acado_diffs(const real_t* in, real_t* out)
{
    const real_t th = in[2], ur = in[3], ul = in[4], xv = in[5], yr = in[6], yl = in[7];
    const real_t span = yl - yr, inv = real_t(1) / span, c = (real_t)std::cos((double)th), s = (real_t)std::sin((double)th);
    const real_t v = (ur * yl - ul * yr) / span, lat = ((ur - ul) * xv) / span;
    for (int i = 0; i < 15; ++i) out[i] = real_t(0);
    out[2] = v * -s + lat * c;                       // d xdot / d theta
    out[3] = (yl * inv) * c + (xv * inv) * s;        // d xdot / d u_r
    out[4] = (-yr * inv) * c + (-xv * inv) * s;      // d xdot / d u_l
    out[7] = v * c - lat * -s;                       // d ydot / d theta
    out[8] = (yl * inv) * s - (xv * inv) * c;
    out[9] = (-yr * inv) * s - (-xv * inv) * c;
    out[13] = inv;
    out[14] = -inv;
}

// CG/acado_qpoases_interface.cpp:39-60: qpOASES' QProblemB on acadoWorkspace.H / g / lb / ub, primal into acadoWorkspace.x, dual
// into acadoWorkspace.y (whose previous content seeds the working set).  Here: the dense working-set solver of
// nmpc_dense.hip on the same arrays, as the caller left them.
int acado_solve(void)
{
    if (!ensure()) return 29;
    (void)hipMemcpyAsync(g_H, acadoWorkspace.H, sizeof(float) * ACADO_QP_NV * ACADO_QP_NV, hipMemcpyHostToDevice, nullptr);
    (void)hipMemcpyAsync(g_g, acadoWorkspace.g, sizeof(float) * ACADO_QP_NV, hipMemcpyHostToDevice, nullptr);
    (void)hipMemcpyAsync(g_lb, acadoWorkspace.lb, sizeof(float) * ACADO_QP_NV, hipMemcpyHostToDevice, nullptr);
    (void)hipMemcpyAsync(g_ub, acadoWorkspace.ub, sizeof(float) * ACADO_QP_NV, hipMemcpyHostToDevice, nullptr);
    (void)hipMemcpyAsync(g_qy, acadoWorkspace.y, sizeof(float) * ACADO_QP_NV, hipMemcpyHostToDevice, nullptr);
    alore_nmpc_dense_qp_data q{g_H, g_g, g_lb, g_ub};
    if (!ok(alore_nmpc_dense_qp(g_h, 1, ACADO_QP_NV, &q, g_qx, g_qy, g_qst, g_qst + 1, nullptr))) return 29;
    int st[2] = {29, 0};
    (void)hipMemcpyAsync(acadoWorkspace.x, g_qx, sizeof(float) * ACADO_QP_NV, hipMemcpyDeviceToHost, nullptr);
    (void)hipMemcpyAsync(acadoWorkspace.y, g_qy, sizeof(float) * ACADO_QP_NV, hipMemcpyDeviceToHost, nullptr);
    (void)hipMemcpyAsync(st, g_qst, sizeof(st), hipMemcpyDeviceToHost, nullptr);
    (void)hipStreamSynchronize(nullptr);
    g_nwsr = st[1];
    return st[0];
}

const char* acado_getErrorString(int error)
{
    switch (error) {
    case 0: return "Successful return";
    case 29: return "Initialisation failed (no GPU engine)";
    case 31: return "Initialisation failed: Hessian not positive definite";
    case 33: return "Initial QP could not be solved due to infeasibility";
    case 58: return "Maximum number of working set recalculations performed";
    default: return "Unknown error";
    }
}
}
