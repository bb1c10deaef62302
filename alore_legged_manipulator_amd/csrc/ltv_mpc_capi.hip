// ltv_mpc_capi.hip -- C ABI of the batched LTV-MPC (include/alore_ltv_mpc.h) and the kernels that sample its references
// from the trajectory store (getRefPoints / smooth_yaw of the `mpc` node, mpc_controller/src/mpc.cpp:634-690, 538-567).
// Built with default floating-point semantics: the argument checks and the heading-unwrap loops must see NaN / Inf as
// what they are (ltv_mpc.hip, the solver kernels, is built without them -- see ltv_mpc.h).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "ltv_mpc.h"
#include "minco_spline.h"
#include "nmpc_kernels.h"

namespace ltv {

// getRefPoints of the `mpc` node on the trajectory store: thread = (robot, i), then smooth_yaw per robot
__global__ void ltv_refs_kernel(nmpc::RefStore s, int B, int T, double dt, double now, double* xref, double* dref, int* at_goal)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)B * T) return;
    const int r = (int)(t / T), i = (int)(t % T);
    const double* m = s.meta + (size_t)r * 8;
    if (m[6] == 0.0) { if (i == 0 && at_goal) at_goal[r] = 0; return; }
    const double duration = m[1], xv = m[2], res = m[3];
    const int np = (int)m[4], nc = (int)m[5];
    const double* dur = s.dur + (size_t)r * s.P;
    const double* coef = s.coef + (size_t)r * s.P * 12;
    const double t_cur = now - m[0];
    double temp_t = t_cur + dt;
    for (int k = 0; k < i; ++k) temp_t += dt;
    const double tq = (temp_t <= duration) ? temp_t : duration;
    int index = (int)floor(tq / res);
    if (index > nc - 1) index = nc - 1;
    if (index < 0) index = 0;
    const double floor_t = index * res, diff_t = tq - floor_t;
    double p1[2], p2[2], p3[2], v1[2], v2[2], v3[2];
    constexpr int PRE = 16; // piece durations fetched up front (independent loads), as in nmpc::ref_sample_node
    double dreg[PRE];
#pragma unroll
    for (int k = 0; k < PRE; ++k) dreg[k] = dur[min(k, np - 1)];
    minco::eval_pv_pre<PRE>(dreg, dur, coef, np, floor_t, p1, v1);
    minco::eval_pv_pre<PRE>(dreg, dur, coef, np, floor_t + diff_t / 2.0, p2, v2);
    minco::eval_pv_pre<PRE>(dreg, dur, coef, np, tq, p3, v3);
    const double* ck = s.ckpt + ((size_t)r * s.C + index) * 2;
    double xd1, yd1, xd2, yd2, xd3, yd3; // one sincos per Simpson node
    minco::xydot(p1, v1, xv, xd1, yd1); minco::xydot(p2, v2, xv, xd2, yd2); minco::xydot(p3, v3, xv, xd3, yd3);
    const double X = ck[0] + diff_t / 6.0 * (xd1 + 4.0 * xd2 + xd3);
    const double Y = ck[1] + diff_t / 6.0 * (yd1 + 4.0 * yd2 + yd3);
    double psi = p3[0];
    if (std::isfinite(psi)) { // a non-finite sample is handed on as it is: get_cmd flags the robot (status 2)
        while (psi > M_PI) psi -= 2 * M_PI;
        while (psi < -M_PI) psi += 2 * M_PI;
    }
    double* xo = xref + ((size_t)r * T + i) * 3;
    xo[0] = X; xo[1] = Y; xo[2] = psi;
    dref[((size_t)r * T + i) * 2] = v3[1];
    dref[((size_t)r * T + i) * 2 + 1] = v3[0];
    if (i == 0 && at_goal) at_goal[r] = (t_cur > duration + 1.0) ? 1 : 0;
}
__global__ void ltv_unwrap_kernel(nmpc::RefStore s, int B, int T, const double* est, double* xref)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= B || s.meta[(size_t)r * 8 + 6] == 0.0) return;
    double* x = xref + (size_t)r * T * 3;
    const double th = est[(size_t)r * 3 + 2];
    if (!std::isfinite(th) || !std::isfinite(x[2])) return; // an Inf heading would never leave the loops below; get_cmd flags the robot (status 2)
    // the walk in registers (the headings were read, changed and read again in memory: several dependent round trips per node)
    double prev = x[2];
    double dy = prev - th;
    while (dy >= M_PI / 2) { prev -= 2 * M_PI; dy = prev - th; }
    while (dy <= -M_PI / 2) { prev += 2 * M_PI; dy = prev - th; }
    x[2] = prev;
    for (int i = 0; i + 1 < T; ++i) {
        double cur = x[3 * (i + 1) + 2];
        if (!std::isfinite(cur)) return;
        dy = cur - prev;
        while (dy >= M_PI / 2) { cur -= 2 * M_PI; dy = cur - prev; }
        while (dy <= -M_PI / 2) { cur += 2 * M_PI; dy = cur - prev; }
        x[3 * (i + 1) + 2] = cur;
        prev = cur;
    }
}

} // namespace ltv

// internal accessor of the NMPC handle's trajectory store (nmpc_capi.hip)
extern "C" int alore_nmpc_internal_refstore(void* nmpc_handle, nmpc::RefStore* out, int* capacity, int* device);

struct alore_ltv_solver {
    alore_ltv_config cfg;
    int device = 0, B = 0;
    std::string err;
    double *d_now = nullptr, *d_xref = nullptr, *d_dref = nullptr, *d_out = nullptr, *d_buff = nullptr, *d_xopt = nullptr, *d_ws = nullptr,
           *d_est = nullptr, *d_cmd = nullptr;
    int *d_st = nullptr, *d_sweeps = nullptr, *d_status = nullptr, *d_goal = nullptr;
    long long* d_stamps = nullptr;
    char* h_stage = nullptr; // pinned
    size_t stage_bytes = 0;
};

namespace {
int lfail(alore_ltv_handle h, int code, const char* what, hipError_t e = hipSuccess)
{
    if (h) { h->err = what; if (e != hipSuccess) { h->err += ": "; h->err += hipGetErrorString(e); } }
    return code;
}
#define LTV_TRY(h, call)                                                  \
    do {                                                                  \
        hipError_t e_ = (call);                                           \
        if (e_ != hipSuccess) return lfail(h, ALORE_LTV_E_HIP, #call, e_); \
    } while (0)
template <class T>
hipError_t zalloc(T** p, size_t n)
{
    hipError_t e = hipMalloc((void**)p, sizeof(T) * (n ? n : 1));
    if (e == hipSuccess) e = hipMemset(*p, 0, sizeof(T) * (n ? n : 1));
    return e;
}
void lfree(alore_ltv_handle h)
{
    void* ptrs[] = {h->d_now, h->d_xref, h->d_dref, h->d_out, h->d_buff, h->d_xopt, h->d_ws, h->d_est, h->d_st, h->d_sweeps, h->d_status, h->d_goal, h->d_cmd};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (h->h_stage) (void)hipHostFree(h->h_stage);
}
} // namespace

extern "C" {

void alore_ltv_default_config(alore_ltv_config* c)
{
    std::memset(c, 0, sizeof(*c));
    c->dt = 0.01; c->predict_steps = 30; c->delay_num = 1;
    c->matrix_q[0] = 15.0; c->matrix_q[1] = 15.0; c->matrix_q[2] = 0.0; c->matrix_q[3] = 1.0;
    c->matrix_r[0] = 0.0; c->matrix_r[1] = 0.0;
    c->matrix_rd[0] = 1.0; c->matrix_rd[1] = 0.05;
    c->max_vel = 3.0; c->min_vel = 0.0; c->max_omega = 3.0; c->max_acc = 2.0; c->max_domega = 4.0;
    c->max_sweeps = 64;
}

int alore_ltv_create(const alore_ltv_config* cfg, int device, int max_robots, alore_ltv_handle* out)
{
    if (!cfg || !out || max_robots < 1) return ALORE_LTV_E_INVALID;
    *out = nullptr;
    if (cfg->predict_steps < 2 || cfg->predict_steps > ltv::MAXT || cfg->delay_num < 0 || cfg->delay_num >= cfg->predict_steps - 1 ||
        !(cfg->dt > 0.0) || !std::isfinite(cfg->dt))
        return ALORE_LTV_E_INVALID;
    {
        const double vals[] = {cfg->matrix_q[0], cfg->matrix_q[1], cfg->matrix_q[2], cfg->matrix_q[3], cfg->matrix_r[0], cfg->matrix_r[1],
                               cfg->matrix_rd[0], cfg->matrix_rd[1], cfg->max_vel, cfg->min_vel, cfg->max_omega, cfg->max_acc, cfg->max_domega};
        for (double v : vals)
            if (!std::isfinite(v)) return ALORE_LTV_E_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ALORE_LTV_E_NO_DEVICE;
    if (device < 0 || device >= ndev) return ALORE_LTV_E_INVALID;
    if (hipSetDevice(device) != hipSuccess) return ALORE_LTV_E_NO_DEVICE;
    alore_ltv_solver* h = new (std::nothrow) alore_ltv_solver;
    if (!h) return ALORE_LTV_E_NOMEM;
    h->cfg = *cfg;
    if (h->cfg.max_sweeps <= 0) h->cfg.max_sweeps = 64;
    h->device = device;
    h->B = max_robots;
    const size_t B = max_robots, T = cfg->predict_steps, dl = cfg->delay_num > 0 ? cfg->delay_num : 1;
    hipError_t e = hipSuccess;
    auto A = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    A(zalloc(&h->d_now, B * 3)); A(zalloc(&h->d_xref, B * T * 3)); A(zalloc(&h->d_dref, B * T * 2)); A(zalloc(&h->d_out, B * T * 2));
    A(zalloc(&h->d_buff, B * dl * 2)); A(zalloc(&h->d_xopt, B * (T + 1) * 3)); A(zalloc(&h->d_ws, T * ltv::NF * B)); A(zalloc(&h->d_est, B * 3));
    if (std::getenv("ALORE_LTV_STAMPS")) A(zalloc(&h->d_stamps, (size_t)8));
    A(zalloc(&h->d_st, T * 2 * B)); A(zalloc(&h->d_sweeps, B)); A(zalloc(&h->d_status, B)); A(zalloc(&h->d_goal, B)); A(zalloc(&h->d_cmd, B * 2));
    h->stage_bytes = sizeof(double) * B * ((T + 1) * 3 + T * 5 + 8) + sizeof(int) * B * 4 + 1024;
    if (e == hipSuccess) e = hipHostMalloc((void**)&h->h_stage, h->stage_bytes, hipHostMallocDefault);
    if (e != hipSuccess) { lfree(h); delete h; return e == hipErrorOutOfMemory ? ALORE_LTV_E_NOMEM : ALORE_LTV_E_HIP; }
    *out = h;
    return ALORE_LTV_OK;
}

int alore_ltv_destroy(alore_ltv_handle h)
{
    if (!h) return ALORE_LTV_E_INVALID;
    (void)hipSetDevice(h->device);
    if (h->d_stamps) {
        long long st[8];
        if (hipMemcpy(st, h->d_stamps, sizeof(st), hipMemcpyDeviceToHost) == hipSuccess)
            std::fprintf(stderr, "[alore_ltv stamps] robot 0, cycles: rollout %lld, backward %lld, forward %lld, rest %lld\n", st[0], st[1], st[2], st[3]);
        (void)hipFree(h->d_stamps);
    }
    lfree(h);
    delete h;
    return ALORE_LTV_OK;
}
const char* alore_ltv_last_error(alore_ltv_handle h) { return h ? h->err.c_str() : "null handle"; }

int alore_ltv_set_refs(alore_ltv_handle h, int B, const double* xref, const double* dref, void* stream)
{
    if (!h || B < 1 || B > h->B || !xref || !dref) return lfail(h, ALORE_LTV_E_INVALID, "set_refs: bad argument");
    LTV_TRY(h, hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const size_t T = h->cfg.predict_steps;
    double* hx = (double*)h->h_stage;
    double* hd = hx + (size_t)B * T * 3;
    std::memcpy(hx, xref, sizeof(double) * B * T * 3);
    std::memcpy(hd, dref, sizeof(double) * B * T * 2);
    LTV_TRY(h, hipMemcpyAsync(h->d_xref, hx, sizeof(double) * B * T * 3, hipMemcpyHostToDevice, s));
    LTV_TRY(h, hipMemcpyAsync(h->d_dref, hd, sizeof(double) * B * T * 2, hipMemcpyHostToDevice, s));
    LTV_TRY(h, hipStreamSynchronize(s)); // the slab is reused
    return ALORE_LTV_OK;
}

int alore_ltv_refs_from_store(alore_ltv_handle h, void* nmpc, int B, double now, const double* est, int* at_goal, void* stream)
{
    if (!h || !nmpc || B < 1 || B > h->B || !est || !std::isfinite(now)) return lfail(h, ALORE_LTV_E_INVALID, "refs_from_store: bad argument");
    nmpc::RefStore rs;
    int cap = 0, dev = -1;
    if (alore_nmpc_internal_refstore(nmpc, &rs, &cap, &dev) != 0 || cap < B || dev != h->device)
        return lfail(h, ALORE_LTV_E_INVALID, "refs_from_store: the NMPC handle has no trajectory store for B robots on this device");
    LTV_TRY(h, hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const int T = h->cfg.predict_steps;
    double* he = (double*)h->h_stage;
    std::memcpy(he, est, sizeof(double) * B * 3);
    LTV_TRY(h, hipMemcpyAsync(h->d_est, he, sizeof(double) * B * 3, hipMemcpyHostToDevice, s));
    const long total = (long)B * T;
    hipLaunchKernelGGL(ltv::ltv_refs_kernel, dim3((unsigned)((total + 127) / 128)), dim3(128), 0, s, rs, B, T, h->cfg.dt, now, h->d_xref, h->d_dref,
                       h->d_goal);
    hipLaunchKernelGGL(ltv::ltv_unwrap_kernel, dim3((B + 63) / 64), dim3(64), 0, s, rs, B, T, h->d_est, h->d_xref);
    LTV_TRY(h, hipGetLastError());
    if (at_goal) LTV_TRY(h, hipMemcpyAsync(at_goal, h->d_goal, sizeof(int) * B, hipMemcpyDeviceToHost, s));
    LTV_TRY(h, hipStreamSynchronize(s));
    return ALORE_LTV_OK;
}

static int ltv_enqueue(alore_ltv_handle h, int B, const double* now_state, int n_relin, int reset, hipStream_t s,
                       double* cmd_host = nullptr, int* status_host = nullptr)
{
    double* hn = (double*)h->h_stage;
    std::memcpy(hn, now_state, sizeof(double) * B * 3);
    ltv::Dev d{};
    d.c = h->cfg; d.B = B; d.stride = h->B;
    static const char* which = std::getenv("ALORE_LTV_KERNEL");
    const bool thread_kernel = which && which[0] == 't';
    if (cmd_host && !thread_kernel) { // tick path: the lanes kernel reads the states once, straight from the pinned slab
        void* dn = nullptr;
        LTV_TRY(h, hipHostGetDevicePointer(&dn, hn, 0));
        d.now = (const double*)dn;
    } else {
        LTV_TRY(h, hipMemcpyAsync(h->d_now, hn, sizeof(double) * B * 3, hipMemcpyHostToDevice, s));
        d.now = h->d_now;
    }
    d.xref = h->d_xref; d.dref = h->d_dref; d.output = h->d_out; d.buff = h->d_buff; d.xopt = h->d_xopt;
    d.ws = h->d_ws; d.st = h->d_st; d.sweeps = h->d_sweeps; d.status = h->d_status; d.cmd = h->d_cmd;
    d.n_relin = n_relin; d.reset = reset;
    d.cmd_host = cmd_host; d.status_host = status_host;
    d.stamps = h->d_stamps;
    // 16 lanes per robot (stages in registers, sweeps lane by lane) unless ALORE_LTV_KERNEL=thread asks for the
    // one-thread-per-robot kernel (diagnostic A/B)
    LTV_TRY(h, ltv::launch_get_cmd(d, thread_kernel, s));
    return ALORE_LTV_OK;
}

int alore_ltv_get_cmd(alore_ltv_handle h, int B, const double* now_state, int n_relin, int reset, void* stream)
{
    if (!h || B < 1 || B > h->B || !now_state || n_relin < 1) return lfail(h, ALORE_LTV_E_INVALID, "get_cmd: bad argument");
    LTV_TRY(h, hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const int rc = ltv_enqueue(h, B, now_state, n_relin, reset, s);
    if (rc != ALORE_LTV_OK) return rc;
    LTV_TRY(h, hipStreamSynchronize(s)); // the staging slab is reused by the next call
    return ALORE_LTV_OK;
}

// one control tick: states in, getCmd, commands (and status) out -- one upload, one launch, one download, one wait
int alore_ltv_tick(alore_ltv_handle h, int B, const double* now_state, int n_relin, int reset, double* cmd, int* status, void* stream)
{
    if (!h || B < 1 || B > h->B || !now_state || n_relin < 1 || !cmd) return lfail(h, ALORE_LTV_E_INVALID, "tick: bad argument");
    LTV_TRY(h, hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    // the kernel writes the 20 bytes per robot a tick returns straight into the pinned slab (device alias of the host
    // pointer): two copy commands and their completion signals less on the critical path of the tick
    double* sc = (double*)h->h_stage + (size_t)B * 3;
    int* st = (int*)(sc + (size_t)B * 2);
    void *dsc = nullptr, *dst = nullptr;
    LTV_TRY(h, hipHostGetDevicePointer(&dsc, sc, 0));
    LTV_TRY(h, hipHostGetDevicePointer(&dst, st, 0));
    const int rc = ltv_enqueue(h, B, now_state, n_relin, reset, s, (double*)dsc, (int*)dst);
    if (rc != ALORE_LTV_OK) return rc;
    LTV_TRY(h, hipStreamSynchronize(s));
    std::memcpy(cmd, sc, sizeof(double) * B * 2);
    if (status) std::memcpy(status, st, sizeof(int) * B);
    return ALORE_LTV_OK;
}

int alore_ltv_results(alore_ltv_handle h, int B, double* output, double* xopt, int* sweeps, int* status, void* stream)
{
    if (!h || B < 1 || B > h->B) return lfail(h, ALORE_LTV_E_INVALID, "results: bad argument");
    LTV_TRY(h, hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const size_t T = h->cfg.predict_steps;
    // through the pinned slab of the handle (a device-to-pageable copy is staged by the runtime in small pieces with a host
    // wait per piece): layout  output | xopt | sweeps | status
    char* base = h->h_stage;
    double* so = (double*)base;
    double* sx = so + (size_t)B * T * 2;
    int* ss = (int*)(sx + (size_t)B * (T + 1) * 3);
    int* st = ss + B;
    if (output) LTV_TRY(h, hipMemcpyAsync(so, h->d_out, sizeof(double) * B * T * 2, hipMemcpyDeviceToHost, s));
    if (xopt) LTV_TRY(h, hipMemcpyAsync(sx, h->d_xopt, sizeof(double) * B * (T + 1) * 3, hipMemcpyDeviceToHost, s));
    if (sweeps) LTV_TRY(h, hipMemcpyAsync(ss, h->d_sweeps, sizeof(int) * B, hipMemcpyDeviceToHost, s));
    if (status) LTV_TRY(h, hipMemcpyAsync(st, h->d_status, sizeof(int) * B, hipMemcpyDeviceToHost, s));
    LTV_TRY(h, hipStreamSynchronize(s));
    if (output) std::memcpy(output, so, sizeof(double) * B * T * 2);
    if (xopt) std::memcpy(xopt, sx, sizeof(double) * B * (T + 1) * 3);
    if (sweeps) std::memcpy(sweeps, ss, sizeof(int) * B);
    if (status) std::memcpy(status, st, sizeof(int) * B);
    return ALORE_LTV_OK;
}

int alore_ltv_commands(alore_ltv_handle h, int B, double* cmd, int* status, void* stream)
{
    if (!h || B < 1 || B > h->B || !cmd) return lfail(h, ALORE_LTV_E_INVALID, "commands: bad argument");
    LTV_TRY(h, hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const size_t T = h->cfg.predict_steps, dl = h->cfg.delay_num;
    double* sc = (double*)h->h_stage;
    int* st = (int*)(sc + (size_t)B * 2);
    // column delay_num of every robot's output, packed by the kernel (16 bytes per robot)
    LTV_TRY(h, hipMemcpyAsync(sc, h->d_cmd, sizeof(double) * B * 2, hipMemcpyDeviceToHost, s));
    if (status) LTV_TRY(h, hipMemcpyAsync(st, h->d_status, sizeof(int) * B, hipMemcpyDeviceToHost, s));
    LTV_TRY(h, hipStreamSynchronize(s));
    std::memcpy(cmd, sc, sizeof(double) * B * 2);
    if (status) std::memcpy(status, st, sizeof(int) * B);
    return ALORE_LTV_OK;
}

int alore_ltv_set_state(alore_ltv_handle h, int B, const double* output, const double* buff, void* stream)
{
    if (!h || B < 1 || B > h->B) return lfail(h, ALORE_LTV_E_INVALID, "set_state: bad argument");
    LTV_TRY(h, hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const size_t T = h->cfg.predict_steps, dl = h->cfg.delay_num;
    if (output) LTV_TRY(h, hipMemcpyAsync(h->d_out, output, sizeof(double) * B * T * 2, hipMemcpyHostToDevice, s));
    if (buff && dl > 0) LTV_TRY(h, hipMemcpyAsync(h->d_buff, buff, sizeof(double) * B * dl * 2, hipMemcpyHostToDevice, s));
    LTV_TRY(h, hipStreamSynchronize(s));
    return ALORE_LTV_OK;
}

} // extern "C"
