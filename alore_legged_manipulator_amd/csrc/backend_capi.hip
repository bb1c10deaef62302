// backend_capi.hip -- implementation of include/alore_backend.h: argument checks, device storage, staging
// through pinned host memory, launches.  No CPU path: alore_backend_create fails without a GPU.
#include <algorithm>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "backend_kernels.h"

struct alore_backend_planner {
    alore_backend_config cfg;
    int device = 0, P = 0, B = 0, count = 0;
    std::string err;
    // map
    double* d_map = nullptr;
    backend::MapView map{};
    // problems
    int *d_M = nullptr, *d_cut = nullptr;
    double *d_inner = nullptr, *d_initT = nullptr, *d_pos = nullptr, *d_head = nullptr, *d_tail = nullptr, *d_sxy = nullptr,
           *d_fxy = nullptr, *d_sxyt = nullptr;
    // results
    double *r_inner = nullptr, *r_T = nullptr, *r_coef = nullptr, *r_tail = nullptr;
    int* r_ok = nullptr;
    backend::Status* r_status = nullptr;
    // workspace
    double *d_hist = nullptr, *d_gram = nullptr, *d_pcr = nullptr, *d_x = nullptr, *d_g = nullptr, *d_lam = nullptr, *d_rho = nullptr, *d_cost = nullptr, *d_err = nullptr;
    int* d_ret = nullptr;
    long long* d_stamps = nullptr; // ALORE_BE_STAMPS=1: diagnostic phase cycles of workgroup 0
    backend::Params* d_params = nullptr; // the kernel reads its parameter block from here
    int* d_order = nullptr;              // problems by piece count, longest first (launch order of alore_backend_plan)
    // pinned staging (one slab, carved per call)
    char* h_stage = nullptr;
    size_t stage_bytes = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
};

namespace {

int fail(alore_backend_handle h, int code, const char* what, hipError_t e = hipSuccess)
{
    if (h) {
        h->err = what;
        if (e != hipSuccess) { h->err += ": "; h->err += hipGetErrorString(e); }
    }
    return code;
}
#define BE_TRY(h, call)                                                  \
    do {                                                                 \
        hipError_t e_ = (call);                                          \
        if (e_ != hipSuccess) return fail(h, ALORE_BE_E_HIP, #call, e_); \
    } while (0)

template <class T>
hipError_t dalloc(T** p, size_t count)
{
    hipError_t e = hipMalloc((void**)p, sizeof(T) * (count ? count : 1));
    if (e == hipSuccess) e = hipMemset(*p, 0, sizeof(T) * (count ? count : 1));
    return e;
}

void free_all(alore_backend_handle h)
{
    if (h->d_stamps) {
        long long st[64];
        if (hipMemcpy(st, h->d_stamps, sizeof(st), hipMemcpyDeviceToHost) == hipSuccess) {
            std::fprintf(stderr, "[alore_backend stamps] workgroup 0, cycles per phase of eval_cost (slot 0 also holds the time outside it):");
            for (int i = 0; i < 44; ++i) std::fprintf(stderr, " %lld", st[i]);
            std::fprintf(stderr, "\n");
        }
        (void)hipFree(h->d_stamps);
    }
    void* ptrs[] = {h->d_map, h->d_M, h->d_cut, h->d_inner, h->d_initT, h->d_pos, h->d_head, h->d_tail, h->d_sxy, h->d_fxy, h->d_sxyt,
                    h->r_inner, h->r_T, h->r_coef, h->r_tail, h->r_ok, h->r_status, h->d_hist, h->d_gram, h->d_pcr, h->d_x, h->d_g, h->d_lam, h->d_rho,
                    h->d_cost, h->d_err, h->d_ret, h->d_params, h->d_order};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (h->h_stage) (void)hipHostFree(h->h_stage);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
}

backend::Params base_params(alore_backend_handle h, int count, int mode)
{
    backend::Params p{};
    p.cfg = h->cfg;
    p.map = h->map;
    p.prob = backend::ProblemStore{h->P, h->d_M, h->d_inner, h->d_initT, h->d_pos, h->d_head, h->d_tail, h->d_sxy, h->d_fxy, h->d_cut};
    p.res = backend::ResultStore{h->r_inner, h->r_T, h->r_coef, h->r_tail, h->r_ok, h->r_status};
    p.hist = h->d_hist;
    p.gram = h->d_gram;
    p.pcr = h->d_pcr;
    p.count = count;
    p.mode = mode;
    p.x_io = h->d_x;
    p.g_out = h->d_g;
    p.cost_out = h->d_cost;
    p.err_out = h->d_err;
    p.ret_out = h->d_ret;
    p.stamps = h->d_stamps;
    return p;
}

} // namespace

extern "C" {

void alore_backend_default_config(alore_backend_config* c)
{
    std::memset(c, 0, sizeof(*c));
    // plan_manager/config/car3ms.yaml
    c->max_vel = 3.0; c->min_vel = 0.0; c->max_acc = 2.0; c->max_omega = 3.0; c->max_domega = 4.0; c->max_cen_acc = 50.0;
    c->direct_v_omega = 0;
    c->n_check = 2;
    c->check_pts[0][0] = 0.3; c->check_pts[1][0] = -0.3;
    // back_end/config/global_planning3ms.yaml
    c->smooth_eps = 0.01;
    c->path_lbfgs = alore_lbfgs_param{256, 2, 8000, 64, 0.0, 5.0e-2, 0.0, 1.0e20, 1.0e-4, 0.9, 1.0e-6, 1.0e-16};
    c->shot_path_past = 8; c->shot_path_horizon = 0.5;
    c->p_time = 20; c->p_bigpath = 200000; c->p_mean_time = 100; c->p_moment = 1000; c->p_acc = 100; c->p_domega = 100;
    c->energy_w[0] = 0.33; c->energy_w[1] = 1.0;
    c->lbfgs = alore_lbfgs_param{256, 3, 8000, 64, 0.0, 5.0e-4, 1.0e-32, 1.0e20, 1.0e-4, 0.9, 1.0e-6, 1.0e-16};
    c->mean_lo = 0.5; c->mean_hi = 2.0;
    c->w_time = 50; c->w_acc = 300; c->w_domega = 300; c->w_collision = 500000; c->w_moment = 300; c->w_mean_time = 300; c->w_cen_acc = 300;
    for (int k = 0; k < 2; ++k) {
        c->lam0[k] = 0; c->rho0[k] = 1.0e4; c->rho_max[k] = 1.0e10; c->gamma[k] = 9.0;
        c->cut_lam0[k] = 0; c->cut_rho0[k] = 1.0e3; c->cut_rho_max[k] = 1.0e10; c->cut_gamma[k] = 5.0;
    }
    c->tol = 0.01; c->cut_tol = 0.5;
    c->sparse_res = 8;
    c->safe_dis = 0.6; c->final_min_safe_dis = 0.10; c->final_check_num = 16; c->safe_replan_max = 3;
    c->icr_xv = 0.2; c->standard_diff = 1; // planner_sim.launch:41-46
    c->max_alm_rounds = 64;
}

int alore_backend_create(const alore_backend_config* cfg, int device, int max_pieces, int max_problems, alore_backend_handle* out)
{
    if (!cfg || !out || max_pieces < 1 || max_problems < 1) return ALORE_BE_E_INVALID;
    *out = nullptr;
    if (cfg->sparse_res != 8 || cfg->n_check < 0 || cfg->n_check > 8 || cfg->lbfgs.mem_size < 1 || cfg->lbfgs.mem_size > backend::MEM_MAX ||
        cfg->path_lbfgs.mem_size < 1 || cfg->path_lbfgs.mem_size > backend::MEM_MAX || cfg->lbfgs.past > 16 || cfg->path_lbfgs.past > 16 ||
        cfg->shot_path_past > 16 || cfg->final_check_num < 1 || cfg->final_check_num > 64)
        return ALORE_BE_E_UNSUPPORTED;
    if (max_pieces > 32) return ALORE_BE_E_UNSUPPORTED;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ALORE_BE_E_NO_DEVICE;
    if (device < 0 || device >= ndev) return ALORE_BE_E_INVALID;
    if (hipSetDevice(device) != hipSuccess) return ALORE_BE_E_NO_DEVICE;
    alore_backend_planner* h = new (std::nothrow) alore_backend_planner;
    if (!h) return ALORE_BE_E_NOMEM;
    h->cfg = *cfg;
    h->device = device;
    h->P = max_pieces <= 16 ? 16 : 32; // two kernel instantiations: piece capacity 16 or 32
    h->B = max_problems;
    const size_t B = (size_t)max_problems, P = (size_t)h->P, ns = 3 * P;
    hipError_t e = hipSuccess;
    auto A = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    A(dalloc(&h->d_M, B)); A(dalloc(&h->d_cut, B)); A(dalloc(&h->d_inner, B * (P - 1) * 2)); A(dalloc(&h->d_initT, B));
    A(dalloc(&h->d_pos, B * P * 2)); A(dalloc(&h->d_head, B * 6)); A(dalloc(&h->d_tail, B * 6)); A(dalloc(&h->d_sxy, B * 2));
    A(dalloc(&h->d_fxy, B * 2)); A(dalloc(&h->d_sxyt, B * 3));
    A(dalloc(&h->r_inner, B * (P - 1) * 2)); A(dalloc(&h->r_T, B * P)); A(dalloc(&h->r_coef, B * P * 12)); A(dalloc(&h->r_tail, B * 6));
    A(dalloc(&h->r_ok, B)); A(dalloc(&h->r_status, B));
    A(dalloc(&h->d_hist, B * backend::MEM_MAX * 2 * ns));
    A(dalloc(&h->d_gram, B * (size_t)backend::GRAM_DOUBLES));
    A(dalloc(&h->d_pcr, B * (size_t)backend::WS_DOUBLES));
    if (std::getenv("ALORE_BE_STAMPS")) A(dalloc(&h->d_stamps, (size_t)64));
    A(dalloc(&h->d_x, B * ns)); A(dalloc(&h->d_g, B * ns)); A(dalloc(&h->d_lam, B * 2)); A(dalloc(&h->d_rho, B * 2));
    A(dalloc(&h->d_cost, B)); A(dalloc(&h->d_err, B * 2)); A(dalloc(&h->d_ret, B * 3)); A(dalloc(&h->d_params, 1)); A(dalloc(&h->d_order, B));
    // staging: the largest single transfer is the problem upload / the result download
    h->stage_bytes = B * (sizeof(int) * 2 + sizeof(double) * ((P - 1) * 2 + 1 + P * 2 + 6 + 6 + 2 + 2 + 3 + P * 12 + P + 2 * ns + 8) +
                          sizeof(backend::Status)) + 4096;
    if (e == hipSuccess) e = hipHostMalloc((void**)&h->h_stage, h->stage_bytes, hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreate(&h->ev0);
    if (e == hipSuccess) e = hipEventCreate(&h->ev1);
    if (e != hipSuccess) {
        free_all(h);
        delete h;
        return e == hipErrorOutOfMemory ? ALORE_BE_E_NOMEM : ALORE_BE_E_HIP;
    }
    *out = h;
    return ALORE_BE_OK;
}

int alore_backend_destroy(alore_backend_handle h)
{
    if (!h) return ALORE_BE_E_INVALID;
    (void)hipSetDevice(h->device);
    free_all(h);
    delete h;
    return ALORE_BE_OK;
}

const char* alore_backend_last_error(alore_backend_handle h) { return h ? h->err.c_str() : "null handle"; }

int alore_backend_set_map(alore_backend_handle h, const double* dist, int nx, int ny, double x_lo, double y_lo, double res)
{
    if (!h || !dist || nx < 2 || ny < 2 || !(res > 0.0)) return fail(h, ALORE_BE_E_INVALID, "set_map: bad argument");
    BE_TRY(h, hipSetDevice(h->device));
    if (h->d_map) { (void)hipFree(h->d_map); h->d_map = nullptr; }
    BE_TRY(h, hipMalloc((void**)&h->d_map, sizeof(double) * (size_t)nx * ny));
    BE_TRY(h, hipMemcpy(h->d_map, dist, sizeof(double) * (size_t)nx * ny, hipMemcpyHostToDevice));
    h->map = backend::MapView{h->d_map, nx, ny, x_lo, y_lo, x_lo + nx * res, y_lo + ny * res, res};
    return ALORE_BE_OK;
}

int alore_backend_build_esdf(alore_backend_handle h, const unsigned char* grid, int nx, int ny, double x_lo, double y_lo, double res,
                             double odom_x, double odom_y, double detection_range, double* dist_out)
{
    if (!h || !grid || nx < 2 || ny < 2 || !(res > 0.0) || !(detection_range > 0.0)) return fail(h, ALORE_BE_E_INVALID, "build_esdf: bad argument");
    BE_TRY(h, hipSetDevice(h->device));
    const size_t n = (size_t)nx * ny;
    const bool same = h->d_map && h->map.nx == nx && h->map.ny == ny && h->map.x_lo == x_lo && h->map.y_lo == y_lo && h->map.res == res;
    if (!same) {
        if (h->d_map) { (void)hipFree(h->d_map); h->d_map = nullptr; }
        BE_TRY(h, hipMalloc((void**)&h->d_map, sizeof(double) * n));
        BE_TRY(h, backend::esdf_fill_max(h->d_map, n, nullptr));
        h->map = backend::MapView{h->d_map, nx, ny, x_lo, y_lo, x_lo + nx * res, y_lo + ny * res, res};
    }
    unsigned char* d_grid = nullptr;
    BE_TRY(h, hipMalloc((void**)&d_grid, n));
    hipError_t e = hipMemcpy(d_grid, grid, n, hipMemcpyHostToDevice);
    int empty = 0;
    if (e == hipSuccess) e = backend::esdf_update(d_grid, nx, ny, res, x_lo, y_lo, odom_x, odom_y, detection_range, h->d_map, nullptr, &empty);
    (void)hipFree(d_grid);
    if (e != hipSuccess) return fail(h, ALORE_BE_E_HIP, "build_esdf", e);
    if (empty) return fail(h, ALORE_BE_E_INVALID, "build_esdf: the window around the odometry does not intersect the map");
    if (dist_out) BE_TRY(h, hipMemcpy(dist_out, h->d_map, sizeof(double) * n, hipMemcpyDeviceToHost));
    return ALORE_BE_OK;
}

int alore_backend_predicted_state(alore_backend_handle h, int count, double resolution, const double* start_time, const double* time,
                                  const double* start_xytheta, double* xytheta, double* vaj, double* oaj, int* forward)
{
    if (!h || count < 1 || count > h->B || !time || !(resolution > 0.0)) return fail(h, ALORE_BE_E_INVALID, "predicted_state: bad argument");
    if (!h->timed || count > h->count) return fail(h, ALORE_BE_E_INVALID, "predicted_state: no finished plan for these slots (alore_backend_plan first)");
    BE_TRY(h, hipSetDevice(h->device));
    const size_t n = count;
    double *d_in = nullptr, *d_out = nullptr;
    int* d_fwd = nullptr;
    BE_TRY(h, hipMalloc((void**)&d_in, sizeof(double) * n * 5));
    hipError_t e = hipMalloc((void**)&d_out, sizeof(double) * n * 9);
    if (e == hipSuccess) e = hipMalloc((void**)&d_fwd, sizeof(int) * n);
    if (e == hipSuccess) e = hipMemcpy(d_in, time, sizeof(double) * n, hipMemcpyHostToDevice);
    if (e == hipSuccess && start_time) e = hipMemcpy(d_in + n, start_time, sizeof(double) * n, hipMemcpyHostToDevice);
    if (e == hipSuccess && start_xytheta) e = hipMemcpy(d_in + 2 * n, start_xytheta, sizeof(double) * n * 3, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        backend::PredictArgs g{count, h->P, h->d_M, h->r_T, h->r_coef, h->d_sxyt, start_time ? d_in + n : nullptr, d_in,
                               start_xytheta ? d_in + 2 * n : nullptr, resolution, h->cfg.icr_xv, h->cfg.standard_diff != 0,
                               d_out, d_out + 3 * n, d_out + 6 * n, d_fwd};
        e = backend::predicted_state(g, nullptr);
    }
    if (e == hipSuccess && xytheta) e = hipMemcpy(xytheta, d_out, sizeof(double) * n * 3, hipMemcpyDeviceToHost);
    if (e == hipSuccess && vaj) e = hipMemcpy(vaj, d_out + 3 * n, sizeof(double) * n * 3, hipMemcpyDeviceToHost);
    if (e == hipSuccess && oaj) e = hipMemcpy(oaj, d_out + 6 * n, sizeof(double) * n * 3, hipMemcpyDeviceToHost);
    if (e == hipSuccess && forward) e = hipMemcpy(forward, d_fwd, sizeof(int) * n, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    (void)hipFree(d_in); (void)hipFree(d_out); (void)hipFree(d_fwd);
    if (e != hipSuccess) return fail(h, ALORE_BE_E_HIP, "predicted_state", e);
    return ALORE_BE_OK;
}

int alore_backend_path_points(alore_backend_handle h, int count, int panels_per_piece, double* xy, double* yaw, int* n_points)
{
    if (!h || count < 1 || count > h->B || panels_per_piece < 1 || !xy || !n_points)
        return fail(h, ALORE_BE_E_INVALID, "path_points: bad argument");
    if (!h->timed || count > h->count) return fail(h, ALORE_BE_E_INVALID, "path_points: no finished plan for these slots (alore_backend_plan first)");
    BE_TRY(h, hipSetDevice(h->device));
    const size_t n = count, per = (size_t)h->P * (panels_per_piece + 1);
    double *d_xy = nullptr, *d_yaw = nullptr;
    int* d_n = nullptr;
    hipError_t e = hipMalloc((void**)&d_xy, sizeof(double) * n * per * 2);
    if (e == hipSuccess) e = hipMemset(d_xy, 0, sizeof(double) * n * per * 2);
    if (e == hipSuccess && yaw) e = hipMalloc((void**)&d_yaw, sizeof(double) * n * h->P * panels_per_piece);
    if (e == hipSuccess && yaw) e = hipMemset(d_yaw, 0, sizeof(double) * n * h->P * panels_per_piece);
    if (e == hipSuccess) e = hipMalloc((void**)&d_n, sizeof(int) * n);
    if (e == hipSuccess) {
        backend::PathArgs g{count, h->P, panels_per_piece, h->d_M, h->r_ok, h->r_T, h->r_coef, h->d_sxyt, h->cfg.icr_xv, h->cfg.standard_diff != 0,
                            d_xy, d_yaw, d_n};
        e = backend::path_points(g, nullptr);
    }
    if (e == hipSuccess) e = hipMemcpy(xy, d_xy, sizeof(double) * n * per * 2, hipMemcpyDeviceToHost);
    if (e == hipSuccess && yaw) e = hipMemcpy(yaw, d_yaw, sizeof(double) * n * h->P * panels_per_piece, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(n_points, d_n, sizeof(int) * n, hipMemcpyDeviceToHost);
    (void)hipFree(d_xy); (void)hipFree(d_yaw); (void)hipFree(d_n);
    if (e != hipSuccess) return fail(h, ALORE_BE_E_HIP, "path_points", e);
    return ALORE_BE_OK;
}

int alore_backend_set_problems(alore_backend_handle h, int count, const alore_flat_traj* pr, void* stream)
{
    if (!h || count < 1 || count > h->B || !pr) return fail(h, ALORE_BE_E_INVALID, "set_problems: bad argument");
    BE_TRY(h, hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const size_t P = h->P, n = count;
    // carve the pinned slab
    char* base = h->h_stage;
    auto take = [&](size_t bytes) { char* p = base; base += (bytes + 15) & ~size_t(15); return p; };
    int* hM = (int*)take(sizeof(int) * n);
    int* hcut = (int*)take(sizeof(int) * n);
    double* hin = (double*)take(sizeof(double) * n * (P - 1) * 2);
    double* hT = (double*)take(sizeof(double) * n);
    double* hpos = (double*)take(sizeof(double) * n * P * 2);
    double* hhead = (double*)take(sizeof(double) * n * 6);
    double* htail = (double*)take(sizeof(double) * n * 6);
    double* hs = (double*)take(sizeof(double) * n * 2);
    double* hf = (double*)take(sizeof(double) * n * 2);
    double* hst = (double*)take(sizeof(double) * n * 3);
    std::memset(h->h_stage, 0, (size_t)(base - h->h_stage));
    for (int b = 0; b < count; ++b) {
        const alore_flat_traj& f = pr[b];
        const int M = f.n_pieces;
        if (M < 1) return fail(h, ALORE_BE_E_INVALID, "set_problems: a problem has no pieces");
        if (M > h->P) return fail(h, ALORE_BE_E_UNSUPPORTED, "set_problems: more pieces than the handle was created for");
        if (M > 1 && (!f.traj_pts || !f.positions)) return fail(h, ALORE_BE_E_INVALID, "set_problems: null array");
        hM[b] = M;
        hcut[b] = f.if_cut;
        hT[b] = f.init_T;
        for (int i = 0; i < M - 1; ++i) {
            hin[(size_t)b * (P - 1) * 2 + 2 * i] = f.traj_pts[3 * i];
            hin[(size_t)b * (P - 1) * 2 + 2 * i + 1] = f.traj_pts[3 * i + 1];
            hpos[(size_t)b * P * 2 + 2 * i] = f.positions[3 * i];
            hpos[(size_t)b * P * 2 + 2 * i + 1] = f.positions[3 * i + 1];
        }
        hpos[(size_t)b * P * 2 + 2 * (M - 1)] = f.final_xytheta[0];
        hpos[(size_t)b * P * 2 + 2 * (M - 1) + 1] = f.final_xytheta[1];
        for (int d = 0; d < 2; ++d)
            for (int k = 0; k < 3; ++k) { hhead[(size_t)b * 6 + d * 3 + k] = f.start_state[d][k]; htail[(size_t)b * 6 + d * 3 + k] = f.final_state[d][k]; }
        for (int k = 0; k < 2; ++k) { hs[(size_t)b * 2 + k] = f.start_xytheta[k]; hf[(size_t)b * 2 + k] = f.final_xytheta[k]; }
        for (int k = 0; k < 3; ++k) hst[(size_t)b * 3 + k] = f.start_xytheta[k];
    }
    {   // launch order: most pieces first.  The number of cost evaluations grows with the number of pieces (correlation 0.5
        // on the Monte-Carlo goals) and a launch ends with its slowest wavefront: the long problems must not start last
        // (measured on 2048 problems: 45 -> 37 ms; with the evaluation counts known in advance 35 ms).
        std::vector<int> ord(n);
        for (size_t b = 0; b < n; ++b) ord[b] = (int)b;
        std::stable_sort(ord.begin(), ord.end(), [&](int a, int b2) { return hM[a] > hM[b2]; });
        BE_TRY(h, hipMemcpy(h->d_order, ord.data(), sizeof(int) * n, hipMemcpyHostToDevice));
    }
    BE_TRY(h, hipMemcpyAsync(h->d_M, hM, sizeof(int) * n, hipMemcpyHostToDevice, s));
    BE_TRY(h, hipMemcpyAsync(h->d_cut, hcut, sizeof(int) * n, hipMemcpyHostToDevice, s));
    BE_TRY(h, hipMemcpyAsync(h->d_inner, hin, sizeof(double) * n * (P - 1) * 2, hipMemcpyHostToDevice, s));
    BE_TRY(h, hipMemcpyAsync(h->d_initT, hT, sizeof(double) * n, hipMemcpyHostToDevice, s));
    BE_TRY(h, hipMemcpyAsync(h->d_pos, hpos, sizeof(double) * n * P * 2, hipMemcpyHostToDevice, s));
    BE_TRY(h, hipMemcpyAsync(h->d_head, hhead, sizeof(double) * n * 6, hipMemcpyHostToDevice, s));
    BE_TRY(h, hipMemcpyAsync(h->d_tail, htail, sizeof(double) * n * 6, hipMemcpyHostToDevice, s));
    BE_TRY(h, hipMemcpyAsync(h->d_sxy, hs, sizeof(double) * n * 2, hipMemcpyHostToDevice, s));
    BE_TRY(h, hipMemcpyAsync(h->d_fxy, hf, sizeof(double) * n * 2, hipMemcpyHostToDevice, s));
    BE_TRY(h, hipMemcpyAsync(h->d_sxyt, hst, sizeof(double) * n * 3, hipMemcpyHostToDevice, s));
    BE_TRY(h, hipStreamSynchronize(s)); // the slab is reused by the next call
    h->count = count;
    return ALORE_BE_OK;
}

int alore_backend_plan(alore_backend_handle h, int count, void* stream)
{
    if (!h || count < 1 || count > h->count) return fail(h, ALORE_BE_E_INVALID, "plan: upload the problems first");
    if (!h->d_map) return fail(h, ALORE_BE_E_INVALID, "plan: no map (alore_backend_set_map)");
    BE_TRY(h, hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    backend::Params p = base_params(h, count, backend::MODE_PLAN);
    if (count == h->count) p.order = h->d_order; // the whole uploaded set: longest problems first
    BE_TRY(h, hipEventRecord(h->ev0, s));
    BE_TRY(h, backend::launch(p, h->d_params, h->P, s));
    BE_TRY(h, hipEventRecord(h->ev1, s));
    h->timed = true;
    return ALORE_BE_OK;
}

int alore_backend_last_plan_ms(alore_backend_handle h, float* ms)
{
    if (!h || !ms || !h->timed) return fail(h, ALORE_BE_E_INVALID, "last_plan_ms: no plan yet");
    BE_TRY(h, hipEventSynchronize(h->ev1));
    BE_TRY(h, hipEventElapsedTime(ms, h->ev0, h->ev1));
    return ALORE_BE_OK;
}

int alore_backend_results(alore_backend_handle h, int count, alore_backend_status* status, double* inner, double* T, double* coef,
                          void* stream)
{
    if (!h || count < 1 || count > h->B) return fail(h, ALORE_BE_E_INVALID, "results: bad argument");
    if (!h->timed || count > h->count) return fail(h, ALORE_BE_E_INVALID, "results: no plan has run for these slots (alore_backend_plan first)");
    BE_TRY(h, hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const size_t P = h->P, n = count;
    char* base = h->h_stage;
    auto take = [&](size_t bytes) { char* p = base; base += (bytes + 15) & ~size_t(15); return p; };
    backend::Status* hs = (backend::Status*)take(sizeof(backend::Status) * n);
    double* hin = (double*)take(sizeof(double) * n * (P - 1) * 2);
    double* hT = (double*)take(sizeof(double) * n * P);
    double* hc = (double*)take(sizeof(double) * n * P * 12);
    if (status) BE_TRY(h, hipMemcpyAsync(hs, h->r_status, sizeof(backend::Status) * n, hipMemcpyDeviceToHost, s));
    if (inner) BE_TRY(h, hipMemcpyAsync(hin, h->r_inner, sizeof(double) * n * (P - 1) * 2, hipMemcpyDeviceToHost, s));
    if (T) BE_TRY(h, hipMemcpyAsync(hT, h->r_T, sizeof(double) * n * P, hipMemcpyDeviceToHost, s));
    if (coef) BE_TRY(h, hipMemcpyAsync(hc, h->r_coef, sizeof(double) * n * P * 12, hipMemcpyDeviceToHost, s));
    BE_TRY(h, hipStreamSynchronize(s));
    if (status) std::memcpy(status, hs, sizeof(backend::Status) * n);
    if (inner) std::memcpy(inner, hin, sizeof(double) * n * (P - 1) * 2);
    if (T) std::memcpy(T, hT, sizeof(double) * n * P);
    if (coef) std::memcpy(coef, hc, sizeof(double) * n * P * 12);
    return ALORE_BE_OK;
}

int alore_backend_device_results(alore_backend_handle h, alore_backend_device_view* out)
{
    if (!h || !out) return ALORE_BE_E_INVALID;
    out->max_pieces = h->P;
    out->n_pieces = h->d_M;
    out->inner = h->r_inner;
    out->T = h->r_T;
    out->coef = h->r_coef;
    out->head = h->d_head;
    out->tail = h->r_tail;
    out->start_xytheta = h->d_sxyt;
    out->ok = h->r_ok;
    return ALORE_BE_OK;
}

static int run_piece(alore_backend_handle h, int count, int mode, int stage, double* x, const double* lam, const double* rho,
                     double safe_dis, double time_weight, int max_iter, double* cost, double* grad, int* ret, int* iters, int* evals,
                     double* xy_err, void* stream)
{
    if (!h || count < 1 || count > h->count || !x || (stage != 1 && stage != 2))
        return fail(h, ALORE_BE_E_INVALID, "eval/lbfgs: bad argument (upload the problems first)");
    if (!h->d_map) return fail(h, ALORE_BE_E_INVALID, "eval/lbfgs: no map");
    BE_TRY(h, hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const size_t ns = 3 * (size_t)h->P, n = count;
    BE_TRY(h, hipMemcpyAsync(h->d_x, x, sizeof(double) * n * ns, hipMemcpyHostToDevice, s));
    if (lam) BE_TRY(h, hipMemcpyAsync(h->d_lam, lam, sizeof(double) * n * 2, hipMemcpyHostToDevice, s));
    if (rho) BE_TRY(h, hipMemcpyAsync(h->d_rho, rho, sizeof(double) * n * 2, hipMemcpyHostToDevice, s));
    backend::Params p = base_params(h, count, mode);
    p.stage = stage;
    p.max_iter = max_iter;
    p.lam_in = lam ? h->d_lam : nullptr;
    p.rho_in = rho ? h->d_rho : nullptr;
    p.safe_dis = safe_dis;
    p.time_weight = time_weight;
    BE_TRY(h, backend::launch(p, h->d_params, h->P, s));
    std::vector<int> r3((size_t)n * 3);
    if (cost) BE_TRY(h, hipMemcpyAsync(cost, h->d_cost, sizeof(double) * n, hipMemcpyDeviceToHost, s));
    if (grad) BE_TRY(h, hipMemcpyAsync(grad, h->d_g, sizeof(double) * n * ns, hipMemcpyDeviceToHost, s));
    if (xy_err) BE_TRY(h, hipMemcpyAsync(xy_err, h->d_err, sizeof(double) * n * 2, hipMemcpyDeviceToHost, s));
    BE_TRY(h, hipMemcpyAsync(r3.data(), h->d_ret, sizeof(int) * n * 3, hipMemcpyDeviceToHost, s));
    if (mode == backend::MODE_LBFGS) BE_TRY(h, hipMemcpyAsync(x, h->d_x, sizeof(double) * n * ns, hipMemcpyDeviceToHost, s));
    BE_TRY(h, hipStreamSynchronize(s));
    for (size_t b = 0; b < n; ++b) {
        if (ret) ret[b] = r3[b * 3];
        if (iters) iters[b] = r3[b * 3 + 1];
        if (evals) evals[b] = r3[b * 3 + 2];
    }
    return ALORE_BE_OK;
}

int alore_backend_eval(alore_backend_handle h, int count, int stage, const double* x, const double* lam, const double* rho,
                       double safe_dis, double time_weight, double* cost, double* grad, double* xy_err, void* stream)
{
    return run_piece(h, count, backend::MODE_EVAL, stage, const_cast<double*>(x), lam, rho, safe_dis, time_weight, 0, cost, grad, nullptr,
                     nullptr, nullptr, xy_err, stream);
}

int alore_backend_lbfgs(alore_backend_handle h, int count, int stage, double* x, const double* lam, const double* rho, double safe_dis,
                        double time_weight, int max_iter, double* cost, int* ret, int* iters, int* evals, double* xy_err, void* stream)
{
    return run_piece(h, count, backend::MODE_LBFGS, stage, x, lam, rho, safe_dis, time_weight, max_iter, cost, nullptr, ret, iters, evals,
                     xy_err, stream);
}

} // extern "C"
