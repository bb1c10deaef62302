// esdf_build.hip -- SDFmap::updateESDF2d (P/utils/plan_env/src/sdf_map.cpp:618-681) on the GPU: the Euclidean signed
// distance field the back_end queries, built from the occupancy grid inside the window odom +- detection_range.
//
// The reference runs Felzenszwalb's lower-envelope distance transform (fillESDF, :683-714) along y for every row, then
// along x for every column, once for the occupied cells (distance outside obstacles) and once for the rest (distance
// inside), and combines them.  One line is a sequential envelope walk; lines are independent: ONE THREAD PER LINE,
// consecutive threads = consecutive columns in the column passes (coalesced), envelope stacks in a global workspace.
// The reference indexes its (X+1) x (Y+1) scratch with stride Y, so the last element of a row is the first of the next
// and the last column's results land on the next row's first column; executed sequentially that is well defined, and
// the same final contents are produced here by letting exactly the thread that would have written last do the write
// (see `keep` below).  All values are sums of squared integers until the final sqrt: results are bit-identical to the
// reference's arithmetic (checked against oracle/backend_oracle.c: be_update_esdf2d, which matches SciPy's exact EDT).
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>

#include "backend_kernels.h"

namespace backend {

namespace {

struct EsdfArgs {
    const unsigned char* grid; // [GLX][GLY] 0 unknown, 1 unoccupied, 2 occupied (sdf_map.h:98)
    int GLY, min_x, min_y, X, Y;
    double res;
    double* tmp;   // (X+1)(Y+1), stride Y
    double* out;   // pos or neg, same indexing
    int* v;        // [lines][D + 2]
    double* z;     // [lines][D + 3]
    int D;         // max(X, Y) + 1
    int negative;  // 0: seeds = occupied cells; 1: seeds = everything else
};

// fillESDF for line `line`; get(q) reads the input of the line, put(q, val) receives (q - v)^2 + get(v)
template <class Get, class Put>
__device__ __forceinline__ void fill_line(int end, int* v, double* z, Get get, Put put)
{
    int k = 0;
    v[0] = 0;
    z[0] = -DBL_MAX;
    z[1] = DBL_MAX;
    for (int q = 1; q <= end; q++) {
        double s;
        k++;
        const double fq = get(q) + q * q;
        do {
            k--;
            const int vk = v[k];
            s = (fq - (get(vk) + vk * vk)) / (2 * q - 2 * vk);
        } while (s <= z[k]);
        k++;
        v[k] = q;
        z[k] = s;
        z[k + 1] = DBL_MAX;
    }
    k = 0;
    for (int q = 0; q <= end; q++) {
        while (z[k + 1] < q) k++;
        const int vk = v[k];
        put(q, (double)((q - vk) * (q - vk)) + get(vk));
    }
}

__global__ void esdf_rows(EsdfArgs a)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x > a.X) return;
    const unsigned char* g = a.grid + (size_t)(x + a.min_x) * a.GLY + a.min_y;
    const int neg = a.negative;
    auto get = [&](int y) -> double {
        const int st = g[y];
        const bool seed = neg ? (st == 1 || st == 0) : (st == 2);
        return seed ? 0.0 : DBL_MAX;
    };
    double* row = a.tmp + (size_t)x * a.Y;
    const bool last_row = x == a.X;
    const int Y = a.Y;
    // element y = Y of row x is element 0 of row x + 1, which row x + 1 (running later in the reference) overwrites
    auto put = [&](int y, double val) { if (y < Y || last_row) row[y] = val; };
    fill_line(a.Y, a.v + (size_t)x * (a.D + 2), a.z + (size_t)x * (a.D + 3), get, put);
}

__global__ void esdf_cols(EsdfArgs a)
{
    const int y = blockIdx.x * blockDim.x + threadIdx.x;
    if (y > a.Y) return;
    const double* in = a.tmp + y;
    double* out = a.out + y;
    const int Y = a.Y;
    const double res = a.res;
    auto get = [&](int x) -> double { return in[(size_t)x * Y]; };
    // column Y (processed last by the reference) writes over rows 1.. of column 0: column 0 keeps only its first entry
    auto put = [&](int x, double val) { if (y != 0 || x == 0) out[(size_t)x * Y] = res * sqrt(val); };
    fill_line(a.X, a.v + (size_t)y * (a.D + 2), a.z + (size_t)y * (a.D + 3), get, put);
}

__global__ void esdf_combine(const double* pos, const double* neg, double* dist_all, int GLY, int min_x, int min_y, int X, int Y, double res)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)X * Y) return;
    const int x = (int)(t / Y), y = (int)(t % Y);
    const size_t i = (size_t)x * Y + y;
    double d = pos[i];
    if (neg[i] > 0.0) d += (-neg[i] + res);
    dist_all[(size_t)(x + min_x) * GLY + y + min_y] = d;
}

__global__ void fill_max(double* p, size_t n)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) p[t] = DBL_MAX;
}

} // namespace

hipError_t esdf_fill_max(double* p, size_t n, hipStream_t s)
{
    fill_max<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(p, n);
    return hipGetLastError();
}

// d_grid: device copy of the state grid; d_dist: the map the optimiser reads (updated in the window)
hipError_t esdf_update(const unsigned char* d_grid, int GLX, int GLY, double res, double x_lo, double y_lo, double odom_x, double odom_y,
                       double range, double* d_dist, hipStream_t s, int* empty_window)
{
    const double inv = 1.0 / res, gx = GLX * res, gy = GLY * res;
    const int min_x = (int)floor(fmax(0.0, odom_x - range - x_lo) * inv), min_y = (int)floor(fmax(0.0, odom_y - range - y_lo) * inv);
    const int max_x = (int)ceil(fmin(gx, odom_x + range - x_lo) * inv) - 1, max_y = (int)ceil(fmin(gy, odom_y + range - y_lo) * inv) - 1;
    const int X = max_x - min_x, Y = max_y - min_y;
    *empty_window = (X < 1 || Y < 1) ? 1 : 0;
    if (*empty_window) return hipSuccess;
    const size_t total = (size_t)(X + 1) * (Y + 1);
    const int D = (X > Y ? X : Y) + 1, lines = D;
    double *tmp = nullptr, *pos = nullptr, *neg = nullptr, *z = nullptr;
    int* v = nullptr;
    hipError_t e = hipMalloc((void**)&tmp, sizeof(double) * total);
    if (e == hipSuccess) e = hipMalloc((void**)&pos, sizeof(double) * total);
    if (e == hipSuccess) e = hipMalloc((void**)&neg, sizeof(double) * total);
    if (e == hipSuccess) e = hipMalloc((void**)&v, sizeof(int) * (size_t)lines * (D + 2));
    if (e == hipSuccess) e = hipMalloc((void**)&z, sizeof(double) * (size_t)lines * (D + 3));
    if (e == hipSuccess) e = hipMemsetAsync(pos, 0, sizeof(double) * total, s);
    if (e == hipSuccess) e = hipMemsetAsync(neg, 0, sizeof(double) * total, s);
    if (e == hipSuccess) {
        for (int pass = 0; pass < 2; ++pass) {
            EsdfArgs a{d_grid, GLY, min_x, min_y, X, Y, res, tmp, pass == 0 ? pos : neg, v, z, D, pass};
            esdf_rows<<<(X + 64) / 64, 64, 0, s>>>(a);
            esdf_cols<<<(Y + 64) / 64, 64, 0, s>>>(a);
        }
        esdf_combine<<<(unsigned)(((size_t)X * Y + 255) / 256), 256, 0, s>>>(pos, neg, d_dist, GLY, min_x, min_y, X, Y, res);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    (void)hipFree(tmp); (void)hipFree(pos); (void)hipFree(neg); (void)hipFree(v); (void)hipFree(z);
    return e;
}

// ---- MSPlanner::get_the_predicted_state / get_the_predicted_state_and_path (optimizer.cpp:1108-1262) -----------------
// One thread per plan walks the Simpson steps in the reference's order (the sum is sequential by definition).
namespace {
__device__ __forceinline__ void plan_eval(const double* T, const double* coef, int M, double t, double* p, double* v, double* a, double* j)
{
    int idx;
    double d = 0.0;
    for (idx = 0; idx < M && t > (d = T[idx]); ++idx) t -= d; // Trajectory::locatePieceIdx (trajectory.hpp:472-490)
    if (idx == M) { --idx; t += T[idx]; }
#pragma unroll
    for (int dm = 0; dm < 2; ++dm) {
        const double* c = coef + 12 * idx + dm;
        const double c0 = c[0], c1 = c[2], c2 = c[4], c3 = c[6], c4 = c[8], c5 = c[10];
        if (p) p[dm] = ((((c5 * t + c4) * t + c3) * t + c2) * t + c1) * t + c0;
        if (v) v[dm] = (((5.0 * c5 * t + 4.0 * c4) * t + 3.0 * c3) * t + 2.0 * c2) * t + c1;
        if (a) a[dm] = ((20.0 * c5 * t + 12.0 * c4) * t + 6.0 * c3) * t + 2.0 * c2;
        if (j) j[dm] = (60.0 * c5 * t + 24.0 * c4) * t + 6.0 * c3;
    }
}
__device__ __forceinline__ void simpson_add(bool standard, double xv, double w6, const double* p1, const double* v1, const double* p2,
                                            const double* v2, const double* p3, const double* v3, double* xyt)
{
    double s1, c1, s2, c2, s3, c3;
    sincos(p1[0], &s1, &c1); sincos(p2[0], &s2, &c2); sincos(p3[0], &s3, &c3);
    if (standard) {
        xyt[0] += w6 * (v1[1] * c1 + 4.0 * v2[1] * c2 + v3[1] * c3);
        xyt[1] += w6 * (v1[1] * s1 + 4.0 * v2[1] * s2 + v3[1] * s3);
    } else {
        const double x1 = v1[1] * c1 + v1[0] * xv * s1, x2 = v2[1] * c2 + v2[0] * xv * s2, x3 = v3[1] * c3 + v3[0] * xv * s3;
        const double y1 = v1[1] * s1 - v1[0] * xv * c1, y2 = v2[1] * s2 - v2[0] * xv * c2, y3 = v3[1] * s3 - v3[0] * xv * c3;
        xyt[0] += w6 * (x1 + 4.0 * x2 + x3);
        xyt[1] += w6 * (y1 + 4.0 * y2 + y3);
    }
    xyt[2] = p3[0];
}
__global__ void predicted_state_kernel(PredictArgs g)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= g.count) return;
    const int M = g.n_pieces[b];
    if (M < 1) { // a slot without a plan: no state to predict (plan_eval would index piece -1)
        for (int k = 0; k < 3; ++k) {
            if (g.xyt_out) g.xyt_out[(size_t)b * 3 + k] = 0.0;
            if (g.vaj_out) g.vaj_out[(size_t)b * 3 + k] = 0.0;
            if (g.oaj_out) g.oaj_out[(size_t)b * 3 + k] = 0.0;
        }
        if (g.forward_out) g.forward_out[b] = 0;
        return;
    }
    const double* T = g.T + (size_t)b * g.P;
    const double* coef = g.coef + (size_t)b * g.P * 12;
    const double start_time = g.start_time ? g.start_time[b] : 0.0, time = g.time[b], step = g.step;
    double xyt[3];
    for (int k = 0; k < 3; ++k) xyt[k] = g.start_xyt ? g.start_xyt[(size_t)b * 3 + k] : g.plan_start_xyt[(size_t)b * 3 + k];
    double total = 0.0;
    for (int i = 0; i < M; ++i) total += T[i];
    const double check = time > total ? total : time;
    const int n = (int)floor((check - start_time) / step);
    const double left = check - n * step - start_time;
    double p1[2], v1[2], p2[2], v2[2], p3[2], v3[2], a3[2], j3[2];
    plan_eval(T, coef, M, start_time, p3, v3, nullptr, nullptr);
    for (int i = 0; i < n; ++i) {
        p1[0] = p3[0]; p1[1] = p3[1]; v1[0] = v3[0]; v1[1] = v3[1];
        plan_eval(T, coef, M, start_time + i * step + step / 2.0, p2, v2, nullptr, nullptr);
        plan_eval(T, coef, M, start_time + i * step + step, p3, v3, nullptr, nullptr);
        simpson_add(g.standard_diff, g.xv, step / 6.0, p1, v1, p2, v2, p3, v3, xyt);
    }
    p1[0] = p3[0]; p1[1] = p3[1]; v1[0] = v3[0]; v1[1] = v3[1];
    plan_eval(T, coef, M, check - left / 2.0, p2, v2, nullptr, nullptr);
    plan_eval(T, coef, M, check, p3, v3, a3, j3);
    simpson_add(g.standard_diff, g.xv, left / 6.0, p1, v1, p2, v2, p3, v3, xyt);
    for (int k = 0; k < 3; ++k) g.xyt_out[(size_t)b * 3 + k] = xyt[k];
    g.oaj_out[(size_t)b * 3] = v3[0]; g.oaj_out[(size_t)b * 3 + 1] = a3[0]; g.oaj_out[(size_t)b * 3 + 2] = j3[0];
    g.vaj_out[(size_t)b * 3] = v3[1]; g.vaj_out[(size_t)b * 3 + 1] = a3[1]; g.vaj_out[(size_t)b * 3 + 2] = j3[1];
    double pe[2], ps[2];
    plan_eval(T, coef, M, time, pe, nullptr, nullptr, nullptr);
    plan_eval(T, coef, M, start_time, ps, nullptr, nullptr, nullptr);
    g.forward_out[b] = pe[1] - ps[1] > 0.0 ? 1 : 0;
}
} // namespace

// ---- MSPlanner::mincoPointPub (optimizer.cpp:1714-1826): the marker points of every plan -------------------------------
// One thread per plan: per piece `res` Simpson panels of the planar velocity in the reference's order of additions (start
// term, 4 x mid term, end term), accumulated from the plan's start position; the last point of a piece is written twice.
namespace {
__global__ void path_points_kernel(PathArgs g)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= g.count) return;
    const int M = g.n_pieces[b], res = g.res;
    double* xy = g.xy_out + (size_t)b * g.P * (res + 1) * 2;
    double* yaw = g.yaw_out ? g.yaw_out + (size_t)b * g.P * res : nullptr;
    if (M < 1 || (g.ok && !g.ok[b])) { g.n_out[b] = 0; return; } // a slot without a plan has no path
    const double* T = g.T + (size_t)b * g.P;
    const double* coef = g.coef + (size_t)b * g.P * 12;
    double px = g.plan_start_xyt[(size_t)b * 3], py = g.plan_start_xyt[(size_t)b * 3 + 1], sumT = 0.0;
    int n = 0;
    for (int i = 0; i < M; ++i) {
        const double step = T[i] / res, halfstep = step / 2.0, C = T[i] / res / 6.0;
        double s1 = 0.0, ax = 0.0, ay = 0.0; // the panel being filled
        for (int j = 0; j <= 2 * res; ++j) {
            double p[2], v[2], sy, cy, fx, fy;
            plan_eval(T, coef, M, s1 + sumT, p, v, nullptr, nullptr);
            s1 += halfstep;
            sincos(p[0], &sy, &cy);
            if ((j & 1) == 0) {
                if (g.standard_diff) { fx = C * v[1] * cy; fy = C * v[1] * sy; }
                else { fx = C * (v[1] * cy + v[0] * g.xv * sy); fy = C * (v[1] * sy - v[0] * g.xv * cy); }
                if (j != 0) { // end term of panel j / 2 - 1: the panel is complete
                    ax += fx; ay += fy;
                    px += ax; py += ay;
                    xy[2 * n] = px; xy[2 * n + 1] = py; ++n;
                    if (yaw) yaw[i * res + j / 2 - 1] = p[0];
                    if (j == 2 * res) { xy[2 * n] = px; xy[2 * n + 1] = py; ++n; }
                }
                ax = fx; ay = fy; // start term of panel j / 2 (0 + fx in the reference)
            } else {
                if (g.standard_diff) { fx = 4.0 * C * v[1] * cy; fy = 4.0 * C * v[1] * sy; }
                else { fx = 4 * C * (v[1] * cy + v[0] * g.xv * sy); fy = 4 * C * (v[1] * sy - v[0] * g.xv * cy); }
                ax += fx; ay += fy;
            }
        }
        sumT += T[i];
    }
    g.n_out[b] = n;
}
} // namespace

hipError_t path_points(const PathArgs& g, hipStream_t s)
{
    path_points_kernel<<<(g.count + 63) / 64, 64, 0, s>>>(g);
    return hipGetLastError();
}

hipError_t predicted_state(const PredictArgs& g, hipStream_t s)
{
    predicted_state_kernel<<<(g.count + 63) / 64, 64, 0, s>>>(g);
    return hipGetLastError();
}

} // namespace backend
