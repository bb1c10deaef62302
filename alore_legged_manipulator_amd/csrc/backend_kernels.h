// backend_kernels.h -- launch interface between backend_capi.hip and backend_kernels.hip
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/alore_backend.h"

namespace backend {

using Config = alore_backend_config;
using LbfgsParam = alore_lbfgs_param;
using Status = alore_backend_status;

constexpr int MEM_MAX = 256; // L-BFGS history slots per problem (global_planning3ms.yaml: mem_size 256)
// per problem, next to the history: for every aligned chunk of 8 slots the 8 x 8 block G[k][l] = s_k . y_l (k < l; the rest stays
// zero) row by row, its transpose, then y's and 1 / y's per slot (backend_kernels.hip: two_loop_gram)
constexpr int PCR_DOUBLES = 5 * 32 * 8 + 32 * 4; // per problem: [step][knot][alpha, gamma (2 x 2 each)], then [knot][D^-1]
// ... followed in the same per-problem workspace by two node arrays of an evaluation that are written and read once or twice and so need
// no LDS: the order-2 node terms of the even nodes [32 * 9][2] and the duration-gradient terms per node [32 * 17]
constexpr int WS_EB = PCR_DOUBLES, WS_NODET = WS_EB + 32 * 9 * 2, WS_CS = WS_NODET + 32 * 17, WS_DOUBLES = WS_CS + 2 * 32 * 17; // ... and cos / sin of the heading per node [node][2]
constexpr int GRAM_G = 0, GRAM_GT = (MEM_MAX / 8) * 64, GRAM_YS = 2 * GRAM_GT, GRAM_RYS = GRAM_YS + MEM_MAX, GRAM_DOUBLES = GRAM_RYS + MEM_MAX;

enum Mode { MODE_PLAN = 0, MODE_EVAL = 1, MODE_LBFGS = 2 };

struct MapView {
    const double* dist;
    int nx, ny;
    double x_lo, y_lo, x_hi, y_hi, res;
};

// device copy of the FlatTrajData batch; P = piece capacity
struct ProblemStore {
    int P;
    const int* M;            // [B]
    const double* inner;     // [B][(P-1)*2]  (yaw, s)
    const double* init_T;    // [B]
    const double* positions; // [B][P*2]      way-points (x, y) then final (x, y)
    const double* head;      // [B][6]        [d][p v a]
    const double* tail;      // [B][6]
    const double* start_xy;  // [B][2]
    const double* final_xy;  // [B][2]
    const int* if_cut;       // [B]
};

struct ResultStore {
    double* inner; // [B][(P-1)*2]
    double* T;     // [B][P]
    double* coef;  // [B][P*12]
    double* tail;  // [B][6]
    int* ok;       // [B]
    Status* status;
};

struct Params {
    Config cfg;
    MapView map;
    ProblemStore prob;
    ResultStore res;
    double* hist; // [B][MEM_MAX][2][3P]
    double* gram; // [B][GRAM_DOUBLES], zero-initialised once (the lower triangles are never written)
    double* pcr;  // [B][WS_DOUBLES] per-problem workspace of an evaluation: elimination factors of the knot system of the current evaluation (spline solve -> adjoint solve)
    int count, mode;
    // MODE_EVAL / MODE_LBFGS
    int stage, max_iter;
    double* x_io;         // [B][3P]
    double* g_out;        // [B][3P]
    const double* lam_in; // [B][2] or null
    const double* rho_in;
    double safe_dis, time_weight;
    double* cost_out; // [B]
    double* err_out;  // [B][2]
    int* ret_out;     // [B][3]: ret, iterations, evaluations
    const int* order;  // optional: workgroup w works on problem order[w] (longest problems first: a launch ends with its slowest wavefront)
    long long* stamps; // diagnostic (ALORE_BE_STAMPS=1): [64] cycles per phase of workgroup 0, [63] = last stamp
};

size_t lds_bytes(int P);
// the parameter block is copied (asynchronously, stream-ordered) into d_params and read from there by the kernel;
// `p` must stay alive until the copy has been issued from pinned memory or has completed (callers keep it in the handle)
hipError_t launch(const Params& p, const Params* d_params, int P, hipStream_t s);

// MSPlanner::get_the_predicted_state[_and_path] on the plans of the last launch (esdf_build.hip)
struct PredictArgs {
    int count, P;
    const int* n_pieces;
    const double *T, *coef;          // ResultStore
    const double* plan_start_xyt;    // [B][3] start pose of the plan (final_initStateXYTheta_)
    const double *start_time, *time; // [B] (start_time may be null = 0)
    const double* start_xyt;         // [B][3] or null (= plan start pose)
    double step, xv;
    bool standard_diff;
    double *xyt_out, *vaj_out, *oaj_out;
    int* forward_out;
};
hipError_t predicted_state(const PredictArgs& g, hipStream_t s);
// MSPlanner::mincoPointPub on the plans of the last launch (esdf_build.hip)
struct PathArgs {
    int count, P, res;
    const int* n_pieces;
    const int* ok;                 // [B] or null: plans the optimiser rejected have no path
    const double *T, *coef;        // ResultStore
    const double* plan_start_xyt;  // [B][3]
    double xv;
    bool standard_diff;
    double* xy_out;                // [B][P (res + 1)][2]
    double* yaw_out;               // [B][P res] or null
    int* n_out;                    // [B] points written
};
hipError_t path_points(const PathArgs& g, hipStream_t s);

// esdf_build.hip: SDFmap::updateESDF2d on the device
hipError_t esdf_fill_max(double* p, size_t n, hipStream_t s);
hipError_t esdf_update(const unsigned char* d_grid, int GLX, int GLY, double res, double x_lo, double y_lo, double odom_x, double odom_y,
                       double range, double* d_dist, hipStream_t s, int* empty_window);

} // namespace backend
