// nmpc_kernels.h -- host-side launch interface of the HIP kernels (internal).
#ifndef ALORE_NMPC_KERNELS_H
#define ALORE_NMPC_KERNELS_H

#include <hip/hip_runtime.h>

#include "../../include/alore_nmpc.h"

namespace nmpc {

// kernel parameters, passed by value
struct RtiParams {
    alore_nmpc_batch b;
    int B;
    int N;
    int n_sqp;
    int max_as_iter;
    int pg_steps; // projected-gradient steps of the working-set prediction (0 = off)
    int RS; // LDS floats per problem (row stride)
    float h, hh, c1h, c2h;
    const float* lin_x; // optional [B][(N+1)*3]: linearisation point of the first iteration (else null)
    const float* lin_u; // optional [B][N*2]
    long long* stamps; // diagnostic builds only: per-block phase cycle counts (8 per block), else null
    unsigned shared; // ALORE_NMPC_SHARED_* bits: members that are ONE copy for the whole batch (problem stride 0)
    const unsigned char* mask; // optional [B]: 0 = the problem is left exactly as it is (alore_nmpc_set_problem_mask), else null
};

// up to GROUP_MAX independent batches served by one grid of the stage-block kernel (by value in the kernel arguments: 24 x 120 B of the 4 KB they hold)
constexpr int GROUP_MAX = 24;
constexpr int TP_KMAX = 16; // two-phase grids: a tail workgroup reads the reports of its batch's blocks, up to 64 * TP_KMAX of them
// Queue of a two-phase grid, per batch (all words 0 between launches): `cnt[block]` = 1 + problems the first-pass workgroup of that block
// left to the tail (0: it has not reported yet), `entries[block * G + i]` = 1 + number of its i-th such problem (0: not written yet; the
// tail workgroup that takes an entry puts it back to 0), `exits` = tail workgroups of the batch that have left (the last one puts
// cnt[] and itself back to 0).  The first pass only stores -- no read-modify-write on a word that the other workgroups of the batch
// hit as well (one returning add per workgroup on a per-batch counter made the whole grid wait for that word: 170 against 130 us).
struct RtiGroup {
    int count;            // batches in this launch
    int blocks_per_batch; // workgroups per batch: block -> batch by division
    int stagger_blocks;   // the first stagger_blocks workgroups (one full residency of the chip) delay their start by
    int stagger_x1024;    // blockIdx * stagger_x1024 / 1024 ticks of the 100 MHz real-time counter (0: no stagger), see the kernel
    int strided;          // 1: batch i = b[0] with every member pointer advanced by i * stride[member] bytes (any count);
                          // 0: batch i = b[i] (count <= GROUP_MAX)
    long long* trace;     // diagnostic (ALORE_NMPC_TRACE): 8 words per workgroup, see the kernel; else null
    int* counter;         // persistent grid: [0] tickets handed out, [1] workgroups that have left (both 0 between launches); else null
    int persist_blocks;   // persistent grid: workgroups launched (one per SIMD slot); items beyond them are taken by ticket
    int xcd_on;           // 1: XCD x (workgroups w = x mod 8) works on xcd_share[x] consecutive blocks from xcd_base[x] (see the kernel)
    int xcd_share[8], xcd_base[8];
    unsigned long long* xcd_end; // [8][4] host memory: finishing times (100 MHz counter) of the last four workgroups of every XCD, or null
    // Two-phase grids (TWOPH builds of the kernel, see nmpc_block_kernel.hip).  The grid is dealt in UNITS of blocks_per_batch + tp_tail
    // workgroups: unit u = the blocks of batch u, then the tail workgroups of batch u - tp_lag.  Batches below tp_count2 run in two phases
    // (first pass: no working-set prediction, one sweep; problems whose working set moves are queued for the tail workgroups, which solve
    // them sixteen at a time with the full body); the rest in one pass.
    int tp_count2, tp_tail, tp_lag;
    int tp_timeout;              // ticks of the 100 MHz counter a tail workgroup waits for its chunk before it gives up (error record)
    int* tp_exits;               // [tp_count2]
    int* tp_cnt;                 // [tp_count2][blocks_per_batch]
    int* tp_entries;             // [tp_count2][blocks_per_batch * 16]
    long long* tp_trace;         // diagnostic (ALORE_NMPC_TP_TRACE): 4 words per workgroup -- real-time counter at its start, inputs landed, end; role | batch << 8 | turns << 40 -- or null
    int* tp_rec;                 // host memory: [0 .. 31] deferred problems of batch (b mod 32), [32] != 0: a tail workgroup timed out; or null
    long long stride[15]; // bytes, in the member order of alore_nmpc_batch
    alore_nmpc_batch b[GROUP_MAX];
};
static_assert(sizeof(alore_nmpc_batch) == 15 * sizeof(void*), "alore_nmpc_batch is 15 pointers");

struct LaunchGeom {
    int L;       // lanes per problem
    int G;       // problems per wavefront
    int wpb;     // wavefronts per workgroup (1 or 4)
    int wreg;    // 1: W_k of a stage lives in the registers of its lane, not in the LDS staging area
    int threads; // = L * G, multiple of 64
    int grid;
    int RS;      // wavefront kernel: LDS floats per problem; stage-block kernel: stages per lane (S)
    size_t lds_bytes;
    int block;   // 1: stage-block kernel (nmpc_block_kernel.hip), 0: wavefront kernel (nmpc_kernels.hip)
};

// number of LDS floats one problem needs (before padding) for horizon N
int rti_row_floats(int N, bool wreg);
// choose lanes/problem + block shape for (B, N); returns false if N does not fit
bool rti_geometry(int B, int N, int forced_L, int lds_limit_bytes, int n_cu, LaunchGeom* g, int forced_wpb = 0);
hipError_t launch_rti(const RtiParams& p, const LaunchGeom& g, hipStream_t s);
// stage-block kernel (nmpc_block_kernel.hip): L = 4, 8 or 16 lanes per problem, each lane owns ceil(N / L) stages
int block_lds_floats(int N, int L);
bool block_geometry(int B, int N, int forced_L, int lds_limit_bytes, int n_cu, LaunchGeom* g, int B_in_flight = 0);
hipError_t launch_rti_block(const RtiParams& p, const LaunchGeom& g, hipStream_t s);
// may a grid of this geometry run in two phases (RtiGroup::tp_*)?
bool rti_block_two_phase_supported(const RtiParams& p, const LaunchGeom& g);
hipError_t launch_rti_block_group(const RtiParams& p, const RtiGroup& grp, const LaunchGeom& g, hipStream_t s);

// nmpc_dense.hip: the condensed QP of the reference's dense interface (acadoWorkspace.H / g / lb / ub, acado_solve)
hipError_t launch_condense(const alore_nmpc_batch& b, const float* lin_x, const float* lin_u, int B, int N, float dt, unsigned shared, float* H,
                           float* g, float* lb, float* ub, hipStream_t s);
size_t condense_lds_bytes(int N);  // dynamic LDS of one workgroup of the two kernels above / below
size_t dense_qp_lds_bytes(int n);
hipError_t launch_dense_qp(int B, int n, const float* H, const float* g, const float* lb, const float* ub, float* x, float* y, int* status,
                           int* n_iter, int max_iter, hipStream_t s);

hipError_t launch_linearize(const alore_nmpc_batch& b, int B, int N, float dt, const alore_nmpc_lin_out& o,
                            hipStream_t s);
hipError_t launch_forward_simulate(const alore_nmpc_batch& b, int B, int N, float dt, hipStream_t s);
hipError_t launch_shift(const alore_nmpc_batch& b, int B, int N, float dt, int strategy, const float* xEnd,
                        const float* uEnd, hipStream_t s);
hipError_t launch_fill(float* p, float v, size_t n, hipStream_t s);

// device-side reference sampling (ref_sampler.hip)
struct RefStore {
    double* dur;   // [B][P]        piece durations
    double* coef;  // [B][P][2][6]  ascending powers
    double* ckpt;  // [B][C][2]     x, y at t = k * res
    double* meta;  // [B][8]        start_time, duration, xv(traj ICR.z), res, n_pieces, n_ckpt, valid, -
    int P, C;
};
// a batch of Polynome messages in device memory (P = stride of the per-message arrays = max pieces)
struct PolyBatch {
    const int* robot;      // [count] destination slot in the store
    const int* n_pieces;   // [count]
    const double* inner;   // [count][max(P-1,1)][2]  (theta, s)
    const double* t_pts;   // [count][P]
    const double* pva;     // [count][12]  init p0 p1 v0 v1 a0 a1, tail p0 p1 v0 v1 a0 a1
    const double* start;   // [count][3]   start_position
    const double* icr;     // [count][3]   (yr, yl, xv) as sent
    const double* t0;      // [count]      traj_start_time
    int P;
};
// workspace doubles per (message, dimension) thread of the spline kernel: knot positions P + 1, inverted Schur
// blocks 3 P, right-hand side 2 P, coefficients 6 P
inline __host__ __device__ int traj_ws_doubles(int P) { return 12 * P + 1; }
hipError_t launch_traj_build(const RefStore& s, const PolyBatch& m, int count, double res, int res_int, double* knot_ws,
                             int* n_panels, double* inc, int* overflow, hipStream_t st);
hipError_t launch_iterate_reset(const alore_nmpc_batch& b, int B, int N, const double* pose, const unsigned char* mask, hipStream_t st);
hipError_t launch_ref_eval(const RefStore& s, int B, double now, double* out /* [B][4] */, hipStream_t st);
// the planner's device-resident results (mirror of alore_backend_device_view, include/alore_backend.h)
struct BackendView {
    int P;
    const int* n_pieces;
    const double* T;
    const double* coef;
    const double* start_xytheta;
    const int* ok;
};
hipError_t launch_traj_from_backend(const RefStore& s, const BackendView& v, int count, double t0, double res, int res_int, double xv,
                                    int* n_panels, double* inc, int* overflow, hipStream_t st);
struct PlantParams { // simulator.h: max_a_, max_domega_, Pose_pub_rate_ (a period), State_Propa_rate_ (a period)
    double max_a, max_domega, pose_pub_period, propa_period;
    int substeps; // StatePropaCallback calls per control tick
};
// the plant step of tick t with the completion of tick t + 1's references (ref_sampler_device.h: plant_ahead_one)
struct PlantAhead {
    const float* u;        // [B][N][2] inputs the solve of tick t left
    float* x0;             // [B][3]    of tick t + 1
    float* y;              // [B][N][5] references of tick t + 1, sampled ahead
    float* yN;             // [B][3]
    const double* meta;    // RefStore::meta
    const double* icr;     // [B][3]
    int* at_goal;          // [B]
    double* pose;          // [B][3]
    double* vw;            // [B][2]
    const double* psi_rel; // [B][N + 1] float64 headings of the walk of tick t + 1, or null: the run ends with tick t
    PlantParams p;
    double now;            // time of tick t
    int B, N, node;
    int on;                // rti_block_sampler_kernel: 1 = this step runs in front of the solve (else the field is ignored)
};
hipError_t launch_plant(const alore_nmpc_batch& b, int B, int N, int node, const double* icr, const int* at_goal,
                        double* pose, double* vw, const PlantParams& p, hipStream_t st);
// the reference sampler's part of a grid that also solves (rti_block_sampler_kernel): pose-independent sampling of the NEXT tick
struct AheadSampler {
    RefStore store;
    float* y;            // [B][N][5] references of the next tick
    float* yN;           // [B][3]
    const double* icr;   // [B][3]
    double* psi_rel;     // [B][N + 1] float64 headings of the walk
    double dt, now;
    int B, N;
    int first_block;     // set by the launcher: workgroups from here on are the sampler's
};
int rti_block_sampler_supported(const RtiParams& p, const LaunchGeom& g); // 2: sampler + plant step in the solver's grid, 1: plant step only, 0: neither
hipError_t launch_rti_block_sampler(const RtiParams& p, const LaunchGeom& g, const AheadSampler& sa, const PlantAhead* plant, hipStream_t s);
// closed_loop_run: the pose-independent part of the sampling of a tick ahead of its pose, and the plant step that completes it
bool ref_sample_ahead_supported(int N);
hipError_t launch_ref_sample_ahead(const RefStore& s, const alore_nmpc_batch& b, int B, int N, double dt, double now, const double* icr,
                                   double* psi_rel, hipStream_t st);
hipError_t launch_plant_ahead(const PlantAhead& a, hipStream_t st);
hipError_t launch_ref_sample(const RefStore& s, const alore_nmpc_batch& b, int B, int N, double dt, double now,
                             const double* est, const double* icr, int* at_goal, double* psi_scratch, int do_smooth,
                             hipStream_t st);

} // namespace nmpc
#endif
