// nmpc_capi.hip -- implementation of the C ABI declared in include/alore_nmpc.h.
// Thin host layer: argument checks, launch geometry, HIP stream/event plumbing.
// There is no CPU path behind this ABI: without a GPU alore_nmpc_create fails.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <new>
#include <string>

#include "nmpc_kernels.h"
#include "nmpc_core.h"
#include "../../include/alore_backend.h"

struct alore_nmpc_solver {
    alore_nmpc_config cfg;
    int n_cu = 256;
    int lds_limit = 160 * 1024;
    std::string err;
    nmpc::LaunchGeom last_geom{};
    bool have_geom = false;
    bool timing = false;
    bool timed_pending = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float last_ms = -1.0f;
    // device-side reference sampling (alore_nmpc_refs_*)
    nmpc::RefStore refs{};
    int refs_B = 0;
    double* d_est = nullptr;  // [B][3]
    double* d_icr = nullptr;  // [B][3]
    double* d_psi = nullptr;  // [B][N+1]
    int* d_goal = nullptr;    // [B]
    // closed loop on the device (alore_nmpc_plant_*, alore_nmpc_closed_loop_tick): pose = d_est, ICR = d_icr
    double* d_vw = nullptr;   // [B][2] current (v, omega) of the plant
    double* d_flat = nullptr; // [B][4] alore_nmpc_refs_eval output
    unsigned char* d_mask = nullptr; // [B] alore_nmpc_closed_loop_reset
    nmpc::PlantParams plant{};
    bool has_plant = false;
    // alore_nmpc_closed_loop_run: the sampler of tick t + 1 runs in the grid of the solve of tick t, into the second of two
    // reference buffers (the caller's y / yN and these), its float64 headings into cl_psi for the plant step to complete
    float* cl_y = nullptr;   // [B][N][5]
    float* cl_yN = nullptr;  // [B][3]
    double* cl_psi[2] = {};  // [B][N + 1]
    int cl_B = 0;
    // Polynome -> store on the device: staging + workspace for chunks of kPolyChunk messages
    static constexpr int kPolyChunk = 2048;
    char* d_poly = nullptr;       // packed message arrays (layout: poly_layout)
    double* d_knot = nullptr;     // [chunk][2][traj_ws_doubles(P)] workspace of the spline kernel
    int* d_panels = nullptr;      // [chunk]
    int* d_overflow = nullptr;    // [1]
    double* d_inc = nullptr;      // [chunk][C * res_int][2], grown on demand
    int* d_panels_be = nullptr;   // [count] panels per plan (alore_nmpc_refs_set_from_backend)
    size_t panels_cap = 0;
    size_t inc_doubles = 0;
    unsigned shared = 0;          // see alore_nmpc_set_shared_members
    const unsigned char* mask = nullptr; // see alore_nmpc_set_problem_mask
    const float* lin_x = nullptr; // see alore_nmpc_set_linearization_point
    const float* lin_u = nullptr;
    // pinned staging for alore_nmpc_batch_upload / _download from pageable host memory
    char* stage_up = nullptr;
    char* stage_down = nullptr;
    char* pose_stage[2] = {};      // alore_nmpc_refs_sample: pose + ICR of a tick
    size_t pose_cap[2] = {};
    hipEvent_t pose_ev[2] = {};
    unsigned pose_turn = 0;
    size_t stage_up_cap = 0, stage_down_cap = 0;
    // alore_nmpc_rti_many: launches of independent batches in flight at once (side streams forked from the caller's)
    int overlap = 16;
    bool auto_pg = false; // warm_start_steps was left to the library: 6 for a launch on its own, 3 inside a grid of many batches
    // the last descriptor set that passed the independence check of alore_nmpc_rti_many, kept whole (with the B and the shared-member
    // mask it was checked for): any contiguous run of it is independent too
    // (four sets, least recently used replaced: a host that alternates between slot ranges -- warm-up slots and timed slots, two fleets --
    // keeps both known; with one set the second range's call overwrote the first and every call paid the check again)
    static constexpr int kIndepSets = 4;
    std::vector<alore_nmpc_batch> indep_set[kIndepSets];
    int indep_B[kIndepSets] = {0, 0, 0, 0};
    unsigned indep_shared[kIndepSets] = {0, 0, 0, 0};
    unsigned long long indep_used[kIndepSets] = {0, 0, 0, 0}, indep_clock = 0;
    int many_mode = 0; // alore_nmpc_rti_many: 0 = groups of batches per grid, 1 = one launch per batch on forked streams
    hipStream_t side[31] = {};
    hipEvent_t fork_ev = nullptr, join_ev[31] = {};
    hipEvent_t stage_up_done = nullptr; // the copies out of stage_up enqueued by the last upload
    // diagnostic phase stamps (env ALORE_NMPC_STAMPS=1): per-phase cycle shares, printed at destroy
    bool stamps = false;
    long long* d_stamps = nullptr;
    // ticket counters of the persistent grids (nmpc_block_kernel.hip: PERSIST): a ring of pairs, one pair per launch in turn, so
    // that grids of this handle that overlap on different streams never share one; every pair is back at 0 when its grid ends
    // XCD shares of the grid builds (nmpc_block_kernel.hip: RtiGroup::xcd_on): relative speed of the eight XCDs as the finishing times of
    // their last workgroups showed it at the previous launches, the host-memory record the running launch writes, its event
    double xcd_speed[8] = {1, 1, 1, 1, 1, 1, 1, 1};
    unsigned long long* xcd_rec = nullptr;  // pinned host memory [8][4] end stamps + [32] start stamp
    hipEvent_t xcd_ev = nullptr;
    bool xcd_pending = false;
    int xcd_updates = 0;
    static constexpr int kTicketRing = 16;
    // two-phase grids (nmpc_block_kernel.hip: TWOPH): queue regions (header + entries per batch, all 0 between launches), one per launch
    // in turn with the stream and an event of its last user; the host-memory record of the last launch (deferred problems per batch,
    // error word) that sizes the next launch's tail
    static constexpr int kTpRing = 4;
    char* d_tp[kTpRing] = {};
    size_t tp_cap[kTpRing] = {};
    hipStream_t tp_stream[kTpRing] = {};
    hipEvent_t tp_ev[kTpRing] = {};
    bool tp_used[kTpRing] = {};
    unsigned tp_turn = 0;
    int* tp_rec = nullptr;      // pinned host memory [40]
    hipEvent_t tp_rec_ev = nullptr;
    bool tp_rec_pending = false;
    double tp_share = 0.30;     // share of a batch's problems the tail is sized for
    int two_phase = -1;         // alore_nmpc_set_two_phase: -1 automatic, 0 never, 1 wherever the build exists
    int tp_last = 0;            // the last grid ran in two phases (launch info)
    int tp_slot_of_launch = -1;
    int tp_info[3] = {0, 0, 0}; // two-phase batches, tail workgroups per batch, lag of the last two-phase grid
    int* d_tickets = nullptr;
    unsigned ticket_turn = 0;
    size_t stamps_cap = 0;
    double stamp_sum[7] = {0, 0, 0, 0, 0, 0, 0};
    double stamp_max_total = 0;
    double stamp_max[7] = {0, 0, 0, 0, 0, 0, 0};
    double stamp_slowest[7] = {0, 0, 0, 0, 0, 0, 0};
    long stamp_n = 0;
};

namespace {

int fail(alore_nmpc_handle h, int code, const char* what, hipError_t e = hipSuccess)
{
    if (h) {
        h->err = what;
        if (e != hipSuccess) {
            h->err += ": ";
            h->err += hipGetErrorString(e);
        }
    }
    return code;
}

#define HIP_TRY(h, call)                                                   \
    do {                                                                   \
        hipError_t e_ = (call);                                            \
        if (e_ != hipSuccess) return fail(h, ALORE_NMPC_E_HIP, #call, e_); \
    } while (0)

struct Member {
    size_t offset; // byte offset of the pointer inside alore_nmpc_batch
    int per_problem(int N) const { return mult * (per_node ? (N + extra) : 1); }
    int mult;      // floats per node (or per problem when !per_node)
    bool per_node;
    int extra;     // nodes = N + extra
    bool is_int;
};

// every member of alore_nmpc_batch with its per-problem element count
const Member kMembers[] = {
    {offsetof(alore_nmpc_batch, x), 3, true, 1, false},
    {offsetof(alore_nmpc_batch, u), 2, true, 0, false},
    {offsetof(alore_nmpc_batch, od), 3, true, 1, false},
    {offsetof(alore_nmpc_batch, y), 5, true, 0, false},
    {offsetof(alore_nmpc_batch, yN), 3, false, 0, false},
    {offsetof(alore_nmpc_batch, W), 25, true, 0, false},
    {offsetof(alore_nmpc_batch, WN), 9, false, 0, false},
    {offsetof(alore_nmpc_batch, x0), 3, false, 0, false},
    {offsetof(alore_nmpc_batch, lbValues), 2, true, 0, false},
    {offsetof(alore_nmpc_batch, ubValues), 2, true, 0, false},
    {offsetof(alore_nmpc_batch, dual), 2, true, 0, false},
    {offsetof(alore_nmpc_batch, status), 1, false, 0, true},
    {offsetof(alore_nmpc_batch, n_iter), 1, false, 0, true},
    {offsetof(alore_nmpc_batch, kkt), 1, false, 0, false},
    {offsetof(alore_nmpc_batch, obj), 1, false, 0, false},
};
constexpr int kNumMembers = sizeof(kMembers) / sizeof(kMembers[0]);

void*& member_ptr(alore_nmpc_batch* b, const Member& m)
{
    return *reinterpret_cast<void**>(reinterpret_cast<char*>(b) + m.offset);
}
void* member_ptr(const alore_nmpc_batch* b, const Member& m)
{
    return *reinterpret_cast<void* const*>(reinterpret_cast<const char*>(b) + m.offset);
}

bool batch_complete(const alore_nmpc_batch* b)
{
    for (int i = 0; i < kNumMembers; ++i) {
        const size_t off = kMembers[i].offset;
        if (off == offsetof(alore_nmpc_batch, kkt) || off == offsetof(alore_nmpc_batch, obj)) continue; // optional
        if (!member_ptr(b, kMembers[i])) return false;
    }
    return true;
}

} // namespace

extern "C" {

const char* alore_nmpc_version(void) { return "alore_nmpc 0.1 (gfx950)"; }

int alore_nmpc_create(const alore_nmpc_config* cfg, alore_nmpc_handle* out)
{
    if (!cfg || !out) return ALORE_NMPC_E_INVALID;
    *out = nullptr;
    if (cfg->N < 1 || !(cfg->dt > 0.0f)) return ALORE_NMPC_E_INVALID;
    {
        const int lp = cfg->lanes_per_problem, lw = lp & ~0x100;
        const bool blk = (lp & 0x100) != 0;
        if (lp != 0 && !(blk ? (lw == 4 || lw == 8 || lw == 16 || lw == 32) : (lw == 4 || lw == 8 || lw == 16 || lw == 32 || lw == 64)))
            return ALORE_NMPC_E_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ALORE_NMPC_E_NO_DEVICE;
    if (cfg->device < 0 || cfg->device >= ndev) return ALORE_NMPC_E_INVALID;
    if (hipSetDevice(cfg->device) != hipSuccess) return ALORE_NMPC_E_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) return ALORE_NMPC_E_NO_DEVICE;
    alore_nmpc_solver* h = new (std::nothrow) alore_nmpc_solver;
    if (!h) return ALORE_NMPC_E_NOMEM;
    h->cfg = *cfg;
    if (h->cfg.max_as_iter <= 0) h->cfg.max_as_iter = 128;
    h->auto_pg = h->cfg.warm_start_steps < 0;
    // swept on MI355X: N = 20 -- 4 / 6 / 8 steps: 23.8 / 21.8 / 22.6 us per in-order launch of 4096; N = 50 (round 6, with the scanned backward
    // sweep a prediction step is cheap beside a second sweep) -- 4 / 6 / 8 / 10: 52.4 / 48.8 / 45.7 / 47.8 us (tools/horizon_in_flight.py)
    if (h->auto_pg) h->cfg.warm_start_steps = h->cfg.N > 32 ? 8 : 6;
    h->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    h->lds_limit = (int)prop.sharedMemPerBlock > 0 ? (int)prop.sharedMemPerBlock : 64 * 1024;
    if (prop.maxSharedMemoryPerMultiProcessor > (size_t)h->lds_limit)
        h->lds_limit = (int)prop.maxSharedMemoryPerMultiProcessor;
    if (h->lds_limit > 160 * 1024) h->lds_limit = 160 * 1024;
    // the horizon must fit the on-chip layout with at least one geometry
    nmpc::LaunchGeom g;
    const bool fits = (cfg->lanes_per_problem & 0x100)
                          ? nmpc::block_geometry(1, cfg->N, cfg->lanes_per_problem & 0xff, h->lds_limit, h->n_cu, &g)
                          : nmpc::rti_geometry(1, cfg->N, cfg->lanes_per_problem, h->lds_limit, h->n_cu, &g);
    if (!fits) {
        delete h;
        return ALORE_NMPC_E_UNSUPPORTED;
    }
    if (hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess) {
        delete h;
        return ALORE_NMPC_E_HIP;
    }
    // streams / events of alore_nmpc_rti_many: made here, not at first use, so that a first use inside a stream capture
    // creates nothing
    bool forks_ok = hipEventCreateWithFlags(&h->fork_ev, hipEventDisableTiming) == hipSuccess;
    for (int w = 0; w < 31 && forks_ok; ++w)
        forks_ok = hipStreamCreateWithFlags(&h->side[w], hipStreamNonBlocking) == hipSuccess &&
                   hipEventCreateWithFlags(&h->join_ev[w], hipEventDisableTiming) == hipSuccess;
    if (!forks_ok) {
        (void)alore_nmpc_destroy(h);
        return ALORE_NMPC_E_HIP;
    }
    if (hipMalloc((void**)&h->d_tickets, sizeof(int) * 2 * alore_nmpc_solver::kTicketRing) != hipSuccess ||
        hipMemset(h->d_tickets, 0, sizeof(int) * 2 * alore_nmpc_solver::kTicketRing) != hipSuccess) {
        (void)alore_nmpc_destroy(h);
        return ALORE_NMPC_E_HIP;
    }
    if (hipHostMalloc((void**)&h->xcd_rec, sizeof(unsigned long long) * 40, hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&h->xcd_ev, hipEventDisableTiming) != hipSuccess) {
        (void)alore_nmpc_destroy(h);
        return ALORE_NMPC_E_HIP;
    }
    std::memset(h->xcd_rec, 0, sizeof(unsigned long long) * 40);
    if (hipHostMalloc((void**)&h->tp_rec, sizeof(int) * 40, hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&h->tp_rec_ev, hipEventDisableTiming) != hipSuccess) {
        (void)alore_nmpc_destroy(h);
        return ALORE_NMPC_E_HIP;
    }
    std::memset(h->tp_rec, 0, sizeof(int) * 40);
    for (int i = 0; i < alore_nmpc_solver::kTpRing; ++i)
        if (hipEventCreateWithFlags(&h->tp_ev[i], hipEventDisableTiming) != hipSuccess) {
            (void)alore_nmpc_destroy(h);
            return ALORE_NMPC_E_HIP;
        }
    if (const char* sp = std::getenv("ALORE_NMPC_XCD_SPEEDS")) { // diagnostic / tests: preset relative speeds "a,b,c,d,e,f,g,h"
        double v[8];
        if (std::sscanf(sp, "%lf,%lf,%lf,%lf,%lf,%lf,%lf,%lf", v, v + 1, v + 2, v + 3, v + 4, v + 5, v + 6, v + 7) == 8)
            for (int x = 0; x < 8; ++x) h->xcd_speed[x] = v[x] > 0.05 ? v[x] : 0.05;
    }
    const char* st = std::getenv("ALORE_NMPC_STAMPS");
    h->stamps = st && st[0] == '1';
    *out = h;
    return ALORE_NMPC_OK;
}

int alore_nmpc_destroy(alore_nmpc_handle h)
{
    if (!h) return ALORE_NMPC_E_INVALID;
    if (h->stamps && h->stamp_n > 0) {
        static const char* names[7] = {"load+linearise", "backward sweeps", "forward sweeps", "kkt+expand",
                                       "objective+store", "total", "ws prediction"};
        std::fprintf(stderr, "[alore_nmpc stamps] mean cycles per workgroup over %ld workgroup-launches:\n", h->stamp_n);
        for (int i = 0; i < 7; ++i)
            std::fprintf(stderr, "  %-16s %10.0f  (%5.1f %%)\n", names[i], h->stamp_sum[i] / h->stamp_n,
                         100.0 * h->stamp_sum[i] / h->stamp_sum[5]);
        std::fprintf(stderr, "  slowest workgroup total: %.0f cycles; its phases:", h->stamp_max_total);
        for (int i = 0; i < 7; ++i) std::fprintf(stderr, " %.0f", h->stamp_slowest[i]);
        std::fprintf(stderr, "\n  per-phase maxima:");
        for (int i = 0; i < 7; ++i) std::fprintf(stderr, " %.0f", h->stamp_max[i]);
        std::fprintf(stderr, "\n");
    }
    if (h->d_stamps) (void)hipFree(h->d_stamps);
    if (h->d_tickets) (void)hipFree(h->d_tickets);
    if (std::getenv("ALORE_NMPC_XCD_DEBUG") && h->xcd_updates > 0)
        std::fprintf(stderr, "[alore_nmpc xcd shares] %d updates; relative speeds %.3f %.3f %.3f %.3f %.3f %.3f %.3f %.3f\n", h->xcd_updates, h->xcd_speed[0],
                     h->xcd_speed[1], h->xcd_speed[2], h->xcd_speed[3], h->xcd_speed[4], h->xcd_speed[5], h->xcd_speed[6], h->xcd_speed[7]);
    if (h->xcd_ev) { (void)hipEventSynchronize(h->xcd_ev); (void)hipEventDestroy(h->xcd_ev); }
    if (h->xcd_rec) (void)hipHostFree(h->xcd_rec);
    if (h->tp_rec_ev) { (void)hipEventSynchronize(h->tp_rec_ev); (void)hipEventDestroy(h->tp_rec_ev); }
    if (h->tp_rec) (void)hipHostFree(h->tp_rec);
    for (int i = 0; i < alore_nmpc_solver::kTpRing; ++i) {
        if (h->tp_ev[i]) { (void)hipEventSynchronize(h->tp_ev[i]); (void)hipEventDestroy(h->tp_ev[i]); }
        if (h->d_tp[i]) (void)hipFree(h->d_tp[i]);
    }
    for (int w = 0; w < 31; ++w) {
        if (h->side[w]) (void)hipStreamDestroy(h->side[w]);
        if (h->join_ev[w]) (void)hipEventDestroy(h->join_ev[w]);
    }
    if (h->fork_ev) (void)hipEventDestroy(h->fork_ev);
    if (h->refs.dur) (void)hipFree(h->refs.dur);
    if (h->refs.coef) (void)hipFree(h->refs.coef);
    if (h->refs.ckpt) (void)hipFree(h->refs.ckpt);
    if (h->refs.meta) (void)hipFree(h->refs.meta);
    for (int i = 0; i < 2; ++i)
        if (h->cl_psi[i]) (void)hipFree(h->cl_psi[i]);
    for (int i = 0; i < 2; ++i) {
        if (h->pose_ev[i]) { (void)hipEventSynchronize(h->pose_ev[i]); (void)hipEventDestroy(h->pose_ev[i]); }
        if (h->pose_stage[i]) (void)hipHostFree(h->pose_stage[i]);
    }
    if (h->cl_y) (void)hipFree(h->cl_y);
    if (h->cl_yN) (void)hipFree(h->cl_yN);
    if (h->d_est) (void)hipFree(h->d_est); // d_icr is its second half
    if (h->d_psi) (void)hipFree(h->d_psi);
    if (h->d_goal) (void)hipFree(h->d_goal);
    if (h->d_vw) (void)hipFree(h->d_vw);
    if (h->d_flat) (void)hipFree(h->d_flat);
    if (h->d_mask) (void)hipFree(h->d_mask);
    if (h->d_poly) (void)hipFree(h->d_poly);
    if (h->d_knot) (void)hipFree(h->d_knot);
    if (h->d_panels) (void)hipFree(h->d_panels);
    if (h->d_overflow) (void)hipFree(h->d_overflow);
    if (h->d_inc) (void)hipFree(h->d_inc);
    if (h->d_panels_be) (void)hipFree(h->d_panels_be);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stage_up_done) { (void)hipEventSynchronize(h->stage_up_done); (void)hipEventDestroy(h->stage_up_done); }
    if (h->stage_up) (void)hipHostFree(h->stage_up);
    if (h->stage_down) (void)hipHostFree(h->stage_down);
    delete h;
    return ALORE_NMPC_OK;
}

const char* alore_nmpc_last_error(alore_nmpc_handle h) { return h ? h->err.c_str() : "null handle"; }

int alore_nmpc_batch_alloc(alore_nmpc_handle h, int B, alore_nmpc_batch* out)
{
    if (!h || !out || B <= 0) return fail(h, ALORE_NMPC_E_INVALID, "batch_alloc: bad argument");
    std::memset(out, 0, sizeof(*out));
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    for (int i = 0; i < kNumMembers; ++i) {
        const Member& m = kMembers[i];
        const size_t bytes = (size_t)B * m.per_problem(h->cfg.N) * 4;
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {
            alore_nmpc_batch_free(h, out);
            return fail(h, ALORE_NMPC_E_NOMEM, "hipMalloc", e);
        }
        (void)hipMemset(p, 0, bytes);
        member_ptr(out, m) = p;
    }
    return ALORE_NMPC_OK;
}

int alore_nmpc_batch_free(alore_nmpc_handle h, alore_nmpc_batch* b)
{
    if (!h || !b) return ALORE_NMPC_E_INVALID;
    for (int i = 0; i < kNumMembers; ++i) {
        void*& p = member_ptr(b, kMembers[i]);
        if (p) (void)hipFree(p);
        p = nullptr;
    }
    return ALORE_NMPC_OK;
}

// Host memory the runtime can DMA from directly (hipHostMalloc / hipHostRegister)?
static bool host_is_pinned(const void* p)
{
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError(); // an unregistered pointer is reported as an error: not one of ours
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

static int grow_stage(alore_nmpc_handle h, char*& buf, size_t& cap, size_t need)
{
    if (need <= cap) return ALORE_NMPC_OK;
    if (buf) (void)hipHostFree(buf);
    buf = nullptr; cap = 0;
    hipError_t e = hipHostMalloc((void**)&buf, need, hipHostMallocDefault);
    if (e != hipSuccess) return fail(h, ALORE_NMPC_E_NOMEM, "hipHostMalloc (staging)", e);
    cap = need;
    return ALORE_NMPC_OK;
}

// Pageable host members go through one pinned staging buffer per direction (a hipMemcpyAsync from pageable memory
// is staged by the runtime in 4 MB pieces with a host wait per piece: 2-3 x slower and never asynchronous);
// members already in pinned memory (alore_nmpc_host_alloc) are copied in place.
static int batch_copy(alore_nmpc_handle h, const alore_nmpc_batch* dev, const alore_nmpc_batch* host, int B,
                      void* stream, bool to_device)
{
    if (!h || !dev || !host || B <= 0) return fail(h, ALORE_NMPC_E_INVALID, "batch copy: bad argument");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    size_t staged = 0;
    bool pinned[kNumMembers];
    for (int i = 0; i < kNumMembers; ++i) {
        const Member& m = kMembers[i];
        void* hp = member_ptr(host, m);
        pinned[i] = false;
        if (!member_ptr(dev, m) || !hp) continue;
        pinned[i] = host_is_pinned(hp);
        if (!pinned[i]) staged += (((size_t)B * m.per_problem(h->cfg.N) * 4) + 255) & ~(size_t)255;
    }
    char* stage = nullptr;
    if (staged) {
        if (to_device) {
            if (h->stage_up_done) HIP_TRY(h, hipEventSynchronize(h->stage_up_done)); // the previous upload still reads it
            if (int rc = grow_stage(h, h->stage_up, h->stage_up_cap, staged)) return rc;
            stage = h->stage_up;
        } else {
            if (int rc = grow_stage(h, h->stage_down, h->stage_down_cap, staged)) return rc;
            stage = h->stage_down;
        }
    }
    size_t off = 0;
    for (int i = 0; i < kNumMembers; ++i) {
        const Member& m = kMembers[i];
        void* d = member_ptr(dev, m);
        void* hp = member_ptr(host, m);
        if (!d || !hp) continue;
        const size_t bytes = (size_t)B * m.per_problem(h->cfg.N) * 4;
        void* src_dst = hp;
        if (!pinned[i]) {
            src_dst = stage + off;
            off += (bytes + 255) & ~(size_t)255;
            if (to_device) std::memcpy(src_dst, hp, bytes);
        }
        if (to_device)
            HIP_TRY(h, hipMemcpyAsync(d, src_dst, bytes, hipMemcpyHostToDevice, s));
        else
            HIP_TRY(h, hipMemcpyAsync(src_dst, d, bytes, hipMemcpyDeviceToHost, s));
    }
    if (staged && to_device) {
        if (!h->stage_up_done) HIP_TRY(h, hipEventCreateWithFlags(&h->stage_up_done, hipEventDisableTiming));
        HIP_TRY(h, hipEventRecord(h->stage_up_done, s));
    }
    if (staged && !to_device) { // pageable destinations: complete the transfer, then hand the data over
        HIP_TRY(h, hipStreamSynchronize(s));
        off = 0;
        for (int i = 0; i < kNumMembers; ++i) {
            const Member& m = kMembers[i];
            void* hp = member_ptr(host, m);
            if (!member_ptr(dev, m) || !hp || pinned[i]) continue;
            const size_t bytes = (size_t)B * m.per_problem(h->cfg.N) * 4;
            std::memcpy(hp, stage + off, bytes);
            off += (bytes + 255) & ~(size_t)255;
        }
    }
    return ALORE_NMPC_OK;
}

int alore_nmpc_host_alloc(size_t bytes, void** out)
{
    if (!out || bytes == 0) return ALORE_NMPC_E_INVALID;
    *out = nullptr;
    return hipHostMalloc(out, bytes, hipHostMallocDefault) == hipSuccess ? ALORE_NMPC_OK : ALORE_NMPC_E_NOMEM;
}

int alore_nmpc_host_free(void* p)
{
    if (!p) return ALORE_NMPC_OK;
    return hipHostFree(p) == hipSuccess ? ALORE_NMPC_OK : ALORE_NMPC_E_HIP;
}

int alore_nmpc_batch_upload(alore_nmpc_handle h, const alore_nmpc_batch* dev, const alore_nmpc_batch* host, int B,
                            void* stream)
{
    return batch_copy(h, dev, host, B, stream, true);
}

int alore_nmpc_batch_download(alore_nmpc_handle h, const alore_nmpc_batch* dev, const alore_nmpc_batch* host, int B,
                              void* stream)
{
    return batch_copy(h, dev, host, B, stream, false);
}

int alore_nmpc_batch_default_bounds(alore_nmpc_handle h, const alore_nmpc_batch* dev, int B, void* stream)
{
    if (!h || !dev || B <= 0 || !dev->lbValues || !dev->ubValues)
        return fail(h, ALORE_NMPC_E_INVALID, "default_bounds: bad argument");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const size_t n = (size_t)B * h->cfg.N * 2;
    HIP_TRY(h, nmpc::launch_fill(const_cast<float*>(dev->lbValues), -3.0f, n, (hipStream_t)stream));
    HIP_TRY(h, nmpc::launch_fill(const_cast<float*>(dev->ubValues), 3.0f, n, (hipStream_t)stream));
    return ALORE_NMPC_OK;
}

namespace {

// may this batch run on the stage-block kernel (nmpc_block_kernel.hip)?  Not when the caller forces lanes of the wavefront
// kernel, a separate linearisation point is set, or a member is not 16-byte aligned.
bool block_eligible(alore_nmpc_handle h, const alore_nmpc_batch* dev)
{
    static const char* force_kernel = getenv("ALORE_NMPC_KERNEL"); // diagnostic: "wave" or "block"
    const int lp = h->cfg.lanes_per_problem;
    bool ok = (lp == 0 || (lp & 0x100)) && !(force_kernel && force_kernel[0] == 'w') && !h->lin_x;
    const void* ptrs[] = {dev->x, dev->u, dev->od, dev->y, dev->W, dev->lbValues, dev->ubValues, dev->dual};
    for (const void* q : ptrs) ok = ok && ((reinterpret_cast<size_t>(q) & 15) == 0);
    return ok;
}

void fill_params(alore_nmpc_handle h, const alore_nmpc_batch* dev, int B, int n_sqp, const nmpc::LaunchGeom& g, nmpc::RtiParams* p)
{
    p->b = *dev;
    p->B = B;
    p->N = h->cfg.N;
    p->n_sqp = n_sqp;
    p->max_as_iter = h->cfg.max_as_iter;
    p->pg_steps = h->cfg.warm_start_steps;
    p->shared = h->shared;
    static const bool no_scan = getenv("ALORE_NMPC_SCAN") && atoi(getenv("ALORE_NMPC_SCAN")) == 0; // diagnostic: backward sweeps never as a scan over the lanes
    if (no_scan) p->shared |= 0x80000000u;
    p->RS = g.RS;
    const nmpc::IrkConst K = nmpc::make_irk(h->cfg.dt);
    p->h = K.h; p->hh = K.hh; p->c1h = K.c1h; p->c2h = K.c2h;
    p->stamps = nullptr;
    p->lin_x = h->lin_x;
    p->lin_u = h->lin_u;
    p->mask = nullptr;
}

// one launch for one batch; B_in_flight = problems of all launches that run concurrently with it (0: only this one)
int rti_group(alore_nmpc_handle h, const alore_nmpc_batch* batches, int count, int B, int n_sqp, void* stream, int B_in_flight,
              const long long* stride);

// `co` (alore_nmpc_closed_loop_run): the sampler of the next tick, to run in the same grid when the mapping has such a build;
// *co_done says whether it did
int rti_one(alore_nmpc_handle h, const alore_nmpc_batch* dev, int B, int n_sqp, void* stream, int B_in_flight, const nmpc::AheadSampler* co = nullptr,
            bool* co_done = nullptr, const nmpc::PlantAhead* plant = nullptr)
{
    if (co_done) *co_done = false;
    if (!batch_complete(dev)) return fail(h, ALORE_NMPC_E_INVALID, "rti: batch has NULL members");
    nmpc::LaunchGeom g;
    static const int forced_wpb = getenv("ALORE_NMPC_WPB") ? atoi(getenv("ALORE_NMPC_WPB")) : 0; // diagnostic: 1 or 4
    const int lp = h->cfg.lanes_per_problem;
    const bool use_block = block_eligible(h, dev) &&
                           nmpc::block_geometry(B, h->cfg.N, lp & 0xff, h->lds_limit, h->n_cu, &g, B_in_flight);
    if (!use_block && !nmpc::rti_geometry(B, h->cfg.N, (lp & 0x100) ? 0 : lp, h->lds_limit, h->n_cu, &g, forced_wpb))
        return fail(h, ALORE_NMPC_E_UNSUPPORTED, "rti: horizon does not fit the LDS layout");
    nmpc::RtiParams p;
    fill_params(h, dev, B, n_sqp, g, &p);
    p.mask = h->mask;
    // one robot / a handful (a few wavefronts, each alone on its SIMD): a second, partial sweep costs less than the prediction steps that
    // would avoid it -- B = 1, N = 20, cold start, synchronous launch p50: 3 / 4 / 6 / 8 steps 22.7 / 23.6 / 23.8 / 24.8 us (round 6)
    if (h->auto_pg && B <= 64 && h->cfg.N <= 32) p.pg_steps = 3;
    hipStream_t s = (hipStream_t)stream;
    if (h->stamps) {
        const size_t need = (size_t)g.grid * g.wpb * 8; // one record per wavefront
        if (need > h->stamps_cap) {
            if (h->d_stamps) (void)hipFree(h->d_stamps);
            h->d_stamps = nullptr;
            h->stamps_cap = 0;
            HIP_TRY(h, hipMalloc((void**)&h->d_stamps, need * sizeof(long long)));
            h->stamps_cap = need;
        }
        p.stamps = h->d_stamps;
    }
    if (h->timing) HIP_TRY(h, hipEventRecord(h->ev0, s));
    // one batch that is a grid of several residencies by itself (B >= 32768 on the packed mapping): what alore_nmpc_rti_many gives the
    // batches of a call -- staggered first residency, XCD shares, the prediction length of a full chip -- applies to it as it stands
    static const bool big_as_group = !(getenv("ALORE_NMPC_BIG_AS_GROUP") && atoi(getenv("ALORE_NMPC_BIG_AS_GROUP")) == 0);
    if (big_as_group && use_block && g.L == 4 && g.RS == 5 && !h->stamps && !co && !plant && (long)g.grid >= 2L * 4 * h->n_cu) {
        const int rc = rti_group(h, dev, 1, B, n_sqp, stream, B, nullptr);
        if (rc != ALORE_NMPC_OK) return rc;
        if (h->timing) {
            HIP_TRY(h, hipEventRecord(h->ev1, s));
            h->timed_pending = true;
        }
        h->last_geom = g;
        h->have_geom = true;
        return ALORE_NMPC_OK;
    }
    const int fused = (co && !h->timing) ? nmpc::rti_block_sampler_supported(p, g) : 0;
    if (fused == 2 || (fused == 1 && plant)) {
        nmpc::AheadSampler sa = *co;
        if (fused == 1) sa.B = 0; // this build has no room for the sampler's wavefronts: the caller launches it behind the solve
        HIP_TRY(h, nmpc::launch_rti_block_sampler(p, g, sa, plant, s));
        *co_done = fused == 2 || co->B == 0;
    } else {
        if (plant) HIP_TRY(h, nmpc::launch_plant_ahead(*plant, s)); // no build of this mapping carries it: its own launch, in front of the solve
        HIP_TRY(h, g.block ? nmpc::launch_rti_block(p, g, s) : nmpc::launch_rti(p, g, s));
    }
    if (h->timing) {
        HIP_TRY(h, hipEventRecord(h->ev1, s));
        h->timed_pending = true;
    }
    h->last_geom = g;
    h->have_geom = true;
    if (h->stamps) { // diagnostic mode: synchronous, never used for timing
        const int n_waves = g.grid * g.wpb;
        std::vector<long long> host((size_t)n_waves * 8);
        HIP_TRY(h, hipStreamSynchronize(s));
        HIP_TRY(h, hipMemcpy(host.data(), h->d_stamps, host.size() * sizeof(long long), hipMemcpyDeviceToHost));
        if (const char* dump = getenv("ALORE_NMPC_STAMPS_DUMP")) { // raw per-wavefront stamps of the last launch
            if (FILE* f = std::fopen(dump, "wb")) { std::fwrite(host.data(), sizeof(long long), host.size(), f); std::fclose(f); }
        }
        for (int b = 0; b < n_waves; ++b) {
            for (int i = 0; i < 7; ++i) h->stamp_sum[i] += (double)host[(size_t)b * 8 + i];
            for (int i = 0; i < 7; ++i)
                if ((double)host[(size_t)b * 8 + i] > h->stamp_max[i]) h->stamp_max[i] = (double)host[(size_t)b * 8 + i];
            if ((double)host[(size_t)b * 8 + 5] > h->stamp_max_total) {
                h->stamp_max_total = (double)host[(size_t)b * 8 + 5];
                for (int i = 0; i < 7; ++i) h->stamp_slowest[i] = (double)host[(size_t)b * 8 + i];
            }
        }
        h->stamp_n += n_waves;
    }
    return ALORE_NMPC_OK;
}

// Are the batches independent problem sets?  Every array a batch WRITES (x, u, dual, status, n_iter, kkt, obj) must be
// disjoint, as an address range, from every array another batch reads or writes.  One sort of the 15 x count ranges and a
// sweep that remembers, for writes and for all ranges, the two furthest-reaching ranges of distinct batches.
bool batches_independent(alore_nmpc_handle h, const alore_nmpc_batch* batches, int count, int B)
{
    struct Range { size_t lo, hi; int id; bool write; };
    std::vector<Range> r;
    r.reserve((size_t)count * kNumMembers);
    const int N = h->cfg.N;
    for (int i = 0; i < count; ++i) {
        for (int m = 0; m < kNumMembers; ++m) {
            const void* q = member_ptr(batches + i, kMembers[m]);
            if (!q) continue;
            const size_t off = kMembers[m].offset;
            const bool write = off == offsetof(alore_nmpc_batch, x) || off == offsetof(alore_nmpc_batch, u) ||
                               off == offsetof(alore_nmpc_batch, dual) || off == offsetof(alore_nmpc_batch, status) ||
                               off == offsetof(alore_nmpc_batch, n_iter) || off == offsetof(alore_nmpc_batch, kkt) ||
                               off == offsetof(alore_nmpc_batch, obj);
            bool one_copy = false; // members that are ONE copy for the batch (alore_nmpc_set_shared_members)
            if (h->shared & ALORE_NMPC_SHARED_W) one_copy = one_copy || off == offsetof(alore_nmpc_batch, W) || off == offsetof(alore_nmpc_batch, WN);
            if (h->shared & ALORE_NMPC_SHARED_BOUNDS) one_copy = one_copy || off == offsetof(alore_nmpc_batch, lbValues) || off == offsetof(alore_nmpc_batch, ubValues);
            if (h->shared & ALORE_NMPC_SHARED_OD) one_copy = one_copy || off == offsetof(alore_nmpc_batch, od);
            const size_t bytes = (size_t)(one_copy ? 1 : B) * kMembers[m].per_problem(N) * 4;
            const size_t lo = reinterpret_cast<size_t>(q);
            r.push_back({lo, lo + bytes, i, write});
        }
    }
    std::sort(r.begin(), r.end(), [](const Range& a, const Range& b) { return a.lo < b.lo; });
    struct Top2 { size_t hi[2] = {0, 0}; int id[2] = {-1, -1};
        void add(size_t h_, int i_) {
            if (i_ == id[0]) { if (h_ > hi[0]) hi[0] = h_; }
            else if (h_ > hi[0]) { if (id[0] != -1) { hi[1] = hi[0]; id[1] = id[0]; } hi[0] = h_; id[0] = i_; }
            else if (i_ == id[1]) { if (h_ > hi[1]) hi[1] = h_; }
            else if (h_ > hi[1]) { hi[1] = h_; id[1] = i_; }
        }
        bool other_reaches(size_t lo, int i_) const { return (id[0] != -1 && id[0] != i_ && hi[0] > lo) || (id[1] != -1 && id[1] != i_ && hi[1] > lo); }
    } writes, all;
    for (const Range& q : r) {
        if (writes.other_reaches(q.lo, q.id)) return false;          // somebody else writes into what this batch touches
        if (q.write && all.other_reaches(q.lo, q.id)) return false;  // this batch writes into what somebody else touches
        all.add(q.hi, q.id);
        if (q.write) writes.add(q.hi, q.id);
    }
    return true;
}

// Is (batches, count) a contiguous run of the descriptor set that last passed the check, for the same B and shared members?  The
// descriptors are compared, not a digest of them.
bool known_independent(alore_nmpc_handle h, const alore_nmpc_batch* batches, int count, int B)
{
    for (int k = 0; k < alore_nmpc_solver::kIndepSets; ++k) {
        const size_t n = h->indep_set[k].size();
        if (n == 0 || (size_t)count > n || h->indep_B[k] != B || h->indep_shared[k] != h->shared) continue;
        for (size_t i0 = 0; i0 + (size_t)count <= n; ++i0)
            if (h->indep_set[k][i0].x == batches[0].x && std::memcmp(&h->indep_set[k][i0], batches, (size_t)count * sizeof(alore_nmpc_batch)) == 0) {
                h->indep_used[k] = ++h->indep_clock;
                return true;
            }
    }
    return false;
}
void remember_independent(alore_nmpc_handle h, const alore_nmpc_batch* batches, int count, int B)
{
    int k = 0;
    for (int i = 1; i < alore_nmpc_solver::kIndepSets; ++i)
        if (h->indep_used[i] < h->indep_used[k]) k = i;
    h->indep_set[k].assign(batches, batches + count);
    h->indep_B[k] = B;
    h->indep_shared[k] = h->shared;
    h->indep_used[k] = ++h->indep_clock;
}

// `count` independent batches on the stage-block kernel as ONE grid (nmpc_block_kernel.hip: RtiGroup): by table
// (count <= GROUP_MAX) or, with `stride`, by constant member strides (any count)
int rti_group(alore_nmpc_handle h, const alore_nmpc_batch* batches, int count, int B, int n_sqp, void* stream, int B_in_flight,
              const long long* stride)
{
    nmpc::LaunchGeom g;
    if (!nmpc::block_geometry(B, h->cfg.N, h->cfg.lanes_per_problem & 0xff, h->lds_limit, h->n_cu, &g, B_in_flight))
        return fail(h, ALORE_NMPC_E_UNSUPPORTED, "rti_many: horizon does not fit the stage-block kernel");
    nmpc::RtiParams p;
    fill_params(h, batches, B, n_sqp, g, &p);
    p.mask = h->mask; // problem b of EVERY batch of the call (alore_nmpc_set_problem_mask)
    // a launch on its own lasts as long as its slowest wavefront, so the prediction runs until (nearly) no problem needs a
    // second sweep (6 .. 9 steps); when the chip is full of wavefronts the 2 % of problems that get one with 3 .. 4 steps cost
    // less than the steps saved (profiles/r04_block_kernel_experiments.txt)
    if (h->auto_pg && (long)B * count >= 16384) p.pg_steps = 4; // round 5 (a prediction step costs 443 instructions, was 573): 2 / 3 / 4 / 5 / 6 steps: 5.97 / 5.61 / 5.50 / 5.72 / 6.20 us per batch
    nmpc::RtiGroup grp;
    grp.count = count;
    grp.blocks_per_batch = g.grid;
    grp.strided = stride ? 1 : 0;
    // staggered start of the first residency (see the kernel): one wavefront per SIMD at this kernel's register count, spread
    // over the time HBM needs for their inputs at ~5 TB/s; only for grids of at least two residencies
    grp.stagger_blocks = 0;
    grp.stagger_x1024 = 0;
    {
        const long resident = 4L * h->n_cu;
        static const char* env_ns = getenv("ALORE_NMPC_STAGGER_NS"); // diagnostic: total spread in ns (0 = off)
        const double bytes_per_block = 4.0 * (51.0 * h->cfg.N + 28.0) * g.G;
        // bytes / (7.5e12 B/s) in ns: 9 us for 1024 wavefronts x 67 KB (round 5, profiles/r05_b_stagger_and_pg_sweep.txt: 6 .. 11 us are
        // equally good, 0 costs 15 us per 20-batch grid, 14 and more 1 .. 3 us)
        double spread_ns = resident * bytes_per_block / 7.5e3;
        if (env_ns) spread_ns = atof(env_ns);
        if ((long)g.grid * count >= 2 * resident && spread_ns > 0.0) {
            grp.stagger_blocks = (int)resident;
            grp.stagger_x1024 = (int)(spread_ns / 10.0 / resident * 1024.0 + 0.5); // ticks of 10 ns per block, x 1024
        }
    }
    for (int m = 0; m < 15; ++m) grp.stride[m] = stride ? stride[m] : 0;
    const int n_tab = stride ? 1 : count;
    for (int i = 0; i < n_tab; ++i) grp.b[i] = batches[i];
    // diagnostic (ALORE_NMPC_TRACE=<file>, never under a stream capture): the instrumented twin of the grid build leaves 8 words per
    // workgroup (real-time counter at start / after the stagger / inputs landed / last store / stores acknowledged, HW_ID, sweeps,
    // diagonal path); the launch is synchronous, one file <file>.<n> per traced grid (tools/trace_timeline.py)
    // ALORE_NMPC_PERSIST=1 (measured alternative, off by default): grids of at least two residencies of the build that fills the
    // (4, 5) mapping run persistent -- one workgroup per SIMD slot, blocks of problems taken by ticket (see the kernel).  It evens out
    // the XCDs (the hardware deals a plain grid to them in fixed shares and they differ by ~5 % in speed: tail 3.6 -> 1.4 % of a
    // 200-batch grid), but the loop around the body costs the register allocation more than that: profiles/r05_persistent_grid.txt
    static const bool persist_on = getenv("ALORE_NMPC_PERSIST") && atoi(getenv("ALORE_NMPC_PERSIST")) == 1;
    grp.counter = nullptr;
    grp.persist_blocks = 0;
    if (persist_on && g.L == 4 && g.RS == 5 && h->cfg.N == 20 && n_sqp == 1 && (long)g.grid * count >= 2L * 4 * h->n_cu) {
        grp.counter = h->d_tickets + 2 * (h->ticket_turn++ % alore_nmpc_solver::kTicketRing);
        grp.persist_blocks = 4 * h->n_cu;
    }
    // Two-phase grid (see the kernel): grids of several residencies of the (4, 5) grid build whose batches are many enough that the tail
    // of a batch can follow its first pass at a distance of more than a residency and the last batches can run in one pass.
    // alore_nmpc_set_two_phase / ALORE_NMPC_TWO_PHASE (diagnostic): 0 never, 1 wherever the build exists.
    grp.tp_count2 = 0; grp.tp_tail = 0; grp.tp_lag = 0; grp.tp_timeout = 0; grp.tp_exits = nullptr; grp.tp_cnt = nullptr; grp.tp_entries = nullptr; grp.tp_trace = nullptr; grp.tp_rec = nullptr;
    h->tp_last = 0;
    {
        static const char* env_tp = getenv("ALORE_NMPC_TWO_PHASE");
        static const char* env_lag = getenv("ALORE_NMPC_TP_LAG");       // diagnostic: units between a first pass and its tail
        static const char* env_single = getenv("ALORE_NMPC_TP_SINGLE"); // diagnostic: batches at the end of the grid that run in one pass
        static const char* env_tail = getenv("ALORE_NMPC_TP_TAIL");     // diagnostic: tail workgroups per batch
        // automatic = off: measured on the bench distribution (a fifth of the problems queued) the mode loses 8 % -- the queued problems pay
        // the 5 - 9 us between a workgroup's start and its inputs a second time (profiles/r06_two_phase.txt); it gains below ~a tenth queued
        const int want = env_tp ? atoi(env_tp) : (h->two_phase < 0 ? 0 : h->two_phase);
        const long resident = 4L * h->n_cu;
        // an error the last two-phase grid left: the queue is put back in order, the call fails loudly
        if (h->tp_rec_pending && hipEventQuery(h->tp_rec_ev) == hipSuccess) {
            h->tp_rec_pending = false;
            if (h->tp_rec[32] != 0) {
                const int code = h->tp_rec[32];
                h->tp_rec[32] = 0;
                (void)hipDeviceSynchronize();
                for (int i = 0; i < alore_nmpc_solver::kTpRing; ++i)
                    if (h->d_tp[i]) (void)hipMemset(h->d_tp[i], 0, h->tp_cap[i]);
                char msg[256];
                std::snprintf(msg, sizeof msg, "rti_many: a tail workgroup of the previous two-phase grid gave up waiting for its queue (code %d); "
                                               "that grid left problems unsolved (alore_nmpc_set_two_phase(h, 0) turns the mode off)", code);
                return fail(h, ALORE_NMPC_E_HIP, msg);
            }
            int mx = 0;
            for (int i = 0; i < 32; ++i) mx = h->tp_rec[i] > mx ? h->tp_rec[i] : mx;
            if (mx > 0) h->tp_share = (double)mx / (double)B; // the fullest queue of the last grid
        }
        bool on = want != 0 && !grp.counter && nmpc::rti_block_two_phase_supported(p, g) && (long)g.grid * count >= 2 * resident;
        int tail = 0, lag = 0, count2 = 0;
        if (on) {
            tail = (int)std::ceil(h->tp_share * 1.08 * g.grid) + 3;
            if (env_tail) tail = atoi(env_tail);
            tail = tail < 1 ? 1 : (tail > g.grid ? g.grid : tail);
            const long unit = (long)g.grid + tail;
            lag = (int)((resident + resident / 8 + unit - 1) / unit);
            if (env_lag) lag = atoi(env_lag);
            lag = lag < 1 ? 1 : lag;
            int single = lag;
            if (env_single) single = atoi(env_single);
            single = single < 0 ? 0 : single;
            count2 = count - single;
            on = count2 >= 1;
        }
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (on) (void)hipStreamIsCapturing((hipStream_t)stream, &cap);
        int slot = -1;
        if (on) { // a queue region: the next of the ring whose last user is this stream or has finished
            const size_t need = (size_t)count2 * sizeof(int) * (16 + (size_t)g.grid * (1 + g.G)); // exits (a line each), reports, entries
            for (int tries = 0; tries < alore_nmpc_solver::kTpRing && slot < 0; ++tries) {
                const int i = (int)(h->tp_turn++ % alore_nmpc_solver::kTpRing);
                if (h->tp_used[i] && h->tp_stream[i] != (hipStream_t)stream && hipEventQuery(h->tp_ev[i]) != hipSuccess) continue;
                if (h->tp_cap[i] < need) {
                    if (cap != hipStreamCaptureStatusNone) continue; // no allocation inside a capture
                    if (h->tp_used[i]) (void)hipEventSynchronize(h->tp_ev[i]);
                    if (h->d_tp[i]) (void)hipFree(h->d_tp[i]);
                    h->d_tp[i] = nullptr; h->tp_cap[i] = 0; h->tp_used[i] = false;
                    const size_t want_bytes = need + need / 4;
                    if (hipMalloc((void**)&h->d_tp[i], want_bytes) != hipSuccess) { (void)hipGetLastError(); continue; }
                    if (hipMemset(h->d_tp[i], 0, want_bytes) != hipSuccess) { (void)hipFree(h->d_tp[i]); h->d_tp[i] = nullptr; continue; }
                    h->tp_cap[i] = want_bytes;
                }
                slot = i;
            }
            on = slot >= 0;
        }
        if (on) {
            grp.tp_count2 = count2;
            grp.tp_tail = tail;
            grp.tp_lag = lag;
            grp.tp_timeout = 20 * 1000 * 100; // 20 ms of the 100 MHz counter
            grp.tp_exits = reinterpret_cast<int*>(h->d_tp[slot]);
            grp.tp_cnt = grp.tp_exits + (size_t)count2 * 16;
            grp.tp_entries = grp.tp_cnt + (size_t)count2 * g.grid;
            if (cap == hipStreamCaptureStatusNone && !h->tp_rec_pending) { // the record: eager launches only, one in flight at a time
                std::memset(h->tp_rec, 0, sizeof(int) * 40);
                void* alias = nullptr;
                if (hipHostGetDevicePointer(&alias, h->tp_rec, 0) == hipSuccess) grp.tp_rec = (int*)alias;
            }
            h->tp_used[slot] = true;
            h->tp_stream[slot] = (hipStream_t)stream;
            h->tp_last = 1;
            h->tp_info[0] = count2; h->tp_info[1] = tail; h->tp_info[2] = lag;
        }
        h->tp_slot_of_launch = on ? slot : -1;
    }
    // XCD shares (see the kernel): grids of at least two residencies of the (4, 5) grid build.  The record of the previous such launch, if it
    // has finished, moves the speed estimates (damped); this launch leaves its own record unless the previous one is still in flight.
    // ALORE_NMPC_XCD_SHARES=0 (diagnostic): equal shares, as the hardware deals them.
    grp.xcd_on = 0;
    grp.xcd_end = nullptr;
    static const bool xcd_shares_on = !(getenv("ALORE_NMPC_XCD_SHARES") && atoi(getenv("ALORE_NMPC_XCD_SHARES")) == 0);
    const long total_items = (long)g.grid * count;
    // Round 6: only grids of at least `min_res` residencies (default 12).  A grid of five residencies ends when its last wavefronts end
    // wherever the shares put them, and shares fitted on a long grid cost the next short one 3 % (20 batches right after 200: 142 us
    // against 137.5 with equal shares, four runs each; repeated 20-batch grids: 132.4 - 134.2 with shares, 131.4 - 132.8 without); the
    // 200-batch grid keeps its 0.5 % (5.39 against 5.42 us per batch).  profiles/r06_contract_first_region.txt
    static const long xcd_min_res = getenv("ALORE_NMPC_XCD_MIN_RES") ? atol(getenv("ALORE_NMPC_XCD_MIN_RES")) : 12;
    if (xcd_shares_on && !grp.counter && grp.tp_count2 == 0 && g.L == 4 && g.RS == 5 && h->cfg.N == 20 && n_sqp == 1 && total_items >= xcd_min_res * 4 * h->n_cu && total_items >= 2L * 4 * h->n_cu && total_items < (1L << 28)) {
        bool record = true;
        if (h->xcd_pending) {
            if (hipEventQuery(h->xcd_ev) == hipSuccess) {
                const unsigned long long t0 = h->xcd_rec[32];
                double dur[8], mean = 0.0;
                bool ok = t0 != 0;
                for (int x = 0; x < 8 && ok; ++x) {
                    unsigned long long e = 0;
                    for (int i = 0; i < 4; ++i) e = h->xcd_rec[x * 4 + i] > e ? h->xcd_rec[x * 4 + i] : e;
                    ok = e > t0;
                    dur[x] = (double)(e - t0);
                    mean += dur[x] / 8.0;
                }
                if (ok && mean > 1000.0) { // 10 us of the 100 MHz counter: anything shorter says nothing
                    double norm = 0.0;
                    for (int x = 0; x < 8; ++x) { // an XCD that finished late is slower than its share assumed
                        double v = h->xcd_speed[x] * std::pow(mean / dur[x], 0.7);
                        h->xcd_speed[x] = v;
                        norm += v / 8.0;
                    }
                    for (int x = 0; x < 8; ++x) {
                        double v = h->xcd_speed[x] / norm;
                        h->xcd_speed[x] = v < 0.85 ? 0.85 : (v > 1.15 ? 1.15 : v);
                    }
                    ++h->xcd_updates;
                }
                h->xcd_pending = false;
            } else {
                record = false; // the record is still being written: equal-or-last shares, no new record
            }
        }
        double sum = 0.0;
        for (int x = 0; x < 8; ++x) sum += h->xcd_speed[x];
        long given = 0;
        double frac[8];
        for (int x = 0; x < 8; ++x) {
            const double want = (double)total_items * h->xcd_speed[x] / sum;
            grp.xcd_share[x] = (int)want;
            frac[x] = want - (double)grp.xcd_share[x];
            given += grp.xcd_share[x];
        }
        while (given < total_items) { // the remainder to the largest fractions
            int bx = 0;
            for (int x = 1; x < 8; ++x) bx = frac[x] > frac[bx] ? x : bx;
            ++grp.xcd_share[bx]; frac[bx] = -1.0; ++given;
        }
        while (given > total_items) { // rounding can only ever give too few; if it ever gave too many, the largest share gives them back
            int bx = 0;
            for (int x = 1; x < 8; ++x) bx = grp.xcd_share[x] > grp.xcd_share[bx] ? x : bx;
            --grp.xcd_share[bx]; --given;
        }
        int base = 0;
        for (int x = 0; x < 8; ++x) { grp.xcd_base[x] = base; base += grp.xcd_share[x]; }
        grp.xcd_on = base == total_items ? 1 : 0;
        if (record) { // never inside a stream capture: every replay would write the host record with no event to tell when
            hipStreamCaptureStatus capx = hipStreamCaptureStatusNone;
            (void)hipStreamIsCapturing((hipStream_t)stream, &capx);
            record = capx == hipStreamCaptureStatusNone;
        }
        if (record) {
            std::memset(h->xcd_rec, 0, sizeof(unsigned long long) * 40);
            void* alias = nullptr;
            if (hipHostGetDevicePointer(&alias, h->xcd_rec, 0) == hipSuccess) grp.xcd_end = (unsigned long long*)alias;
        }
    }
    static const char* trace_path = getenv("ALORE_NMPC_TRACE");
    grp.trace = nullptr;
    if (trace_path && grp.tp_count2 == 0 && g.L == 4 && g.RS == 5 && h->cfg.N == 20 && n_sqp == 1 && batches[0].kkt && batches[0].obj) {
        const size_t words = (size_t)g.grid * count * 8;
        long long* d_trace = nullptr;
        HIP_TRY(h, hipMalloc((void**)&d_trace, words * sizeof(long long)));
        grp.trace = d_trace;
        hipError_t e = hipMemsetAsync(d_trace, 0, words * sizeof(long long), (hipStream_t)stream);
        if (e == hipSuccess) e = nmpc::launch_rti_block_group(p, grp, g, (hipStream_t)stream);
        if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
        std::vector<long long> host(words);
        if (e == hipSuccess) e = hipMemcpy(host.data(), d_trace, words * sizeof(long long), hipMemcpyDeviceToHost);
        (void)hipFree(d_trace);
        if (e != hipSuccess) return fail(h, ALORE_NMPC_E_HIP, "rti_many: traced launch", e);
        static int trace_seq = 0; // one file per traced grid: <path>.<n>
        const std::string fname = std::string(trace_path) + "." + std::to_string(trace_seq++);
        if (FILE* f = std::fopen(fname.c_str(), "wb")) {
            const long long hdr[8] = {0x4543415254LL /* "TRACE" */, (long long)g.grid * count, count, g.grid, grp.stagger_blocks, grp.stagger_x1024, p.pg_steps, 0};
            std::fwrite(hdr, sizeof(long long), 8, f);
            std::fwrite(host.data(), sizeof(long long), host.size(), f);
            std::fclose(f);
        }
        h->last_geom = g;
        h->last_geom.grid = g.grid * count;
        h->have_geom = true;
        return ALORE_NMPC_OK;
    }
    // diagnostic (ALORE_NMPC_TP_TRACE=<file>, never under a stream capture): every workgroup of a two-phase grid leaves the real-time counter at
    // its start / inputs landed / end and its role; synchronous, one file <file>.<n> per grid (tools/tp_trace.py)
    static const char* tp_trace_path = getenv("ALORE_NMPC_TP_TRACE");
    if (tp_trace_path && grp.tp_count2 > 0) {
        const long long units = std::max((long long)count, (long long)grp.tp_count2 + grp.tp_lag);
        const size_t words = (size_t)(units * ((long long)g.grid + grp.tp_tail)) * 4;
        long long* d_trace = nullptr;
        HIP_TRY(h, hipMalloc((void**)&d_trace, words * sizeof(long long)));
        grp.tp_trace = d_trace;
        hipError_t e = hipMemsetAsync(d_trace, 0, words * sizeof(long long), (hipStream_t)stream);
        if (e == hipSuccess) e = nmpc::launch_rti_block_group(p, grp, g, (hipStream_t)stream);
        if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
        std::vector<long long> host(words);
        if (e == hipSuccess) e = hipMemcpy(host.data(), d_trace, words * sizeof(long long), hipMemcpyDeviceToHost);
        (void)hipFree(d_trace);
        if (e != hipSuccess) return fail(h, ALORE_NMPC_E_HIP, "rti_many: traced two-phase launch", e);
        static int tp_trace_seq = 0;
        const std::string fname = std::string(tp_trace_path) + "." + std::to_string(tp_trace_seq++);
        if (FILE* f = std::fopen(fname.c_str(), "wb")) {
            const long long hdr[8] = {0x5054LL, (long long)(words / 4), count, g.grid, grp.tp_count2, grp.tp_tail, grp.tp_lag, 0};
            std::fwrite(hdr, sizeof(long long), 8, f);
            std::fwrite(host.data(), sizeof(long long), host.size(), f);
            std::fclose(f);
        }
        if (grp.tp_rec && hipEventRecord(h->tp_rec_ev, (hipStream_t)stream) == hipSuccess) h->tp_rec_pending = true;
        h->last_geom = g;
        h->last_geom.grid = g.grid * count;
        h->have_geom = true;
        return ALORE_NMPC_OK;
    }
    HIP_TRY(h, nmpc::launch_rti_block_group(p, grp, g, (hipStream_t)stream));
    if (h->tp_slot_of_launch >= 0) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing((hipStream_t)stream, &cap);
        if (cap == hipStreamCaptureStatusNone) {
            (void)hipEventRecord(h->tp_ev[h->tp_slot_of_launch], (hipStream_t)stream);
            if (grp.tp_rec && hipEventRecord(h->tp_rec_ev, (hipStream_t)stream) == hipSuccess) h->tp_rec_pending = true;
        }
    }
    if (grp.xcd_end) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing((hipStream_t)stream, &cap);
        if (cap == hipStreamCaptureStatusNone && hipEventRecord(h->xcd_ev, (hipStream_t)stream) == hipSuccess) h->xcd_pending = true;
    }
    h->last_geom = g;
    h->last_geom.grid = g.grid * count;
    h->have_geom = true;
    return ALORE_NMPC_OK;
}

// Do the batches sit at constant strides -- batch i = batch 0 with every member pointer advanced by i * stride[member] bytes
// (the slots of one arena, the slices of one tensor per member)?  stride[] in the member order of alore_nmpc_batch.
bool constant_strides(const alore_nmpc_batch* batches, int count, long long* stride)
{
    if (count < 2) return false;
    const char* const* p0 = reinterpret_cast<const char* const*>(batches);
    const char* const* p1 = reinterpret_cast<const char* const*>(batches + 1);
    for (int m = 0; m < 15; ++m) {
        if ((p0[m] == nullptr) != (p1[m] == nullptr)) return false;
        stride[m] = p0[m] ? (long long)(p1[m] - p0[m]) : 0;
    }
    for (int i = 2; i < count; ++i) {
        const char* const* pi = reinterpret_cast<const char* const*>(batches + i);
        for (int m = 0; m < 15; ++m) {
            if ((pi[m] == nullptr) != (p0[m] == nullptr)) return false;
            if (p0[m] && pi[m] != p0[m] + (long long)i * stride[m]) return false;
        }
    }
    return true;
}

} // namespace

int alore_nmpc_rti(alore_nmpc_handle h, const alore_nmpc_batch* dev, int B, int n_sqp, void* stream)
{
    if (!h || !dev || B <= 0 || n_sqp < 1) return fail(h, ALORE_NMPC_E_INVALID, "rti: bad argument");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    return rti_one(h, dev, B, n_sqp, stream, 0);
}

int alore_nmpc_rti_many(alore_nmpc_handle h, const alore_nmpc_batch* batches, int count, int B, int n_sqp, void* stream)
{
    if (!h || !batches || count < 1 || B <= 0 || n_sqp < 1) return fail(h, ALORE_NMPC_E_INVALID, "rti_many: bad argument");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    for (int i = 0; i < count; ++i)
        if (!batch_complete(batches + i)) return fail(h, ALORE_NMPC_E_INVALID, "rti_many: batch has NULL members");
    // Independent batches (checked: address ranges) are kept in flight together.  Default: GROUPS -- up to GROUP_MAX batches
    // per grid of the stage-block kernel (one launch, one graph node, no ramp and drain between the batches; the lane
    // mapping is chosen for all problems of the grid).  alore_nmpc_set_many_mode(h, 1): STREAMS -- one launch per batch,
    // round-robin over `overlap` internal streams forked from and joined back into the caller's (the round-3 form; also
    // what batches that cannot use the stage-block kernel get).  A batch listed twice (successive iterations of the same
    // problems), overlap = 1 or per-launch timing keep the launches in order on `stream`.
    int ways = h->overlap;
    if (const char* e = std::getenv("ALORE_NMPC_OVERLAP")) ways = std::atoi(e);
    ways = ways < 1 ? 1 : (ways > 32 ? 32 : ways);
    if (ways > count) ways = count;
    if (ways > 1 && h->stamps) ways = 1;
    if (ways > 1 && h->timing && h->many_mode != 0) ways = 1; // per-launch timing of the streams mode: in order; the groups mode times its grid(s) as one
    if (ways > 1) {
        // the independence check (a sort of 15 x count address ranges) is remembered for the descriptor set it passed on: a host
        // that steps the same slots every tick pays for it once (alore_nmpc_rti_many_prepare: before the first tick)
        if (!known_independent(h, batches, count, B)) {
            if (batches_independent(h, batches, count, B)) remember_independent(h, batches, count, B);
            else ways = 1;
        }
    }
    if (ways == 1) {
        for (int i = 0; i < count; ++i) {
            const int rc = rti_one(h, batches + i, B, n_sqp, stream, 0);
            if (rc != ALORE_NMPC_OK) return rc;
        }
        return ALORE_NMPC_OK;
    }
    int mode = h->many_mode;
    if (const char* e = std::getenv("ALORE_NMPC_MANY")) mode = (e[0] == 's') ? 1 : 0; // diagnostic: "streams" / "groups"
    bool groups = mode == 0;
    if (groups) {
        nmpc::LaunchGeom g;
        groups = nmpc::block_geometry(B, h->cfg.N, h->cfg.lanes_per_problem & 0xff, h->lds_limit, h->n_cu, &g, B);
        for (int i = 0; i < count && groups; ++i)
            groups = block_eligible(h, batches + i) && (batches[i].kkt != nullptr) == (batches[0].kkt != nullptr) &&
                     (batches[i].obj != nullptr) == (batches[0].obj != nullptr);
    }
    hipStream_t main_s = (hipStream_t)stream;
    const long clampB = 0x7fffffffL;
    if (groups) {
        // Batches at constant strides (the slots of one arena): ONE grid for all of them, whatever their number.  Otherwise
        // equal groups of at most GROUP_MAX batches (descriptor table in the kernel arguments), one grid each, in order on the
        // caller's stream.  ALORE_NMPC_GROUP_STREAMS=2 (diagnostic): successive table groups alternate between the caller's
        // stream and one side stream so that the tail of one grid runs under the head of the next.
        long long stride[15];
        nmpc::LaunchGeom g1;
        (void)nmpc::block_geometry(B, h->cfg.N, h->cfg.lanes_per_problem & 0xff, h->lds_limit, h->n_cu, &g1, B);
        if (constant_strides(batches, count, stride) && (long long)g1.grid * count <= 0x7fffffffLL) {
            const long inflight = (long)B * count;
            // alore_nmpc_set_timing: HIP events on the launch stream directly around the grid (alore_nmpc_get_launch_info: last_kernel_ms is
            // then the duration of the ONE grid that served the `count` batches)
            if (h->timing) HIP_TRY(h, hipEventRecord(h->ev0, main_s));
            const int rc = rti_group(h, batches, count, B, n_sqp, stream, (int)(inflight > clampB ? clampB : inflight), stride);
            if (h->timing && rc == ALORE_NMPC_OK) {
                HIP_TRY(h, hipEventRecord(h->ev1, main_s));
                h->timed_pending = true;
            }
            return rc;
        }
        static const bool alternate = getenv("ALORE_NMPC_GROUP_STREAMS") && atoi(getenv("ALORE_NMPC_GROUP_STREAMS")) == 2;
        const int n_groups = (count + nmpc::GROUP_MAX - 1) / nmpc::GROUP_MAX;
        const int per = (count + n_groups - 1) / n_groups;
        const bool two = alternate && n_groups > 1;
        const long inflight = (long)B * per * (two ? 2 : 1);
        const int Bf = (int)(inflight > clampB ? clampB : inflight);
        // alore_nmpc_set_timing: the events go on the caller's stream around ALL grids of the call (in front of the fork, behind the join)
        if (h->timing) HIP_TRY(h, hipEventRecord(h->ev0, main_s));
        if (two) {
            HIP_TRY(h, hipEventRecord(h->fork_ev, main_s));
            HIP_TRY(h, hipStreamWaitEvent(h->side[0], h->fork_ev, 0));
        }
        int rc = ALORE_NMPC_OK;
        for (int gi = 0, first = 0; first < count && rc == ALORE_NMPC_OK; ++gi, first += per) {
            const int n = (count - first < per) ? count - first : per;
            rc = rti_group(h, batches + first, n, B, n_sqp, (two && (gi & 1)) ? (void*)h->side[0] : (void*)main_s, Bf, nullptr);
        }
        if (two) { // join even after a failed launch: the side stream must not stay forked (an open capture would be lost)
            const hipError_t e1 = hipEventRecord(h->join_ev[0], h->side[0]);
            const hipError_t e2 = hipStreamWaitEvent(main_s, h->join_ev[0], 0);
            if (rc == ALORE_NMPC_OK && e1 != hipSuccess) return fail(h, ALORE_NMPC_E_HIP, "rti_many: join", e1);
            if (rc == ALORE_NMPC_OK && e2 != hipSuccess) return fail(h, ALORE_NMPC_E_HIP, "rti_many: join", e2);
        }
        if (h->timing && rc == ALORE_NMPC_OK) {
            HIP_TRY(h, hipEventRecord(h->ev1, main_s));
            h->timed_pending = true;
        }
        return rc;
    }
    // STREAMS.  With timing on the launches of this path never fork (rti_one records the handle's ONE pair of events: two of them on
    // different streams would pair events of different launches): in order on the caller's stream.
    if (h->timing) {
        for (int i = 0; i < count; ++i) {
            const int rc1 = rti_one(h, batches + i, B, n_sqp, stream, 0);
            if (rc1 != ALORE_NMPC_OK) return rc1;
        }
        return ALORE_NMPC_OK;
    }
    HIP_TRY(h, hipEventRecord(h->fork_ev, main_s));
    int forked = 0;
    int rc = ALORE_NMPC_OK;
    for (int w = 1; w < ways && rc == ALORE_NMPC_OK; ++w) {
        const hipError_t e = hipStreamWaitEvent(h->side[w - 1], h->fork_ev, 0);
        if (e != hipSuccess) rc = fail(h, ALORE_NMPC_E_HIP, "rti_many: fork", e);
        else forked = w;
    }
    const long inflight = (long)B * ways;
    const int Bf = (int)(inflight > clampB ? clampB : inflight); // the automatic lane mapping packs for all of them
    for (int i = 0; i < count && rc == ALORE_NMPC_OK; ++i) {
        const int w = i % ways;
        rc = rti_one(h, batches + i, B, n_sqp, w == 0 ? (void*)main_s : (void*)h->side[w - 1], Bf);
    }
    for (int w = 1; w <= forked; ++w) { // join every forked stream, whatever happened above
        const hipError_t e1 = hipEventRecord(h->join_ev[w - 1], h->side[w - 1]);
        const hipError_t e2 = hipStreamWaitEvent(main_s, h->join_ev[w - 1], 0);
        if (rc == ALORE_NMPC_OK && e1 != hipSuccess) rc = fail(h, ALORE_NMPC_E_HIP, "rti_many: join", e1);
        if (rc == ALORE_NMPC_OK && e2 != hipSuccess) rc = fail(h, ALORE_NMPC_E_HIP, "rti_many: join", e2);
    }
    return rc;
}

int alore_nmpc_synchronize(alore_nmpc_handle h, void* stream)
{
    if (!h) return ALORE_NMPC_E_INVALID;
    HIP_TRY(h, hipStreamSynchronize((hipStream_t)stream));
    return ALORE_NMPC_OK;
}

int alore_nmpc_rti_many_prepare(alore_nmpc_handle h, const alore_nmpc_batch* batches, int count, int B)
{
    if (!h || !batches || count < 1 || B <= 0) return fail(h, ALORE_NMPC_E_INVALID, "rti_many_prepare: bad argument");
    for (int i = 0; i < count; ++i)
        if (!batch_complete(batches + i)) return fail(h, ALORE_NMPC_E_INVALID, "rti_many_prepare: batch has NULL members");
    if (known_independent(h, batches, count, B)) return ALORE_NMPC_OK;
    if (!batches_independent(h, batches, count, B))
        return fail(h, ALORE_NMPC_E_INVALID, "rti_many_prepare: the batches overlap (alore_nmpc_rti_many would run them in order)");
    remember_independent(h, batches, count, B);
    return ALORE_NMPC_OK;
}

int alore_nmpc_set_problem_mask(alore_nmpc_handle h, const unsigned char* mask)
{
    if (!h) return ALORE_NMPC_E_INVALID;
    h->mask = mask;
    return ALORE_NMPC_OK;
}

int alore_nmpc_set_many_mode(alore_nmpc_handle h, int mode)
{
    if (!h || mode < 0 || mode > 1) return fail(h, ALORE_NMPC_E_INVALID, "set_many_mode: 0 (groups) or 1 (streams)");
    h->many_mode = mode;
    return ALORE_NMPC_OK;
}

int alore_nmpc_set_two_phase(alore_nmpc_handle h, int mode)
{
    if (!h || mode < -1 || mode > 1) return fail(h, ALORE_NMPC_E_INVALID, "set_two_phase: -1 (automatic), 0 (never) or 1 (wherever the build exists)");
    h->two_phase = mode;
    return ALORE_NMPC_OK;
}

int alore_nmpc_get_two_phase_info(alore_nmpc_handle h, alore_nmpc_two_phase_info* out)
{
    if (!h || !out) return ALORE_NMPC_E_INVALID;
    out->last_grid_two_phase = h->tp_last;
    out->two_phase_batches = h->tp_info[0];
    out->tail_workgroups_per_batch = h->tp_info[1];
    out->lag_units = h->tp_info[2];
    out->tail_share = (float)h->tp_share;
    return ALORE_NMPC_OK;
}

int alore_nmpc_set_launch_overlap(alore_nmpc_handle h, int ways)
{
    if (!h || ways < 1 || ways > 32) return fail(h, ALORE_NMPC_E_INVALID, "set_launch_overlap: ways must be 1 .. 32");
    h->overlap = ways;
    return ALORE_NMPC_OK;
}

int alore_nmpc_linearize(alore_nmpc_handle h, const alore_nmpc_batch* dev, int B, const alore_nmpc_lin_out* out,
                         void* stream)
{
    if (!h || !dev || !out || B <= 0 || !dev->x || !dev->u || !dev->od)
        return fail(h, ALORE_NMPC_E_INVALID, "linearize: bad argument");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, nmpc::launch_linearize(*dev, B, h->cfg.N, h->cfg.dt, *out, (hipStream_t)stream));
    return ALORE_NMPC_OK;
}

int alore_nmpc_condense(alore_nmpc_handle h, const alore_nmpc_batch* dev, int B, const alore_nmpc_dense_qp_data* out, void* stream)
{
    if (!h || !dev || !out || B <= 0 || !out->H || !out->g || !out->lb || !out->ub || !dev->x || !dev->u || !dev->od || !dev->y || !dev->yN ||
        !dev->W || !dev->WN || !dev->x0 || !dev->lbValues || !dev->ubValues)
        return fail(h, ALORE_NMPC_E_INVALID, "condense: bad argument");
    if (h->cfg.N > 64 || (long)nmpc::condense_lds_bytes(h->cfg.N) > (long)h->lds_limit)
        return fail(h, ALORE_NMPC_E_UNSUPPORTED, "condense: horizons up to 64 (the E blocks of a problem live in LDS) and within this device's LDS per workgroup");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, nmpc::launch_condense(*dev, h->lin_x, h->lin_u, B, h->cfg.N, h->cfg.dt, h->shared, out->H, out->g, out->lb, out->ub, (hipStream_t)stream));
    return ALORE_NMPC_OK;
}

int alore_nmpc_dense_qp(alore_nmpc_handle h, int B, int n, const alore_nmpc_dense_qp_data* qp, float* x, float* y, int* status, int* n_iter,
                        void* stream)
{
    if (!h || !qp || B <= 0 || n < 1 || n > 128 || !qp->H || !qp->g || !qp->lb || !qp->ub || !x || !y || !status || !n_iter)
        return fail(h, ALORE_NMPC_E_INVALID, "dense_qp: bad argument (n <= 128)");
    if ((long)nmpc::dense_qp_lds_bytes(n) > (long)h->lds_limit)
        return fail(h, ALORE_NMPC_E_UNSUPPORTED, "dense_qp: the factor of an n x n Hessian does not fit this device's LDS per workgroup (n = 128 needs 68 KB)");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, nmpc::launch_dense_qp(B, n, qp->H, qp->g, qp->lb, qp->ub, x, y, status, n_iter, h->cfg.max_as_iter, (hipStream_t)stream));
    return ALORE_NMPC_OK;
}

int alore_nmpc_forward_simulate(alore_nmpc_handle h, const alore_nmpc_batch* dev, int B, void* stream)
{
    if (!h || !dev || B <= 0 || !dev->x || !dev->u || !dev->od)
        return fail(h, ALORE_NMPC_E_INVALID, "forward_simulate: bad argument");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, nmpc::launch_forward_simulate(*dev, B, h->cfg.N, h->cfg.dt, (hipStream_t)stream));
    return ALORE_NMPC_OK;
}

int alore_nmpc_shift(alore_nmpc_handle h, const alore_nmpc_batch* dev, int B, int strategy, const float* xEnd,
                     const float* uEnd, void* stream)
{
    if (!h || !dev || B <= 0 || !dev->x || !dev->u || !dev->od)
        return fail(h, ALORE_NMPC_E_INVALID, "shift: bad argument");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, nmpc::launch_shift(*dev, B, h->cfg.N, h->cfg.dt, strategy, xEnd, uEnd, (hipStream_t)stream));
    return ALORE_NMPC_OK;
}

int alore_nmpc_refs_init(alore_nmpc_handle h, int B, int max_pieces, int max_checkpoints)
{
    if (!h || B <= 0 || max_pieces <= 0 || max_checkpoints <= 0 || h->refs.dur)
        return fail(h, ALORE_NMPC_E_INVALID, "refs_init: bad argument or already initialised");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    h->refs.P = max_pieces;
    h->refs.C = max_checkpoints;
    h->refs_B = B;
    struct Want { void** p; size_t bytes; };
    const Want want[] = {
        {(void**)&h->refs.dur, sizeof(double) * B * max_pieces},
        {(void**)&h->refs.coef, sizeof(double) * B * max_pieces * 12},
        {(void**)&h->refs.ckpt, sizeof(double) * B * max_checkpoints * 2},
        {(void**)&h->refs.meta, sizeof(double) * B * 8},
        {(void**)&h->d_est, sizeof(double) * B * 6}, // pose [B][3], then ICR [B][3]: one block, one copy per tick
        {(void**)&h->d_psi, sizeof(double) * B * (h->cfg.N + 1)},
        {(void**)&h->d_goal, sizeof(int) * B},
    };
    hipError_t e = hipSuccess;
    for (const Want& w : want) { // everything zeroed: slots without a trajectory read as invalid / not at goal
        e = hipMalloc(w.p, w.bytes);
        if (e == hipSuccess) e = hipMemset(*w.p, 0, w.bytes);
        if (e != hipSuccess) break;
    }
    if (e != hipSuccess) { // a retry must see "not initialised"
        for (const Want& w : want) {
            if (*w.p) (void)hipFree(*w.p);
            *w.p = nullptr;
        }
        h->refs_B = 0;
        return fail(h, ALORE_NMPC_E_NOMEM, "refs_init: hipMalloc", e);
    }
    h->d_icr = h->d_est + (size_t)B * 3;
    return ALORE_NMPC_OK;
}

int alore_nmpc_refs_set_trajectory(alore_nmpc_handle h, int robot, int n_pieces, const double* durations,
                                   const double* coeffs, int n_ckpt, const double* ckpt_xy, double start_time,
                                   double state_seq_res, double xv, void* stream)
{
    if (!h || !h->refs.dur || robot < 0 || robot >= h->refs_B || n_pieces <= 0 || n_pieces > h->refs.P || n_ckpt <= 0 ||
        n_ckpt > h->refs.C || !durations || !coeffs || !ckpt_xy || !(state_seq_res > 0.0))
        return fail(h, ALORE_NMPC_E_INVALID, "refs_set_trajectory: bad argument");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    double total = 0.0;
    for (int i = 0; i < n_pieces; ++i) total += durations[i];
    const double meta[8] = {start_time, total, xv, state_seq_res, (double)n_pieces, (double)n_ckpt, 1.0, 0.0};
    HIP_TRY(h, hipMemcpyAsync(h->refs.dur + (size_t)robot * h->refs.P, durations, sizeof(double) * n_pieces, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(h->refs.coef + (size_t)robot * h->refs.P * 12, coeffs, sizeof(double) * n_pieces * 12, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(h->refs.ckpt + (size_t)robot * h->refs.C * 2, ckpt_xy, sizeof(double) * n_ckpt * 2, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(h->refs.meta + (size_t)robot * 8, meta, sizeof(meta), hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipStreamSynchronize(s)); // the host buffers (and `meta`) may go away after return
    return ALORE_NMPC_OK;
}

namespace {
struct PolyLayout { // byte offsets of the packed message arrays of one chunk
    size_t robot, n_pieces, inner, t_pts, pva, start, icr, t0, end;
};
PolyLayout poly_layout(int chunk, int P)
{
    PolyLayout L;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 15) & ~size_t(15); return at; };
    L.robot = take(sizeof(int) * chunk);
    L.n_pieces = take(sizeof(int) * chunk);
    L.inner = take(sizeof(double) * chunk * (P > 1 ? P - 1 : 1) * 2);
    L.t_pts = take(sizeof(double) * chunk * P);
    L.pva = take(sizeof(double) * chunk * 12);
    L.start = take(sizeof(double) * chunk * 3);
    L.icr = take(sizeof(double) * chunk * 3);
    L.t0 = take(sizeof(double) * chunk);
    L.end = o;
    return L;
}
} // namespace

int alore_nmpc_refs_set_polynomes(alore_nmpc_handle h, int count, const int* robots, const alore_polynome* msgs,
                                  double state_seq_res, int integral_res_int, void* stream)
{
    if (!h || !h->refs.dur || count < 0 || (count > 0 && (!robots || !msgs)) || !(state_seq_res > 0.0) || integral_res_int < 1)
        return fail(h, ALORE_NMPC_E_INVALID, "refs_set_polynomes: bad argument");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    const int P = h->refs.P, CH = alore_nmpc_solver::kPolyChunk, Pi = (P > 1 ? P - 1 : 1);
    const PolyLayout Lmax = poly_layout(CH, P);
    if (!h->d_poly) {
        HIP_TRY(h, hipMalloc((void**)&h->d_poly, Lmax.end));
        HIP_TRY(h, hipMalloc((void**)&h->d_knot, sizeof(double) * CH * 2 * nmpc::traj_ws_doubles(P)));
        HIP_TRY(h, hipMalloc((void**)&h->d_panels, sizeof(int) * CH));
        HIP_TRY(h, hipMalloc((void**)&h->d_overflow, sizeof(int)));
    }
    // messages per round: as many as the staging allows (one upload, three kernels and one wait per round), fewer when the
    // Simpson increments of a round ([C x res_int x 2] doubles per message) would pass 512 MB
    const size_t inc_per_msg = (size_t)h->refs.C * integral_res_int * 2;
    int chunk = (int)(((size_t)512 << 20) / (inc_per_msg * sizeof(double)));
    chunk = chunk < 64 ? 64 : (chunk > CH ? CH : chunk);
    if (chunk > count) chunk = count < 1 ? 1 : count;
    const size_t need = (size_t)chunk * inc_per_msg;
    if (need > h->inc_doubles) {
        if (h->d_inc) (void)hipFree(h->d_inc);
        h->d_inc = nullptr;
        HIP_TRY(h, hipMalloc((void**)&h->d_inc, sizeof(double) * need));
        h->inc_doubles = need;
    }
    HIP_TRY(h, hipMemsetAsync(h->d_overflow, 0, sizeof(int), s));
    const PolyLayout L = poly_layout(chunk, P); // staging laid out for the round size in use: a single message stays a small copy
    std::vector<char> pack(L.end);
    for (int base = 0; base < count; base += chunk) {
        const int n = (count - base < chunk) ? count - base : chunk;
        std::fill(pack.begin(), pack.end(), 0);
        int* p_robot = reinterpret_cast<int*>(pack.data() + L.robot);
        int* p_np = reinterpret_cast<int*>(pack.data() + L.n_pieces);
        double* p_inner = reinterpret_cast<double*>(pack.data() + L.inner);
        double* p_t = reinterpret_cast<double*>(pack.data() + L.t_pts);
        double* p_pva = reinterpret_cast<double*>(pack.data() + L.pva);
        double* p_start = reinterpret_cast<double*>(pack.data() + L.start);
        double* p_icr = reinterpret_cast<double*>(pack.data() + L.icr);
        double* p_t0 = reinterpret_cast<double*>(pack.data() + L.t0);
        for (int i = 0; i < n; ++i) {
            const alore_polynome& m = msgs[base + i];
            if (robots[base + i] < 0 || robots[base + i] >= h->refs_B)
                return fail(h, ALORE_NMPC_E_INVALID, "refs_set_polynomes: robot index out of range");
            p_robot[i] = robots[base + i];
            p_np[i] = m.n_pieces; // > P is reported by the kernel through the overflow flag
            const int M = (m.n_pieces >= 1 && m.n_pieces <= P) ? m.n_pieces : 0;
            if (M > 0 && (!m.t_pts || (M > 1 && !m.innerpoints)))
                return fail(h, ALORE_NMPC_E_INVALID, "refs_set_polynomes: null array in a message");
            for (int k = 0; k < M; ++k) p_t[(size_t)i * P + k] = m.t_pts[k];
            for (int k = 0; k < 2 * (M - 1); ++k) p_inner[(size_t)i * Pi * 2 + k] = m.innerpoints[k];
            for (int d = 0; d < 2; ++d) {
                p_pva[i * 12 + d] = m.init_p[d]; p_pva[i * 12 + 2 + d] = m.init_v[d]; p_pva[i * 12 + 4 + d] = m.init_a[d];
                p_pva[i * 12 + 6 + d] = m.tail_p[d]; p_pva[i * 12 + 8 + d] = m.tail_v[d]; p_pva[i * 12 + 10 + d] = m.tail_a[d];
            }
            for (int k = 0; k < 3; ++k) { p_start[i * 3 + k] = m.start_position[k]; p_icr[i * 3 + k] = m.ICR[k]; }
            p_t0[i] = m.traj_start_time;
        }
        HIP_TRY(h, hipMemcpyAsync(h->d_poly, pack.data(), L.end, hipMemcpyHostToDevice, s));
        nmpc::PolyBatch pb;
        pb.robot = reinterpret_cast<const int*>(h->d_poly + L.robot);
        pb.n_pieces = reinterpret_cast<const int*>(h->d_poly + L.n_pieces);
        pb.inner = reinterpret_cast<const double*>(h->d_poly + L.inner);
        pb.t_pts = reinterpret_cast<const double*>(h->d_poly + L.t_pts);
        pb.pva = reinterpret_cast<const double*>(h->d_poly + L.pva);
        pb.start = reinterpret_cast<const double*>(h->d_poly + L.start);
        pb.icr = reinterpret_cast<const double*>(h->d_poly + L.icr);
        pb.t0 = reinterpret_cast<const double*>(h->d_poly + L.t0);
        pb.P = P;
        HIP_TRY(h, nmpc::launch_traj_build(h->refs, pb, n, state_seq_res, integral_res_int, h->d_knot, h->d_panels,
                                           h->d_inc, h->d_overflow, s));
        HIP_TRY(h, hipStreamSynchronize(s)); // `pack` is reused by the next chunk
    }
    int ov = 0;
    HIP_TRY(h, hipMemcpyAsync(&ov, h->d_overflow, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    if (ov & 1) return fail(h, ALORE_NMPC_E_INVALID, "refs_set_polynomes: a message has more pieces than max_pieces (or none)");
    if (ov & 2) return fail(h, ALORE_NMPC_E_INVALID, "refs_set_polynomes: a trajectory needs more checkpoints than max_checkpoints");
    return ALORE_NMPC_OK;
}

int alore_nmpc_refs_set_from_backend(alore_nmpc_handle h, const void* view, int count, double traj_start_time, double xv,
                                     double state_seq_res, int integral_res_int, void* stream)
{
    const alore_backend_device_view* v = static_cast<const alore_backend_device_view*>(view);
    if (!h || !h->refs.dur || !v || count < 1 || count > h->refs_B || !(state_seq_res > 0.0) || integral_res_int < 1)
        return fail(h, ALORE_NMPC_E_INVALID, "refs_set_from_backend: bad argument");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    if (!h->d_overflow) HIP_TRY(h, hipMalloc((void**)&h->d_overflow, sizeof(int)));
    if ((size_t)count > h->panels_cap) {
        if (h->d_panels_be) (void)hipFree(h->d_panels_be);
        h->d_panels_be = nullptr;
        HIP_TRY(h, hipMalloc((void**)&h->d_panels_be, sizeof(int) * count));
        h->panels_cap = count;
    }
    const size_t need = (size_t)count * h->refs.C * integral_res_int * 2;
    if (need > h->inc_doubles) {
        if (h->d_inc) (void)hipFree(h->d_inc);
        h->d_inc = nullptr;
        h->inc_doubles = 0;
        HIP_TRY(h, hipMalloc((void**)&h->d_inc, sizeof(double) * need));
        h->inc_doubles = need;
    }
    HIP_TRY(h, hipMemsetAsync(h->d_overflow, 0, sizeof(int), s));
    const nmpc::BackendView bv{v->max_pieces, v->n_pieces, v->T, v->coef, v->start_xytheta, v->ok};
    HIP_TRY(h, nmpc::launch_traj_from_backend(h->refs, bv, count, traj_start_time, state_seq_res, integral_res_int, xv, h->d_panels_be,
                                              h->d_inc, h->d_overflow, s));
    int ov = 0;
    HIP_TRY(h, hipMemcpyAsync(&ov, h->d_overflow, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    if (ov & 1) return fail(h, ALORE_NMPC_E_INVALID, "refs_set_from_backend: a plan has more pieces than max_pieces");
    if (ov & 2) return fail(h, ALORE_NMPC_E_INVALID, "refs_set_from_backend: a trajectory needs more checkpoints than max_checkpoints");
    return ALORE_NMPC_OK;
}

int alore_nmpc_refs_download(alore_nmpc_handle h, int robot, double* meta8, double* durations, double* coeffs, double* ckpt_xy)
{
    if (!h || !h->refs.dur || robot < 0 || robot >= h->refs_B) return fail(h, ALORE_NMPC_E_INVALID, "refs_download: bad argument");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipDeviceSynchronize());
    if (meta8) HIP_TRY(h, hipMemcpy(meta8, h->refs.meta + (size_t)robot * 8, sizeof(double) * 8, hipMemcpyDeviceToHost));
    if (durations) HIP_TRY(h, hipMemcpy(durations, h->refs.dur + (size_t)robot * h->refs.P, sizeof(double) * h->refs.P, hipMemcpyDeviceToHost));
    if (coeffs) HIP_TRY(h, hipMemcpy(coeffs, h->refs.coef + (size_t)robot * h->refs.P * 12, sizeof(double) * h->refs.P * 12, hipMemcpyDeviceToHost));
    if (ckpt_xy) HIP_TRY(h, hipMemcpy(ckpt_xy, h->refs.ckpt + (size_t)robot * h->refs.C * 2, sizeof(double) * h->refs.C * 2, hipMemcpyDeviceToHost));
    return ALORE_NMPC_OK;
}

int alore_nmpc_refs_sample(alore_nmpc_handle h, const alore_nmpc_batch* dev, int B, double now, const double* est,
                           const double* icr, int do_smooth, int* at_goal, void* stream)
{
    if (!h || !h->refs.dur || !dev || B <= 0 || B > h->refs_B || !est || !icr || !dev->y || !dev->yN || !dev->od || !dev->x0)
        return fail(h, ALORE_NMPC_E_INVALID, "refs_sample: bad argument");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    // pose and ICR from pageable memory go through pinned staging (two buffers in turn, so that the call never waits for its
    // predecessor's copies): a hipMemcpyAsync out of pageable memory blocks the host for the whole transfer, twice per tick
    const size_t bytes = sizeof(double) * (size_t)B * 3;
    const double *se = est, *si = icr;
    int turn = -1;
    if (!host_is_pinned(est) || !host_is_pinned(icr)) {
        turn = h->pose_turn++ & 1;
        if (h->pose_ev[turn]) HIP_TRY(h, hipEventSynchronize(h->pose_ev[turn]));
        if (int rc = grow_stage(h, h->pose_stage[turn], h->pose_cap[turn], 2 * bytes)) return rc;
        std::memcpy(h->pose_stage[turn], est, bytes);
        std::memcpy(h->pose_stage[turn] + bytes, icr, bytes);
        se = reinterpret_cast<const double*>(h->pose_stage[turn]);
        si = reinterpret_cast<const double*>(h->pose_stage[turn] + bytes);
    }
    if (turn >= 0 && B == h->refs_B) { // staged back to back, stored back to back: one copy
        HIP_TRY(h, hipMemcpyAsync(h->d_est, se, 2 * bytes, hipMemcpyHostToDevice, s));
    } else {
        HIP_TRY(h, hipMemcpyAsync(h->d_est, se, bytes, hipMemcpyHostToDevice, s));
        HIP_TRY(h, hipMemcpyAsync(h->d_icr, si, bytes, hipMemcpyHostToDevice, s));
    }
    if (turn >= 0) {
        if (!h->pose_ev[turn]) HIP_TRY(h, hipEventCreateWithFlags(&h->pose_ev[turn], hipEventDisableTiming));
        HIP_TRY(h, hipEventRecord(h->pose_ev[turn], s));
    }
    HIP_TRY(h, nmpc::launch_ref_sample(h->refs, *dev, B, h->cfg.N, (double)h->cfg.dt, now, h->d_est, h->d_icr, h->d_goal,
                                       h->d_psi, do_smooth, s));
    if (at_goal) {
        HIP_TRY(h, hipMemcpyAsync(at_goal, h->d_goal, sizeof(int) * B, hipMemcpyDeviceToHost, s));
        HIP_TRY(h, hipStreamSynchronize(s));
    }
    return ALORE_NMPC_OK;
}

int alore_nmpc_refs_eval(alore_nmpc_handle h, int B, double now, double* out, void* stream)
{
    if (!h || !h->refs.dur || B <= 0 || B > h->refs_B || !out) return fail(h, ALORE_NMPC_E_INVALID, "refs_eval: bad argument");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    if (!h->d_flat) HIP_TRY(h, hipMalloc((void**)&h->d_flat, sizeof(double) * h->refs_B * 4));
    HIP_TRY(h, nmpc::launch_ref_eval(h->refs, B, now, h->d_flat, s));
    HIP_TRY(h, hipMemcpyAsync(out, h->d_flat, sizeof(double) * B * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    return ALORE_NMPC_OK;
}

int alore_nmpc_plant_init(alore_nmpc_handle h, const alore_plant_params* p)
{
    if (!h || !h->refs.dur || !p || p->substeps < 1 || !(p->state_propa_period > 0.0) || !(p->pose_pub_period > 0.0))
        return fail(h, ALORE_NMPC_E_INVALID, "plant_init: needs refs_init first and positive periods");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if (!h->d_vw) HIP_TRY(h, hipMalloc((void**)&h->d_vw, sizeof(double) * h->refs_B * 2));
    HIP_TRY(h, hipMemset(h->d_vw, 0, sizeof(double) * h->refs_B * 2));
    h->plant.max_a = p->max_acc; h->plant.max_domega = p->max_domega;
    h->plant.pose_pub_period = p->pose_pub_period; h->plant.propa_period = p->state_propa_period;
    h->plant.substeps = p->substeps;
    // closed_loop_run's second reference buffer and the float64 headings of the walk sampled ahead (nothing is allocated inside a run)
    if (h->cl_B < h->refs_B) {
        const int Bc = h->refs_B, N = h->cfg.N;
        if (h->cl_y) (void)hipFree(h->cl_y);
        if (h->cl_yN) (void)hipFree(h->cl_yN);
        h->cl_y = nullptr; h->cl_yN = nullptr; h->cl_B = 0;
        for (int i = 0; i < 2; ++i) { if (h->cl_psi[i]) (void)hipFree(h->cl_psi[i]); h->cl_psi[i] = nullptr; }
        HIP_TRY(h, hipMalloc(&h->cl_y, sizeof(float) * (size_t)Bc * N * 5));
        HIP_TRY(h, hipMalloc(&h->cl_yN, sizeof(float) * (size_t)Bc * 3));
        for (int i = 0; i < 2; ++i) HIP_TRY(h, hipMalloc(&h->cl_psi[i], sizeof(double) * (size_t)Bc * (N + 1)));
        h->cl_B = Bc;
    }
    h->has_plant = true;
    return ALORE_NMPC_OK;
}

int alore_nmpc_plant_set_state(alore_nmpc_handle h, int B, const double* pose, const double* vw, const double* icr, void* stream)
{
    if (!h || !h->has_plant || B <= 0 || B > h->refs_B || !pose || !icr)
        return fail(h, ALORE_NMPC_E_INVALID, "plant_set_state: bad argument");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(h, hipMemcpyAsync(h->d_est, pose, sizeof(double) * B * 3, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(h->d_icr, icr, sizeof(double) * B * 3, hipMemcpyHostToDevice, s));
    if (vw) HIP_TRY(h, hipMemcpyAsync(h->d_vw, vw, sizeof(double) * B * 2, hipMemcpyHostToDevice, s));
    else HIP_TRY(h, hipMemsetAsync(h->d_vw, 0, sizeof(double) * B * 2, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    return ALORE_NMPC_OK;
}

int alore_nmpc_plant_get_state(alore_nmpc_handle h, int B, double* pose, double* vw, int* at_goal, void* stream)
{
    if (!h || !h->has_plant || B <= 0 || B > h->refs_B) return fail(h, ALORE_NMPC_E_INVALID, "plant_get_state: bad argument");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    if (pose) HIP_TRY(h, hipMemcpyAsync(pose, h->d_est, sizeof(double) * B * 3, hipMemcpyDeviceToHost, s));
    if (vw) HIP_TRY(h, hipMemcpyAsync(vw, h->d_vw, sizeof(double) * B * 2, hipMemcpyDeviceToHost, s));
    if (at_goal) HIP_TRY(h, hipMemcpyAsync(at_goal, h->d_goal, sizeof(int) * B, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    return ALORE_NMPC_OK;
}

int alore_nmpc_closed_loop_reset(alore_nmpc_handle h, const alore_nmpc_batch* dev, int B, const unsigned char* mask, void* stream)
{
    if (!h || !h->has_plant || !dev || !dev->x || !dev->u || B <= 0 || B > h->refs_B)
        return fail(h, ALORE_NMPC_E_INVALID, "closed_loop_reset: bad argument");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    unsigned char* d_mask = nullptr;
    if (mask) {
        if (!h->d_mask) HIP_TRY(h, hipMalloc((void**)&h->d_mask, (size_t)h->refs_B));
        HIP_TRY(h, hipMemcpyAsync(h->d_mask, mask, (size_t)B, hipMemcpyHostToDevice, s));
        HIP_TRY(h, hipStreamSynchronize(s)); // `mask` may go away after return
        d_mask = h->d_mask;
    }
    HIP_TRY(h, nmpc::launch_iterate_reset(*dev, B, h->cfg.N, h->d_est, d_mask, s));
    return ALORE_NMPC_OK;
}

int alore_nmpc_closed_loop_tick(alore_nmpc_handle h, const alore_nmpc_batch* dev, int B, double now, int delay_num, void* stream)
{
    if (!h || !h->has_plant || !dev || B <= 0 || B > h->refs_B || delay_num < 0)
        return fail(h, ALORE_NMPC_E_INVALID, "closed_loop_tick: bad argument");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    const int N = h->cfg.N;
    // CmdCallback: references from the measured pose, one real-time iteration, command = input column delay_num
    HIP_TRY(h, nmpc::launch_ref_sample(h->refs, *dev, B, N, (double)h->cfg.dt, now, h->d_est, h->d_icr, h->d_goal, h->d_psi, 1, s));
    const int rc = alore_nmpc_rti(h, dev, B, 1, stream);
    if (rc != ALORE_NMPC_OK) return rc;
    HIP_TRY(h, nmpc::launch_plant(*dev, B, N, delay_num < N ? delay_num : N - 1, h->d_icr, h->d_goal, h->d_est, h->d_vw, h->plant, s));
    return ALORE_NMPC_OK;
}

int alore_nmpc_refs_at_goal(alore_nmpc_handle h, int B, int* at_goal, void* stream)
{
    if (!h || !h->refs.dur || B <= 0 || B > h->refs_B || !at_goal) return fail(h, ALORE_NMPC_E_INVALID, "refs_at_goal: bad argument");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipMemcpyAsync(at_goal, h->d_goal, sizeof(int) * B, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return ALORE_NMPC_OK;
}

namespace {
static __global__ void pack_input_column_kernel(const float* u, const int* status, int B, int N, int node, float* cmd, int* st)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    cmd[2 * b] = u[((size_t)b * N + node) * 2];
    cmd[2 * b + 1] = u[((size_t)b * N + node) * 2 + 1];
    if (st) st[b] = status[b];
}
} // namespace

int alore_nmpc_input_column(alore_nmpc_handle h, const alore_nmpc_batch* dev, int B, int node, float* cmd, int* status, void* stream)
{
    if (!h || !dev || !dev->u || B <= 0 || node < 0 || node >= h->cfg.N || !cmd || (status && !dev->status))
        return fail(h, ALORE_NMPC_E_INVALID, "input_column: bad argument");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    const size_t need = (size_t)B * 12;
    if (int rc = grow_stage(h, h->stage_down, h->stage_down_cap, need)) return rc;
    // the pack kernel writes the 12 bytes per problem straight into the pinned slab (device alias of the host pointer)
    void* dalias = nullptr;
    HIP_TRY(h, hipHostGetDevicePointer(&dalias, h->stage_down, 0));
    float* dc = (float*)dalias;
    int* ds = (int*)(dc + (size_t)B * 2);
    hipLaunchKernelGGL(pack_input_column_kernel, dim3((B + 255) / 256), dim3(256), 0, s, dev->u, dev->status, B, h->cfg.N, node, dc,
                       status ? ds : nullptr);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipStreamSynchronize(s));
    std::memcpy(cmd, h->stage_down, sizeof(float) * B * 2);
    if (status) std::memcpy(status, h->stage_down + sizeof(float) * B * 2, sizeof(int) * B);
    return ALORE_NMPC_OK;
}

int alore_nmpc_set_shared_members(alore_nmpc_handle h, unsigned mask)
{
    if (!h || (mask & ~(unsigned)(ALORE_NMPC_SHARED_W | ALORE_NMPC_SHARED_BOUNDS | ALORE_NMPC_SHARED_OD)))
        return fail(h, ALORE_NMPC_E_INVALID, "set_shared_members: unknown bit");
    h->shared = mask;
    return ALORE_NMPC_OK;
}

int alore_nmpc_closed_loop_run(alore_nmpc_handle h, const alore_nmpc_batch* dev, int B, double t0, double dt_tick, int n_ticks,
                               int delay_num, void* stream)
{
    if (n_ticks < 0 || !(dt_tick > 0.0)) return fail(h, ALORE_NMPC_E_INVALID, "closed_loop_run: bad argument");
    static const bool serial = [] { const char* e = getenv("ALORE_NMPC_CLOSED_LOOP_SERIAL"); return e && atoi(e) != 0; }();
    const bool ahead = !serial && n_ticks >= 3 && h && h->has_plant && dev && dev->y && dev->yN && dev->x0 && B > 0 && B <= h->refs_B &&
                       delay_num >= 0 && nmpc::ref_sample_ahead_supported(h->cfg.N);
    if (!ahead) {
        for (int t = 0; t < n_ticks; ++t) {
            const int rc = alore_nmpc_closed_loop_tick(h, dev, B, t0 + dt_tick * t, delay_num, stream);
            if (rc != ALORE_NMPC_OK) return rc;
        }
        return ALORE_NMPC_OK;
    }
    // The chain of a tick is sampler -> solve -> plant, and the sampler needs the pose only for x0 and for the turns that the heading
    // walk starts from (smooth_yaw's first step): everything else of tick t + 1 is sampled by extra workgroups of the grid that
    // solves tick t (rti_block_sampler_kernel), into the other of two reference buffers; the plant step of tick t writes x0 and
    // shifts the headings, and runs in front of the solve of tick t + 1 in that solve's grid.  One launch per tick on the caller's
    // stream.  (A second stream for the sampler was built first: its two cross-stream events per tick cost what the overlap saved.)
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    const int N = h->cfg.N, node = delay_num < N ? delay_num : N - 1;
    if (h->cl_B < B) return fail(h, ALORE_NMPC_E_INVALID, "closed_loop_run: more robots than plant_init allocated for");
    alore_nmpc_batch buf[2] = {*dev, *dev};
    buf[1].y = h->cl_y;
    buf[1].yN = h->cl_yN;
    // both buffers start as the caller's references: a robot without a trajectory keeps them, whichever buffer its tick reads
    HIP_TRY(h, hipMemcpyAsync(h->cl_y, dev->y, sizeof(float) * (size_t)B * N * 5, hipMemcpyDeviceToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(h->cl_yN, dev->yN, sizeof(float) * (size_t)B * 3, hipMemcpyDeviceToDevice, s));
    // tick 0 is sampled whole (od, x0 from the current pose)
    HIP_TRY(h, nmpc::launch_ref_sample(h->refs, *dev, B, N, (double)h->cfg.dt, t0, h->d_est, h->d_icr, h->d_goal, h->d_psi, 1, s));
    // per tick ONE launch: [plant step of tick t - 1 -> solve of tick t] beside [sampler of tick t + 1]
    auto plant_of = [&](int t, bool more) {
        const int cur = t & 1, nxt = cur ^ 1;
        nmpc::PlantAhead a{};
        a.u = buf[cur].u;
        a.x0 = const_cast<float*>(buf[nxt].x0);
        a.y = const_cast<float*>(buf[nxt].y);
        a.yN = const_cast<float*>(buf[nxt].yN);
        a.meta = h->refs.meta;
        a.icr = h->d_icr;
        a.at_goal = h->d_goal;
        a.pose = h->d_est;
        a.vw = h->d_vw;
        a.psi_rel = more ? h->cl_psi[nxt] : nullptr;
        a.p = h->plant;
        a.now = t0 + dt_tick * t;
        a.B = B; a.N = N; a.node = node;
        return a;
    };
    for (int t = 0; t < n_ticks; ++t) {
        const bool more = t + 1 < n_ticks;
        const int cur = t & 1, nxt = cur ^ 1;
        nmpc::AheadSampler sa{};
        sa.store = h->refs;
        sa.y = const_cast<float*>(buf[nxt].y);
        sa.yN = const_cast<float*>(buf[nxt].yN);
        sa.icr = h->d_icr;
        sa.psi_rel = h->cl_psi[nxt];
        sa.dt = (double)h->cfg.dt;
        sa.now = t0 + dt_tick * (t + 1);
        sa.B = more ? B : 0; // the last tick has nothing to sample for
        sa.N = N;
        bool sampled = false;
        if (!batch_complete(&buf[cur])) return fail(h, ALORE_NMPC_E_INVALID, "closed_loop_run: batch has NULL members");
        const nmpc::PlantAhead prev = plant_of(t > 0 ? t - 1 : 0, true);
        const int rc = rti_one(h, &buf[cur], B, 1, stream, 0, &sa, &sampled, t > 0 ? &prev : nullptr);
        if (rc != ALORE_NMPC_OK) return rc;
        if (more && !sampled) // no build of this mapping carries the sampler: its own launch, in the chain
            HIP_TRY(h, nmpc::launch_ref_sample_ahead(h->refs, buf[nxt], B, N, sa.dt, sa.now, h->d_icr, h->cl_psi[nxt], s));
    }
    HIP_TRY(h, nmpc::launch_plant_ahead(plant_of(n_ticks - 1, false), s)); // the plant step of the last tick
    if ((n_ticks - 1) & 1) { // the last tick read the internal buffer: the caller's y / yN are those of the last tick afterwards, as in a tick-by-tick run
        HIP_TRY(h, hipMemcpyAsync(const_cast<float*>(dev->y), h->cl_y, sizeof(float) * (size_t)B * N * 5, hipMemcpyDeviceToDevice, s));
        HIP_TRY(h, hipMemcpyAsync(const_cast<float*>(dev->yN), h->cl_yN, sizeof(float) * (size_t)B * 3, hipMemcpyDeviceToDevice, s));
    }
    return ALORE_NMPC_OK;
}

int alore_nmpc_set_linearization_point(alore_nmpc_handle h, const float* x_lin, const float* u_lin)
{
    if (!h || ((x_lin == nullptr) != (u_lin == nullptr))) return fail(h, ALORE_NMPC_E_INVALID, "linearization point: bad argument");
    h->lin_x = x_lin;
    h->lin_u = u_lin;
    return ALORE_NMPC_OK;
}

// internal (not in the public header): the trajectory store of a handle, for the LTV-MPC reference sampler (ltv_mpc.hip)
int alore_nmpc_internal_refstore(void* nmpc_handle, nmpc::RefStore* out, int* capacity, int* device)
{
    alore_nmpc_handle h = static_cast<alore_nmpc_handle>(nmpc_handle);
    if (!h || !h->refs.dur || !out) return -1;
    *out = h->refs;
    if (capacity) *capacity = h->refs_B;
    if (device) *device = h->cfg.device;
    return 0;
}

int alore_nmpc_set_timing(alore_nmpc_handle h, int enable)
{
    if (!h) return ALORE_NMPC_E_INVALID;
    h->timing = enable != 0;
    h->timed_pending = false;
    h->last_ms = -1.0f;
    return ALORE_NMPC_OK;
}

int alore_nmpc_get_launch_info(alore_nmpc_handle h, alore_nmpc_launch_info* out)
{
    if (!h || !out) return ALORE_NMPC_E_INVALID;
    if (!h->have_geom) return fail(h, ALORE_NMPC_E_INVALID, "launch_info: no launch yet");
    if (h->timing && h->timed_pending) {
        HIP_TRY(h, hipEventSynchronize(h->ev1));
        HIP_TRY(h, hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
        h->timed_pending = false;
    }
    out->lanes_per_problem = h->last_geom.L | (h->last_geom.block ? 0x100 : 0);
    out->problems_per_block = h->last_geom.G * h->last_geom.wpb;
    out->threads_per_block = h->last_geom.threads;
    out->grid = h->last_geom.grid;
    out->lds_bytes_per_block = (int)h->last_geom.lds_bytes;
    out->last_kernel_ms = h->timing ? h->last_ms : -1.0f;
    return ALORE_NMPC_OK;
}

} // extern "C"
