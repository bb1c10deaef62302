// ltv_mpc.h -- what the two translation units of the LTV-MPC share (internal).
//   ltv_mpc.hip       the get_cmd kernels: built WITHOUT NaN / Inf / signed-zero semantics (the stage algebra multiplies by
//                     structural zeros that fold away under them); they test their inputs for non-finite values by bit pattern
//   ltv_mpc_capi.hip  the C ABI of include/alore_ltv_mpc.h and the reference-sampling kernels: default floating-point semantics
#ifndef ALORE_LTV_MPC_INTERNAL_H
#define ALORE_LTV_MPC_INTERNAL_H

#include <hip/hip_runtime.h>

#include "../../include/alore_ltv_mpc.h"

namespace ltv {

constexpr int MAXT = 64;
enum : int { FREE = 0, BOX_LO = 1, BOX_HI = 2, RATE_LO = 3, RATE_HI = 4 };
constexpr int NF = 38; // doubles per stage record
constexpr int REC_STRIDE = NF * 4 + 10; // lanes kernel, LDS doubles per stage (4 robots): consecutive lanes 8 dwords apart mod 128

struct Dev {
    alore_ltv_config c;
    int B;               // robots in this launch
    int stride;          // robot stride of the interleaved arrays (= max_robots)
    const double* now;   // [B][3]
    const double* xref;  // [B][T][3]
    const double* dref;  // [B][T][2]
    double* output;      // [B][T][2]  in/out (previous output -> new output)
    double* buff;        // [B][d][2]  in/out
    double* xopt;        // [B][T+1][3]
    double* ws;          // [T][NF][stride]
    int* st;             // [T][2][stride] working set (kept between calls: warm start)
    int* sweeps;         // [B]
    int* status;         // [B]
    double* cmd;         // [B][2] the command a tick publishes: column delay_num of the output
    double* cmd_host;    // optional: the same, written straight into pinned host memory (alore_ltv_tick: no copy back)
    int* status_host;    // optional, with cmd_host
    int n_relin, reset;
    long long* stamps;   // diagnostic (ALORE_LTV_STAMPS=1): cycles of robot 0 in rollout / backward / forward / rest
};

// status of a robot (alore_ltv_results / _commands / _tick): 0 solved, 1 sweep cap reached, 2 a measured state, reference or
// stored previous output of the robot is NaN or Inf -- nothing is solved for it, its command is zero, its stored state stays
constexpr int STATUS_OK = 0, STATUS_SWEEP_CAP = 1, STATUS_NON_FINITE = 2;

// is the float64 value stored at p NaN or +-Inf?  Decided on the BITS, read from memory as an integer: holds in a file built
// with -fno-honor-nans / -fno-honor-infinities, where a test on a floating-point VALUE (a comparison, even a bit cast of it)
// may be folded away
__host__ __device__ inline bool non_finite_at(const double* p)
{
    const unsigned long long u = *reinterpret_cast<const volatile unsigned long long*>(p); // volatile: never merged with the float64 load of the same address
    return ((u >> 52) & 0x7ffull) == 0x7ffull;
}

// enqueue getCmd for d.B robots (lanes kernel unless thread_kernel); raises the LDS limit of the long-horizon build once
hipError_t launch_get_cmd(const Dev& d, bool thread_kernel, hipStream_t s);

} // namespace ltv
#endif
