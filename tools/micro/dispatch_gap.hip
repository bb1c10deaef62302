// dispatch_gap.hip -- how long a SIMD stays empty between one workgroup and the next, by what the workgroup holds.
// One-wavefront workgroups that spin for `spin` ticks of the 100 MHz counter and record start / end / HW_ID; the host prints, per
// configuration, the median gap between the end of a wavefront and the start of the next one on the same SIMD.
//   hipcc --offload-arch=gfx950 -O2 dispatch_gap.hip -o dispatch_gap && ./dispatch_gap
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>

struct Big { long long pad[380]; }; // 3 KB of kernel arguments, like RtiGroup

template <bool REGS512, bool SCRATCH, bool BIGARG>
__global__ __launch_bounds__(64) void spin_kernel(long long* out, int spin, int idx, Big big)
{
    extern __shared__ float lds[];
    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
    // how long the first read of a kernel argument takes for a new wavefront: `spin` arrives by a scalar load from the
    // kernel-argument segment; the second stamp cannot be taken before it is there
    int spin_now = spin;
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(spin_now));
    const long long t0b = (long long)__builtin_amdgcn_s_memrealtime() + (spin_now == 0x7fffffff ? 1 : 0);
    const long long hw = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
    if (REGS512) {
        asm volatile("v_mov_b32 v255, 0" ::: "v255");
        asm volatile("v_accvgpr_write_b32 a255, v255" ::: "a255");
    }
    int acc = 0;
    if (SCRATCH) {
        volatile int buf[24];
        for (int i = 0; i < 24; ++i) buf[i] = i + idx;
        acc = buf[(threadIdx.x + idx) % 24];
    }
    if (BIGARG) acc += (int)big.pad[idx & 255];
    lds[threadIdx.x] = (float)acc;
    while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < spin) __builtin_amdgcn_s_sleep(4);
    const long long t1 = (long long)__builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[(size_t)blockIdx.x * 4 + 0] = t0;
        out[(size_t)blockIdx.x * 4 + 1] = t1;
        out[(size_t)blockIdx.x * 4 + 2] = hw;
        out[(size_t)blockIdx.x * 4 + 3] = t0b - t0 + ((long long)lds[threadIdx.x] == 12345678 ? 1 : 0);
    }
}

template <bool R, bool S, bool A>
static void run(const char* name, int lds_bytes, int grid, int spin)
{
    long long* d = nullptr;
    hipMalloc((void**)&d, (size_t)grid * 4 * sizeof(long long));
    Big big = {};
    auto k = spin_kernel<R, S, A>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k, dim3(grid), dim3(64), lds_bytes, 0, d, spin, 0, big);
        hipDeviceSynchronize();
    }
    std::vector<long long> h((size_t)grid * 4);
    hipMemcpy(h.data(), d, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    hipFree(d);
    std::map<long long, std::vector<std::pair<long long, long long>>> by;
    long long tmin = h[0], tmax = h[1];
    for (int i = 0; i < grid; ++i) {
        const long long hw = h[(size_t)i * 4 + 2];
        const long long key = ((hw >> 32) << 20) | (hw & 0xFFFF0) >> 4; // xcc | se, sh, cu, simd
        by[key].push_back({h[(size_t)i * 4], h[(size_t)i * 4 + 1]});
        tmin = std::min(tmin, h[(size_t)i * 4]);
        tmax = std::max(tmax, h[(size_t)i * 4 + 1]);
    }
    std::vector<double> gaps;
    size_t most = 0;
    for (auto& kv : by) {
        auto& v = kv.second;
        std::sort(v.begin(), v.end());
        most = std::max(most, v.size());
        for (size_t i = 1; i < v.size(); ++i)
            if (v[i].first >= v[i - 1].second) gaps.push_back((v[i].first - v[i - 1].second) * 0.01);
    }
    std::sort(gaps.begin(), gaps.end());
    std::vector<double> arg;
    for (int i = 0; i < grid; ++i) arg.push_back(h[(size_t)i * 4 + 3] * 0.01);
    std::sort(arg.begin(), arg.end());
    auto pct = [&](double p) { return gaps.empty() ? 0.0 : gaps[(size_t)(p * (gaps.size() - 1))]; };
    std::printf("%-44s lds %6d B: SIMDs %4zu (max %zu wg each), grid %7.1f us, gap p10 %5.2f  p50 %5.2f  p90 %5.2f us (n = %zu); first kernel-argument read p10 %5.2f p50 %5.2f p90 %5.2f us\n", name, lds_bytes,
                by.size(), most, (tmax - tmin) * 0.01, pct(0.1), pct(0.5), pct(0.9), gaps.size(), arg[arg.size() / 10], arg[arg.size() / 2], arg[arg.size() * 9 / 10]);
}

int main()
{
    const int grid = 5120, spin = 2000; // 20 us
    run<true, false, true>("512 regs, no scratch, 3 KB args", 39936, grid, spin);
    run<true, true, true>("512 regs, scratch, 3 KB args", 39936, grid, spin);
    run<true, false, false>("512 regs, no scratch, small args use", 39936, grid, spin);
    run<false, false, true>("few regs, no scratch, 3 KB args", 39936, grid, spin);
    run<false, true, true>("few regs, scratch, 3 KB args", 39936, grid, spin);
    run<true, false, true>("512 regs, no scratch, 3 KB args", 1024, grid, spin);
    run<false, false, true>("few regs, no scratch, 3 KB args", 1024, grid, spin);
    run<true, true, true>("512 regs, scratch, 3 KB args", 1024, grid, spin);
    return 0;
}
