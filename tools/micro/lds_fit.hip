// How many 64-thread workgroups of a given dynamic LDS size does a CU of this GPU hold?  (the runtime's occupancy query, and a
// direct count: workgroups that spin until told, counted per CU through their XCC / SE / CU ids)
//   hipcc --offload-arch=gfx950 -O2 tools/micro/lds_fit.hip -o tools/micro/lds_fit
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

__global__ void hold(unsigned long long* out, long long ticks)
{
    extern __shared__ char lds[];
    lds[threadIdx.x] = 1;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        out[2 * blockIdx.x] = ((unsigned long long)(xcc & 0xF) << 32) | hw;
        out[2 * blockIdx.x + 1] = t0;
    }
}

int main()
{
    unsigned long long* d;
    const int nb = 256 * 12;
    hipMalloc(&d, nb * 16);
    std::vector<unsigned long long> h(2 * nb);
    for (int lds : {16384, 18000, 19968, 20096, 20224, 20480, 21504, 22016, 22528, 23040, 26368, 26880, 27136, 27392, 31744, 32256, 32768}) {
        hipFuncSetAttribute((const void*)hold, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        int q = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, hold, 64, lds);
        hipLaunchKernelGGL(hold, dim3(nb), dim3(64), lds, 0, d, 20000LL); // 200 us: everything that fits is resident together
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, nb * 16, hipMemcpyDeviceToHost);
        // workgroups that started within the first 50 us, per CU
        unsigned long long tmin = ~0ull;
        for (int i = 0; i < nb; ++i) tmin = h[2 * i + 1] < tmin ? h[2 * i + 1] : tmin;
        std::map<unsigned long long, int> per;
        for (int i = 0; i < nb; ++i)
            if (h[2 * i + 1] - tmin < 5000) {
                const unsigned long long id = h[2 * i], hw = id & 0xFFFFFFFF, xcc = id >> 32;
                per[(xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF)]++;
            }
        int mx = 0, mn = 1 << 30;
        for (auto& kv : per) { mx = kv.second > mx ? kv.second : mx; mn = kv.second < mn ? kv.second : mn; }
        printf("LDS %6d B per workgroup: occupancy query %2d, resident per CU at once: min %d max %d over %zu CUs\n", lds, q, mn, mx, per.size());
    }
    return 0;
}
