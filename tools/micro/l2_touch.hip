// Does touching one dword per cache line bring the line into the XCD's L2 so that a full read a little later hits?
//   hipcc --offload-arch=gfx950 -O2 tools/micro/l2_touch.hip -o tools/micro/l2_touch
// One wavefront per workgroup, 64 KB regions of a 4 GB buffer, every region used once (cold in L2 and in the last-level cache).
// Per region: [optional touch pass with stride S bytes] -> wait -> spin `gap` us -> full read (16 B per lane, contiguous) timed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k(const char* buf, size_t region, int stride, int gap_ticks, int other_wg_touches, unsigned long long* out, float* sinkp)
{
    const int wg = blockIdx.x, lane = threadIdx.x;
    // the region this workgroup reads; when other_wg_touches, the touch is done by workgroup wg - 8 (same XCD: workgroups go round-robin over 8)
    const char* mine = buf + (size_t)wg * region;
    float acc = 0.f;
    if (stride > 0) {
        const char* tgt = other_wg_touches ? buf + (size_t)(wg + 8) * region : mine;
        for (size_t off = (size_t)lane * stride; off < region; off += (size_t)64 * stride) acc += *(const volatile float*)(tgt + off);
    }
    asm volatile("s_waitcnt vmcnt(0)");
    const unsigned long long ts = __builtin_amdgcn_s_memrealtime();
    while ((long long)(__builtin_amdgcn_s_memrealtime() - ts) < gap_ticks) __builtin_amdgcn_s_sleep(2);
    if (other_wg_touches && wg < 8) return; // nobody touched these
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float4 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = *(const float4*)(mine + ((size_t)i * 64 + lane) * 16); // 16 KB per round ...
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += v[i].x + v[i].w;
    float4 w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = *(const float4*)(mine + 16384 + ((size_t)i * 64 + lane) * 16);
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += w[i].x + w[i].w;
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) out[wg] = t1 - t0;
    if (acc == 123.456f) *sinkp = acc;
}

int main()
{
    const size_t region = 65536, total = (size_t)4 << 30;
    char* buf; hipMalloc(&buf, total); hipMemset(buf, 1, total);
    unsigned long long* out; hipMalloc(&out, 8 * 65536);
    float* sinkp; hipMalloc(&sinkp, 4);
    const int nwg = 2048; // 2048 x 64 KB = 128 MB per launch; launches walk through the buffer
    size_t cursor = 0;
    std::vector<unsigned long long> h(nwg);
    auto run = [&](const char* what, int stride, int gap_us, int other) {
        double sum = 0; int n = 0;
        for (int rep = 0; rep < 3; ++rep) {
            if (cursor + (size_t)(nwg + 8) * region > total) cursor = 0;
            hipLaunchKernelGGL(k, dim3(nwg), dim3(64), 0, 0, buf + cursor, region, stride, gap_us * 100, other, out, sinkp);
            hipDeviceSynchronize();
            cursor += (size_t)(nwg + 8) * region;
            hipMemcpy(h.data(), out, nwg * 8, hipMemcpyDeviceToHost);
            if (rep == 0) continue;
            for (int i = other ? 8 : 0; i < nwg; ++i) { sum += h[i] * 0.01; ++n; }
        }
        printf("%-58s full read of 32 KB: %.2f us\n", what, sum / n);
    };
    run("no touch", 0, 2, 0);
    run("touch 1 dword / 128 B, 2 us before", 128, 2, 0);
    run("touch 1 dword / 64 B, 2 us before", 64, 2, 0);
    run("touch 1 dword / 32 B, 2 us before", 32, 2, 0);
    run("touch 1 dword / 128 B, 10 us before", 128, 10, 0);
    run("touch by the workgroup 8 before (same XCD), 128 B, 2 us", 128, 2, 1);
    run("touch by the workgroup 8 before (same XCD), 64 B, 2 us", 64, 2, 1);
    return 0;
}
