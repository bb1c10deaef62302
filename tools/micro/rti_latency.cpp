// rti_latency.cpp -- what a C++ controller sees of one synchronous real-time iteration for ONE robot through the C ABI:
// enqueue (alore_nmpc_rti returns), kernel + wake-up (hipStreamSynchronize returns), over 2000 cold-start solves of a fixed problem.
//   g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include tools/micro/rti_latency.cpp -L alore_legged_manipulator_amd -lalore_nmpc \
//       -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/alore_legged_manipulator_amd -o tools/micro/rti_latency
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <vector>

#include "alore_nmpc.h"

int main(int argc, char** argv)
{
    const int B = argc > 1 ? std::atoi(argv[1]) : 1, N = 20;
    alore_nmpc_handle h = nullptr;
    alore_nmpc_config cfg{N, 0.01f, 0, 0, 0, -1};
    if (alore_nmpc_create(&cfg, &h) != ALORE_NMPC_OK) { std::fprintf(stderr, "create failed\n"); return 1; }
    alore_nmpc_batch dev{};
    alore_nmpc_batch_alloc(h, B, &dev);
    const size_t nx = (size_t)B * (N + 1) * 3, nu = (size_t)B * N * 2;
    std::vector<float> x(nx, 0.f), u(nu, 0.f), od(nx), y((size_t)B * N * 5), yN((size_t)B * 3), W((size_t)B * N * 25, 0.f), WN((size_t)B * 9, 0.f), x0((size_t)B * 3, 0.f),
        lb(nu, -8.f), ub(nu, 8.f), dual(nu, 0.f);
    for (int b = 0; b < B; ++b) {
        for (int k = 0; k <= N; ++k) { od[((size_t)b * (N + 1) + k) * 3] = 0.1f; od[((size_t)b * (N + 1) + k) * 3 + 1] = -0.3f; od[((size_t)b * (N + 1) + k) * 3 + 2] = 0.3f; }
        for (int k = 0; k < N; ++k) {
            float* yk = &y[((size_t)b * N + k) * 5];
            yk[0] = 0.02f * (k + 1); yk[1] = 0.01f * (k + 1); yk[2] = 0.05f; yk[3] = 5.f; yk[4] = 5.f; // a reference that asks for more than the bounds give
            float* Wk = &W[((size_t)b * N + k) * 25];
            Wk[0] = Wk[6] = 10.f; Wk[12] = 0.5f; Wk[18] = Wk[24] = 0.1f;
        }
        yN[b * 3] = 0.4f; yN[b * 3 + 1] = 0.2f; yN[b * 3 + 2] = 0.05f;
        WN[b * 9] = WN[b * 9 + 4] = 10.f; WN[b * 9 + 8] = 0.5f;
    }
    alore_nmpc_batch host{};
    host.x = x.data(); host.u = u.data(); host.od = od.data(); host.y = y.data(); host.yN = yN.data(); host.W = W.data(); host.WN = WN.data(); host.x0 = x0.data();
    host.lbValues = lb.data(); host.ubValues = ub.data(); host.dual = dual.data();
    alore_nmpc_batch_upload(h, &dev, &host, B, nullptr);
    hipDeviceSynchronize();
    std::vector<double> enq, tot;
    std::vector<int> st(B);
    for (int it = 0; it < 2100; ++it) {
        hipMemcpy(dev.x, x.data(), nx * 4, hipMemcpyHostToDevice); // the cold start again
        hipMemcpy(dev.u, u.data(), nu * 4, hipMemcpyHostToDevice);
        hipMemcpy(dev.dual, dual.data(), nu * 4, hipMemcpyHostToDevice);
        hipDeviceSynchronize();
        const auto t0 = std::chrono::steady_clock::now();
        alore_nmpc_rti(h, &dev, B, 1, nullptr);
        const auto t1 = std::chrono::steady_clock::now();
        hipStreamSynchronize(nullptr);
        const auto t2 = std::chrono::steady_clock::now();
        if (it >= 100) { enq.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count()); tot.push_back(std::chrono::duration<double, std::micro>(t2 - t0).count()); }
    }
    hipMemcpy(st.data(), dev.status, B * 4, hipMemcpyDeviceToHost);
    std::vector<int> ni(B);
    hipMemcpy(ni.data(), dev.n_iter, B * 4, hipMemcpyDeviceToHost);
    std::sort(enq.begin(), enq.end()); std::sort(tot.begin(), tot.end());
    std::printf("B = %d: enqueue p50 %.2f us; enqueue + kernel + stream synchronisation p50 %.2f  p99 %.2f us; status %d, sweeps %d\n", B, enq[enq.size() / 2], tot[tot.size() / 2],
                tot[tot.size() * 99 / 100], st[0], ni[0]);
    alore_nmpc_batch_free(h, &dev);
    alore_nmpc_destroy(h);
    return 0;
}
