// nmpc_batch.cpp -- a C++ host driving the batched NMPC through the C ABI only (include/alore_nmpc.h), no Python:
// create -> device batch -> upload (pinned staging) -> one real-time iteration for B problems -> download.
// Reads the batch members from a flat binary file and writes x, u, status, kkt to another one, so that a test can
// compare the run with the same batch solved through the ctypes binding (tests/test_cpp_host.py).
//
//   g++ -std=c++17 -Iinclude examples/nmpc_batch.cpp -Lalore_legged_manipulator_amd -lalore_nmpc \
//       -Wl,-rpath,$PWD/alore_legged_manipulator_amd -o nmpc_batch
//   ./nmpc_batch in.bin out.bin
//
// in.bin : int32 B, int32 N, then float32 x[B][(N+1)*3] u[B][N*2] od[B][(N+1)*3] y[B][N*5] yN[B][3] W[B][N*25] WN[B][9]
//          x0[B][3] lbValues[B][N*2] ubValues[B][N*2] dual[B][N*2]
// out.bin: float32 x, u; int32 status; float32 kkt
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "alore_nmpc.h"

#define CHECK(call)                                                                                  \
    do {                                                                                             \
        const int rc_ = (call);                                                                      \
        if (rc_ != ALORE_NMPC_OK) {                                                                  \
            std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, h ? alore_nmpc_last_error(h) : "no handle"); \
            return 2;                                                                                \
        }                                                                                            \
    } while (0)

int main(int argc, char** argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 1; }
    alore_nmpc_handle h = nullptr;
    FILE* fi = std::fopen(argv[1], "rb");
    if (!fi) { std::perror(argv[1]); return 1; }
    int B = 0, N = 0;
    if (std::fread(&B, 4, 1, fi) != 1 || std::fread(&N, 4, 1, fi) != 1 || B <= 0 || N <= 0) return 1;
    const size_t nx = (size_t)B * (N + 1) * 3, nu = (size_t)B * N * 2;
    struct Member { size_t n; std::vector<float> v; };
    Member x{nx}, u{nu}, od{nx}, y{(size_t)B * N * 5}, yN{(size_t)B * 3}, W{(size_t)B * N * 25}, WN{(size_t)B * 9}, x0{(size_t)B * 3},
        lb{nu}, ub{nu}, dual{nu};
    for (Member* m : {&x, &u, &od, &y, &yN, &W, &WN, &x0, &lb, &ub, &dual}) {
        m->v.resize(m->n);
        if (std::fread(m->v.data(), 4, m->n, fi) != m->n) { std::fprintf(stderr, "short input file\n"); return 1; }
    }
    std::fclose(fi);

    alore_nmpc_config cfg{N, 0.01f, /*device*/ 0, /*max_as_iter*/ 0, /*lanes_per_problem*/ 0, /*warm_start_steps*/ -1};
    CHECK(alore_nmpc_create(&cfg, &h));
    alore_nmpc_batch dev{};
    CHECK(alore_nmpc_batch_alloc(h, B, &dev));
    alore_nmpc_batch host{};
    host.x = x.v.data(); host.u = u.v.data(); host.od = od.v.data(); host.y = y.v.data(); host.yN = yN.v.data();
    host.W = W.v.data(); host.WN = WN.v.data(); host.x0 = x0.v.data(); host.lbValues = lb.v.data(); host.ubValues = ub.v.data();
    host.dual = dual.v.data();
    CHECK(alore_nmpc_batch_upload(h, &dev, &host, B, nullptr));
    CHECK(alore_nmpc_rti(h, &dev, B, 1, nullptr));
    std::vector<int> status(B);
    std::vector<float> kkt(B);
    alore_nmpc_batch back{};
    back.x = x.v.data(); back.u = u.v.data(); back.status = status.data(); back.kkt = kkt.data();
    CHECK(alore_nmpc_batch_download(h, &dev, &back, B, nullptr));

    FILE* fo = std::fopen(argv[2], "wb");
    if (!fo) { std::perror(argv[2]); return 1; }
    std::fwrite(x.v.data(), 4, nx, fo);
    std::fwrite(u.v.data(), 4, nu, fo);
    std::fwrite(status.data(), 4, B, fo);
    std::fwrite(kkt.data(), 4, B, fo);
    std::fclose(fo);
    int solved = 0;
    for (int s : status) solved += s == 0;
    std::printf("B=%d N=%d solved=%d\n", B, N, solved);
    CHECK(alore_nmpc_batch_free(h, &dev));
    CHECK(alore_nmpc_destroy(h));
    return solved == B ? 0 : 3;
}
