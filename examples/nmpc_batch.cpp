// nmpc_batch.cpp -- a C++ host driving the batched NMPC through the C ABI only (include/alore_nmpc.h), no Python:
// create -> device batch -> upload (pinned staging) -> one real-time iteration for B problems -> download.
// Reads the batch members from a flat binary file and writes x, u, status, kkt to another one, so that a test can
// compare the run with the same batch solved through the ctypes binding (tests/test_cpp_host.py).
//
//   g++ -std=c++17 -Iinclude examples/nmpc_batch.cpp -Lalore_legged_manipulator_amd -lalore_nmpc \
//       -Wl,-rpath,$PWD/alore_legged_manipulator_amd -o nmpc_batch
//   ./nmpc_batch in.bin out.bin [slots]
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

    // Optional third argument K: the same batch in K independent device slots, all stepped by ONE alore_nmpc_rti_many call
    // (launches kept in flight on forked streams, joined into the caller's stream); every slot must return the bits of the
    // single launch above.  The lane mapping is pinned for this part: the automatic choice packs for the launches in flight,
    // and different mappings agree to rounding only.
    if (argc > 3) {
        const int K = std::atoi(argv[3]);
        if (K < 2 || K > 64) { std::fprintf(stderr, "slots must be 2 .. 64\n"); return 1; }
        alore_nmpc_launch_info info{};
        CHECK(alore_nmpc_get_launch_info(h, &info));
        alore_nmpc_handle h2 = nullptr;
        alore_nmpc_config cfg2 = cfg;
        cfg2.lanes_per_problem = info.lanes_per_problem;
        {
            const int rc_ = alore_nmpc_create(&cfg2, &h2);
            if (rc_ != ALORE_NMPC_OK) { std::fprintf(stderr, "second handle: %d\n", rc_); return 2; }
        }
        std::vector<alore_nmpc_batch> slots(K);
        std::vector<float> x_in(nx), u_in(nu);
        {   // the inputs again (x, u were overwritten by the download above)
            FILE* f2 = std::fopen(argv[1], "rb");
            if (!f2 || std::fseek(f2, 8, SEEK_SET) != 0 || std::fread(x_in.data(), 4, nx, f2) != nx || std::fread(u_in.data(), 4, nu, f2) != nu) return 1;
            std::fclose(f2);
        }
        host.x = x_in.data(); host.u = u_in.data();
        for (int k = 0; k < K; ++k) {
            if (alore_nmpc_batch_alloc(h2, B, &slots[k]) != ALORE_NMPC_OK || alore_nmpc_batch_upload(h2, &slots[k], &host, B, nullptr) != ALORE_NMPC_OK) {
                std::fprintf(stderr, "slot %d: %s\n", k, alore_nmpc_last_error(h2));
                return 2;
            }
        }
        if (alore_nmpc_rti_many(h2, slots.data(), K, B, 1, nullptr) != ALORE_NMPC_OK) { std::fprintf(stderr, "rti_many: %s\n", alore_nmpc_last_error(h2)); return 2; }
        int identical = 0;
        std::vector<float> xs(nx), us(nu), ks(B);
        std::vector<int> ss(B);
        for (int k = 0; k < K; ++k) {
            alore_nmpc_batch b2{};
            b2.x = xs.data(); b2.u = us.data(); b2.status = ss.data(); b2.kkt = ks.data();
            if (alore_nmpc_batch_download(h2, &slots[k], &b2, B, nullptr) != ALORE_NMPC_OK) return 2;
            identical += (xs == x.v && us == u.v && ss == status && ks == kkt) ? 1 : 0;
            (void)alore_nmpc_batch_free(h2, &slots[k]);
        }
        std::printf("slots=%d identical=%d\n", K, identical);
        (void)alore_nmpc_destroy(h2);
        if (identical != K) return 4;
    }
    CHECK(alore_nmpc_batch_free(h, &dev));
    CHECK(alore_nmpc_destroy(h));
    return solved == B ? 0 : 3;
}
