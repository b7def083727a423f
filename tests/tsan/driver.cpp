// ThreadSanitizer driver for the library's host threading code (runtime.hip: LocalGroup, multi_run, the loop-back collectives,
// the group shutdown; staging.hip: the staged-copy workers), built for the CPU against the stub device layer in this
// directory.  Exit code 0 and no ThreadSanitizer report = pass (tests/test_tsan_cpu.py).
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "internal.hpp"

using namespace tlsq;

#define CHECK(c)                                                        \
    do {                                                                \
        if (!(c)) {                                                     \
            fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); \
            exit(2);                                                    \
        }                                                               \
    } while (0)

int main() {
    const int n = 4;
    const int devs[n] = {0, 0, 0, 0};   // the same device four times: a loop-back group (host-staged collectives)
    tlsq_handle h = nullptr;
    CHECK(tlsq_create_multi(n, devs, &h) == TLSQ_OK && tlsq_ngpus(h) == n);

    // 1. collectives in lock step on all ranks: all-reduce (sum, max), all-gather, the host-scalar form
    for (int round = 0; round < 50; ++round) {
        const size_t cnt = 64 + 13 * (size_t)(round % 7);
        std::atomic<int> bad{0};
        const int st = multi_run(h, [&](Handle* hr, int r, int nr) -> int {
            void* p;
            TLSQ_TRY(ws_get(hr, WS_G, (cnt * (size_t)nr + cnt) * 8, &p));
            double* v = (double*)p;
            for (size_t i = 0; i < cnt; ++i) v[i] = (double)(r + 1) * (double)(i + 1);
            TLSQ_TRY(comm_allreduce(hr, v, cnt, ncclSum));
            for (size_t i = 0; i < cnt; ++i)
                if (v[i] != (double)(nr * (nr + 1) / 2) * (double)(i + 1)) bad.fetch_add(1);
            double mx = (double)r;
            TLSQ_TRY(comm_allreduce_host_scalar(hr, &mx, ncclMax));
            if (mx != (double)(nr - 1)) bad.fetch_add(1);
            for (size_t i = 0; i < cnt; ++i) v[i] = (double)r;
            double* all = v + cnt;
            TLSQ_TRY(comm_allgather(hr, v, all, cnt));
            for (int q = 0; q < nr; ++q)
                if (all[(size_t)q * cnt] != (double)q || all[(size_t)q * cnt + cnt - 1] != (double)q) bad.fetch_add(1);
            return TLSQ_OK;
        });
        CHECK(st == TLSQ_OK && bad.load() == 0);
    }

    // 2. a rank leaves the group call with an error while the others are in (or on their way to) a collective: nobody may
    //    hang, everybody comes back, the root cause is the status that is reported, and the group works again afterwards
    for (int bad_rank = 0; bad_rank < n; ++bad_rank) {
        const int st = multi_run(h, [&](Handle* hr, int r, int) -> int {
            void* p;
            TLSQ_TRY(ws_get(hr, WS_G, 4096, &p));
            double* v = (double*)p;
            for (int it = 0; it < 20; ++it) {
                if (it == 7 && r == bad_rank) return set_err(hr, TLSQ_ERR_ARG, "injected failure on rank %d", r);
                v[0] = 1.0;
                TLSQ_TRY(comm_allreduce(hr, v, 1, ncclSum));
            }
            return TLSQ_OK;
        });
        CHECK(st == TLSQ_ERR_ARG);
        CHECK(strstr(tlsq_last_error(h), "injected failure") != nullptr);
        const int again = multi_run(h, [&](Handle* hr, int, int nr) -> int {
            double one = 1.0;
            TLSQ_TRY(comm_allreduce_host_scalar(hr, &one, ncclSum));
            return one == (double)nr ? TLSQ_OK : TLSQ_ERR_COMM;
        });
        CHECK(again == TLSQ_OK);
    }

    // 3. the staged-copy workers: a strided host matrix to the "device" and back, together with a second job in the other
    //    direction (what a host-pointer rpca call does around the solve), on the group's ranks concurrently
    {
        const int64_t M = 20011, N = 192, ld = M + 5;
        std::vector<double> src((size_t)ld * N), back((size_t)ld * N, -1.0);
        for (size_t i = 0; i < src.size(); ++i) src[i] = (double)(i % 1000003);
        std::atomic<int> bad{0};
        const int st = multi_run(h, [&](Handle* hr, int r, int nr) -> int {
            const int64_t bs = M / nr, lo = r * bs, rows = r == nr - 1 ? M - lo : bs;
            void* d;
            TLSQ_TRY(ws_get(hr, WS_D, (size_t)rows * N * 8, &d));
            const StageJob up{d, rows, src.data() + lo, ld, rows, N, 8, true};
            TLSQ_TRY(staged_copy(hr, &up, 1));
            const double* dd = (const double*)d;
            for (int64_t c = 0; c < N; c += 17)
                for (int64_t i = 0; i < rows; i += 101)
                    if (dd[i + c * rows] != src[(size_t)(lo + i + c * ld)]) bad.fetch_add(1);
            const StageJob down{back.data() + lo, ld, d, rows, rows, N, 8, false};
            TLSQ_TRY(staged_copy(hr, &down, 1));
            return TLSQ_OK;
        });
        CHECK(st == TLSQ_OK && bad.load() == 0);
        for (int64_t c = 0; c < N; ++c)
            for (int64_t i = 0; i < M; ++i) CHECK(back[(size_t)(i + c * ld)] == src[(size_t)(i + c * ld)]);
    }
    CHECK(tlsq_destroy(h) == TLSQ_OK);

    // 4. independent handles on concurrent host threads (each with its own stager)
    {
        std::vector<std::thread> th;
        std::atomic<int> bad{0};
        for (int t = 0; t < 3; ++t)
            th.emplace_back([&, t] {
                tlsq_handle hh = nullptr;
                if (tlsq_create(0, &hh) != TLSQ_OK) {
                    bad.fetch_add(1);
                    return;
                }
                const int64_t M = 300000 + 1000 * t, N = 8;
                std::vector<double> a((size_t)M * N, (double)t), b((size_t)M * N, -1.0);
                void* d;
                if (ws_get(hh, WS_D, (size_t)M * N * 8, &d) < 0) bad.fetch_add(1);
                const StageJob up{d, M, a.data(), M, M, N, 8, true};
                const StageJob down{b.data(), M, d, M, M, N, 8, false};
                if (staged_copy(hh, &up, 1) < 0 || staged_copy(hh, &down, 1) < 0 || a != b) bad.fetch_add(1);
                tlsq_destroy(hh);
            });
        for (auto& x : th) x.join();
        CHECK(bad.load() == 0);
    }
    printf("tsan driver: ok\n");
    return 0;
}
