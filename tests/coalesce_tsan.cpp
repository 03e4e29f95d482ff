/* coalesce_tsan.cpp -- the leader / follower queue behind the per-read reference functions (scrappie_amd/csrc/sh_coalesce.h)
 * under -fsanitize=thread: T caller threads, each making CALLS calls of three kinds against three queues, with stub launches
 * (a sleep and a checksum) in place of the GPU work.  Checks, per call: the result is the one computed from ITS input (two-phase
 * queue: from the bytes its own thread copied into the launch's staging buffer); per queue: every request served exactly once.
 *   g++ -std=c++17 -O1 -g -fsanitize=thread -pthread -Iscrappie_amd/csrc tests/coalesce_tsan.cpp -o /tmp/coalesce_tsan && /tmp/coalesce_tsan 48 200
 * Exit code 0 and no ThreadSanitizer report = clean (tests/test_host_cpu.py::test_coalescer_under_thread_sanitizer). */
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>

#include "sh_coalesce.h"

struct Req {
    int phase = 0;
    std::vector<uint32_t> in;     /* the call's input */
    uint32_t *dst = nullptr;      /* where its own thread copies it (two-phase queue) */
    int kind = 0;                 /* compatibility class: only equal kinds share a launch */
    uint64_t result = 0;
    int served = 0;
};

static uint64_t checksum(const uint32_t *p, size_t n, int kind) {
    uint64_t h = 1469598103934665603ull + (uint64_t)kind;
    for (size_t i = 0; i < n; i++) h = (h ^ p[i]) * 1099511628211ull;
    return h;
}

static ShPresence g_presence;
static ShCoalescer<Req> g_plain, g_two;
static std::vector<uint32_t> g_stage;          /* the two-phase queue's staging buffer (owned by the leader between stage and serve) */
static std::atomic<unsigned long long> g_served{0};

static void take_same_kind(std::deque<Req *> &q, std::vector<Req *> &batch, size_t max_reqs) {
    const int kind = q.front()->kind;
    for (auto it = q.begin(); it != q.end() && batch.size() < max_reqs;) {
        if ((*it)->kind == kind) { batch.push_back(*it); it = q.erase(it); } else ++it;
    }
}

static uint64_t call_plain(Req &r) {
    ShInside in(g_presence);
    g_plain.run(r, g_presence, 64, false,
        [](std::deque<Req *> &q, std::vector<Req *> &b) { take_same_kind(q, b, 64); },
        [](std::vector<Req *> &) { return true; }, [](Req &) {},
        [](std::vector<Req *> &b) {
            std::this_thread::sleep_for(std::chrono::microseconds(300));
            for (Req *c : b) { c->result = checksum(c->in.data(), c->in.size(), c->kind); c->served++; g_served++; }
        });
    return r.result;
}

static uint64_t call_two_phase(Req &r) {
    ShInside in(g_presence);
    std::vector<size_t> at;
    g_two.run(r, g_presence, 32, true,
        [](std::deque<Req *> &q, std::vector<Req *> &b) { take_same_kind(q, b, 32); },
        [&](std::vector<Req *> &b) {
            size_t total = 0;
            at.clear();
            for (Req *c : b) { at.push_back(total); total += c->in.size(); }
            g_stage.assign(total, 0xdeadbeefu);
            for (size_t k = 0; k < b.size(); k++) b[k]->dst = g_stage.data() + at[k];
            return true;
        },
        [](Req &c) { memcpy(c.dst, c.in.data(), c.in.size() * 4); },
        [&](std::vector<Req *> &b) {
            std::this_thread::sleep_for(std::chrono::microseconds(200));
            for (size_t k = 0; k < b.size(); k++) { b[k]->result = checksum(g_stage.data() + at[k], b[k]->in.size(), b[k]->kind); b[k]->served++; g_served++; }
        });
    return r.result;
}

int main(int argc, char **argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 32, CALLS = argc > 2 ? atoi(argv[2]) : 100;
    std::atomic<int> bad{0};
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++) th.emplace_back([&, t] {
        uint32_t s = 12345u + (uint32_t)t * 7919u;
        for (int c = 0; c < CALLS; c++) {
            Req r;
            s = s * 1664525u + 1013904223u;
            r.kind = (int)((s >> 8) % 3);
            r.in.resize(16 + (s >> 16) % 200);
            for (auto &v : r.in) { s = s * 1664525u + 1013904223u; v = s; }
            const uint64_t want = checksum(r.in.data(), r.in.size(), r.kind);
            const uint64_t got = (c & 1) ? call_two_phase(r) : call_plain(r);
            if (got != want || r.served != 1 || r.phase != 3) bad++;
            if ((s >> 20) % 7 == 0) std::this_thread::sleep_for(std::chrono::microseconds((s >> 4) % 400));      /* the host side of the loop body */
        }
    });
    for (auto &x : th) x.join();
    const unsigned long long want = (unsigned long long)T * CALLS;
    printf("coalesce_tsan: %d threads x %d calls: %llu served (want %llu), %d wrong; plain queue %llu launches (largest %zu), two-phase queue %llu launches (largest %zu)\n",
           T, CALLS, g_served.load(), want, bad.load(), g_plain.n_batches, g_plain.max_batch, g_two.n_batches, g_two.max_batch);
    return (bad.load() || g_served.load() != want) ? 1 : 0;
}
