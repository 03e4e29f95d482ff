/* sh_coalesce.h -- one leader / follower queue behind the per-read reference functions (posterior, decode_transducer,
 * decode_crf): the calls that are waiting run as ONE launch.  Plain C++ (no HIP): tests/coalesce_tsan.cpp runs it under
 * -fsanitize=thread with stub launches.
 *
 * The reference's loop body is re-entrant by construction -- every OpenMP thread owns its read (scrappie_raw.c:355,387).  Here
 * a call joins a queue; the first caller that finds no launch in preparation becomes the LEADER: it waits for company (below),
 * takes the compatible requests off the queue, optionally has every member copy its own input into the launch's staging buffer
 * (all at once, each on its own thread: a posterior is 3.3 MB per read), runs the launch with the lock released, marks the
 * members done and wakes everybody; the next waiting caller becomes the next leader.
 *
 * Waiting for company: a launch lasts as long as its longest read's chain (~10 ms for 4000 samples) whatever it holds, and the
 * callers the last launch has just released come back one by one over the next millisecond or two.  The leader waits until
 * 3/4 of the threads seen inside the per-read functions lately (`peak`, maintained by ShInside) have joined, in windows of
 * window_us: a window in which nobody arrives ends the wait, and so does max_us in all.  A caller that joins NOTIFIES the leader
 * (its own condition variable: with one shared variable every join woke every waiting caller -- 256 callers, 65 000 wake-ups per
 * launch, enough CPU time to run an 8-rank job's 16-CPU quota dry: profiles/r6_batch64_throttle.txt), so the
 * leader sees the target reached at once instead of sleeping out its window.  A process with one calling thread never waits
 * (target 1).
 *
 * Req needs:  int phase  (0 queued, 1 asked to copy its input, 2 copied, 3 done).
 */
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <mutex>
#include <vector>

/* how many threads are inside the per-read functions right now / were lately: the coalescers' idea of how much company to expect */
struct ShPresence {
    std::atomic<int> inside{0}, peak{0};
    std::atomic<long long> peak_ms{0};
    void enter() {
        const int n = ++inside;
        const long long now = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
        if (n >= peak.load() || now - peak_ms.load() > 500) { peak.store(n); peak_ms.store(now); }
    }
    void leave() { --inside; }
};
struct ShInside {
    ShPresence &p;
    explicit ShInside(ShPresence &p_) : p(p_) { p.enter(); }
    ~ShInside() { p.leave(); }
};

struct ShCoalesceTuning {
    int window_us, max_us;
    static const ShCoalesceTuning &get() {      /* SCRAPPIE_HIP_COALESCE_US (default 500), SCRAPPIE_HIP_COALESCE_MAX_US (default 10000); read once */
        static const ShCoalesceTuning t = [] {
            const char *w = getenv("SCRAPPIE_HIP_COALESCE_US"), *m = getenv("SCRAPPIE_HIP_COALESCE_MAX_US");
            return ShCoalesceTuning{w ? std::max(0, atoi(w)) : 500, m ? std::max(0, atoi(m)) : 10000};
        }();
        return t;
    }
};

template <class Req>
class ShCoalescer {
 public:
    std::mutex mu;
    int target_pct = 75;             /* the share of the threads seen lately a leader waits for (per-read calls: 3/4 -- the others are between two calls; batch
                                      * calls: all of them -- a caller brings 64 reads and is blocked until its launch has run, so a launch that leaves a quarter of
                                      * the callers behind costs a whole second launch: bench.py batch64, 256 threads, 2 engine calls -> 1) */
    int window_mul = 1;              /* the leader's waiting window in units of SCRAPPIE_HIP_COALESCE_US (batch calls: 4 -- a launch left behind costs ~20 ms, and hundreds
                                      * of caller threads on a few CPUs take more than 500 us to all arrive) */
    unsigned long long n_batches = 0, n_reads = 0, service_us = 0;      /* (under mu) */
    size_t max_batch = 0, last_batch = 0;

    /* Called by every caller with its own request; returns when r.phase == 3.
     *   take(queue, batch)   leader, under the lock: move the requests of the next launch from the queue (front first) into batch
     *   stage(batch) -> bool leader, under the lock: give every member a destination for its input; false = no copy phase (or no memory:
     *                        the launch is then skipped and the members are released with whatever their result fields hold)
     *   copy_in(req)         the member's own thread, lock released
     *   serve(batch)         leader, lock released: the launch; fills the members' results
     * two_phase = false: stage / copy_in are not called. */
    template <class Take, class Stage, class CopyIn, class Serve>
    void run(Req &r, ShPresence &presence, size_t max_reqs, bool two_phase, Take take, Stage stage, CopyIn copy_in, Serve serve) {
        const ShCoalesceTuning &tn = ShCoalesceTuning::get();
        std::unique_lock<std::mutex> lk(mu);
        q_.push_back(&r);
        cv_lead_.notify_one();                                        /* a leader waiting for company counts again (nobody else listens there: joining
                                                                       * must not wake the hundreds of callers that are waiting for their launch) */
        while (r.phase != 3) {
            if (r.phase == 1) {                                       /* my input into the launch's staging buffer, beside everybody else's */
                lk.unlock();
                copy_in(r);
                lk.lock();
                r.phase = 2;
                if (--copying_ == 0) cv_lead_.notify_one();
                continue;
            }
            if (running_ || r.phase != 0) { cv_.wait(lk); continue; }
            running_ = true;                                          /* leader */
            if (tn.window_us > 0) {
                const auto t0 = std::chrono::steady_clock::now();
                for (;;) {
                    const size_t target = ((size_t)presence.peak.load() * (size_t)target_pct + 99) / 100, before = q_.size();
                    if (before >= target || before >= max_reqs) break;
                    timed_wait(lk, tn.window_us * window_mul, [&] { return q_.size() >= std::min(target, max_reqs); });
                    if (q_.size() == before) break;                   /* nobody came in a whole window: they are not coming */
                    if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(tn.max_us)) break;
                }
            }
            std::vector<Req *> batch;
            take(q_, batch);
            bool go = !batch.empty();
            if (go && two_phase) {
                go = stage(batch);
                if (go) {
                    copying_ = (int)batch.size();
                    for (Req *c : batch) c->phase = 1;
                    cv_.notify_all();
                    if (r.phase == 1) {                               /* (the leader's own request, if it is in the batch) */
                        lk.unlock();
                        copy_in(r);
                        lk.lock();
                        r.phase = 2; --copying_;
                    }
                    while (copying_ > 0) cv_lead_.wait(lk);
                }
            }
            if (go) {
                lk.unlock();
                const auto tb0 = std::chrono::steady_clock::now();
                serve(batch);
                const auto tb1 = std::chrono::steady_clock::now();
                lk.lock();
                service_us += (unsigned long long)std::chrono::duration_cast<std::chrono::microseconds>(tb1 - tb0).count();
            }
            for (Req *c : batch) c->phase = 3;
            n_batches++; n_reads += batch.size(); max_batch = std::max(max_batch, batch.size()); last_batch = batch.size();
            running_ = false;
            cv_.notify_all();
        }
    }

 private:
    /* wait_for on the steady clock is pthread_cond_clockwait, which the ThreadSanitizer runtime of this toolchain (gcc 11) does not
     * intercept: it then misses the unlock inside the wait and reports every other holder of the mutex as a race.  Under the
     * sanitizer the same wait runs on the system clock (pthread_cond_timedwait, intercepted); the product uses the steady clock. */
    template <class Pred>
    void timed_wait(std::unique_lock<std::mutex> &lk, int us, Pred pred) {
#if defined(__SANITIZE_THREAD__)
        cv_lead_.wait_until(lk, std::chrono::system_clock::now() + std::chrono::microseconds(us), pred);
#else
        cv_lead_.wait_for(lk, std::chrono::microseconds(us), pred);
#endif
    }
    std::condition_variable cv_;          /* members and callers in the queue: phase changes, the leader's seat free (broadcast, once or twice per launch) */
    std::condition_variable cv_lead_;     /* the leader alone: company has arrived / the last member has copied its input (one waiter, notify_one per event) */
    std::deque<Req *> q_;
    bool running_ = false;
    int copying_ = 0;
};
