#ifndef BESSX_SYNC_H
#define BESSX_SYNC_H
// bessx_sync.h -- the host-side concurrency primitives of libbessx.so, free of any HIP call so that they can be built
// and hammered under ThreadSanitizer on the CPU (tools/tsan/sync_harness.cpp, `make -C tools/tsan`):
//   * FoldPool        -- the host threads that queue the launches of chains that run side by side (CV fold fits,
//                        bessx_cv.cpp; chunk chains, bessx_kchunks.cpp);
//   * FillRendezvous  -- the rule "nobody reads the Gram column cache's slot map while it is rewritten" among chunk
//                        chains: a chain that has to fill waits until every other chain stands still.
// What touches the device is handed in by the caller: the per-thread initialisation (hipSetDevice) and the function
// that drains a chain's stream (hipStreamSynchronize).
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

// A bounded wait on a condition variable.  The library waits on the steady clock (a deadline must not move with the wall
// clock); libstdc++ implements that with pthread_cond_clockwait, which the ThreadSanitizer runtime of gcc 11 does not
// intercept -- it then misses the unlock / relock inside the wait and reports a "double lock".  The sanitizer build
// (-DBESSX_SYNC_TSAN, tools/tsan) waits on the system clock instead (pthread_cond_timedwait, intercepted): the only
// difference between the code under test and the code shipped.
template <class Pred>
inline bool bessx_timed_wait(std::condition_variable &cv, std::unique_lock<std::mutex> &lk, double seconds, Pred pred) {
#ifdef BESSX_SYNC_TSAN
  return cv.wait_until(lk, std::chrono::system_clock::now() +
                               std::chrono::duration_cast<std::chrono::system_clock::duration>(
                                   std::chrono::duration<double>(seconds)), pred);
#else
  return cv.wait_for(lk, std::chrono::duration<double>(seconds), pred);
#endif
}

// Host threads that queue the chains' launches: launches that alternate between streams cost the host ~10 us each
// (measured: 35 launches per round, 12 ms per path of configs[3]); one thread per chain queues its 7 on its own stream
// while the others do the same.  Workers spin for a job for a while after the last one, then block on a condition
// variable (an idle session holds no core).  The spin is ~4 ms where the host has cores to spare (longer than the
// longest gap inside a path -- a union fill of three groups is 2.5 ms; with 1 ms the workers slept through the fills and
// configs[3] took 32.9 instead of 29.4 ms) and ~0.2 ms where K spinning threads per session would oversubscribe it
// (fewer than 4 hardware threads per chain: several ranks or sessions per host); BESSX_POOL_SPIN_US overrides.
// The caller's wait for its workers is bounded: spin, then sleep on a condition variable, and give up at the
// session's deadline (a worker stuck inside a HIP call) -- the pool is then marked broken and never joined.
struct FoldPool {
  std::vector<std::thread> th;
  std::mutex mu;
  std::condition_variable cv, cv_done;
  unsigned ticket = 0;  // (under mu) number of the current job
  std::atomic<unsigned> ticket_hint{0};  // ... its copy for the spinning phase
  std::atomic<int> pending{0};
  bool quit = false, broken = false;
  std::function<void(int)> job;
  std::function<void()> thread_init;  // first thing every worker does (the library: hipSetDevice of the session's device)
  int spin_iters = 200000;  // pauses of ~40-50 cycles
  void worker(int k) {
    if (thread_init) thread_init();
    unsigned seen = 0;
    for (;;) {
      bool got = false;
      for (int spin = 0; spin < spin_iters && !got; spin++) {
        got = ticket_hint.load(std::memory_order_acquire) != seen;
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
      }
      std::function<void(int)> mine;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return quit || ticket != seen; });
        if (quit) return;
        seen = ticket;
        mine = job;
      }
      mine(k);
      if (pending.fetch_sub(1, std::memory_order_acq_rel) == 1) {
        std::lock_guard<std::mutex> lk(mu);  // (the caller may be asleep on cv_done)
        cv_done.notify_all();
      }
    }
  }
  void start(int nworkers, std::function<void()> init = nullptr) {
    thread_init = std::move(init);
    const unsigned hw = std::thread::hardware_concurrency();
    spin_iters = (hw >= 4u * (unsigned)(nworkers + 1)) ? 200000 : 10000;
    if (const char *ev = std::getenv("BESSX_POOL_SPIN_US")) spin_iters = std::max(0, std::atoi(ev)) * 50;
    for (int k = 1; k <= nworkers; k++) th.emplace_back([this, k] { worker(k); });
  }
  // runs fn(0) on the caller and fn(1..nworkers) on the workers; true when all are done, false when the workers did
  // not finish within deadline_s (the pool is then broken: its threads may still be inside fn)
  bool run(const std::function<void(int)> &fn, double deadline_s) {
    if (broken) return false;
    {
      std::lock_guard<std::mutex> lk(mu);
      job = fn;
      pending.store((int)th.size(), std::memory_order_relaxed);
      ticket++;
      ticket_hint.store(ticket, std::memory_order_release);
    }
    cv.notify_all();
    fn(0);
    for (int spin = 0; spin < 400000; spin++) {  // ~8 ms: the workers queue a handful of launches each
      if (pending.load(std::memory_order_acquire) == 0) return true;
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    }
    std::unique_lock<std::mutex> lk(mu);
    const bool ok = bessx_timed_wait(cv_done, lk, deadline_s, [&] { return pending.load(std::memory_order_acquire) == 0; });
    if (!ok) broken = true;
    return ok;
  }
  void stop() {
    {
      std::lock_guard<std::mutex> lk(mu);
      quit = true;
    }
    cv.notify_all();
    for (auto &t : th) {
      if (broken)
        t.detach();  // a worker that never came back from a HIP call cannot be joined
      else
        t.join();
    }
    th.clear();
  }
};


// The fill rendezvous of the chunk chains (bessx_kchunks.cpp): C chains run side by side and READ one Gram column cache;
// a chain that needs a column the cache lacks may rewrite the slot map only while every other chain stands still.
//   running       chains that may have kernels in flight
//   fill_pending  a chain holds (or waits for) the right to fill
//   abandoned     a chain failed, or a wait ran into the deadline: nobody waits any longer
// `drain` = what makes the calling chain quiet (the library: hipStreamSynchronize of its stream); it is called with the
// mutex released.
// Round 5, `concurrent` rounds: the fills are STAGED -- the filling chain writes Gram columns into slots no reader can
// see yet and publishes the slot map entries with the last launch of its fill (k_cov_publish_slots) -- so nobody has to
// stand still: the rendezvous shrinks to the right to fill (one fill at a time) and a count of completed fills, by
// which a chain tells that the look-up it parked on is out of date (another chain's fill ended since: look up again).
struct FillRendezvous {
  std::mutex mu;
  std::condition_variable cv;
  int running = 0;
  bool fill_pending = false;
  bool abandoned = false;
  bool concurrent = false;
  unsigned long long fills_done = 0;
  double deadline_s = 30.0;

  // the caller of a round sets `running` to the number of chains the round starts BEFORE any of them runs: a chain
  // that parks at once must not take the others for finished
  void round(int chains, bool concurrent_fills = false) {
    std::lock_guard<std::mutex> lk(mu);
    running = chains;
    fill_pending = false;
    abandoned = false;
    concurrent = concurrent_fills;
  }

  unsigned long long generation() {
    std::lock_guard<std::mutex> lk(mu);
    return fills_done;
  }

  // a chain has run to the end of its share of the round (failed: nobody waits any longer)
  void leave(bool failed) {
    std::lock_guard<std::mutex> lk(mu);
    running--;
    if (failed) {
      abandoned = true;
      fill_pending = false;
    }
    cv.notify_all();
  }

  // between two candidates of a chain: if another chain waits to fill, drain this chain and stand still until it has
  template <class Drain>
  void safe_point(Drain &&drain) {
    std::unique_lock<std::mutex> lk(mu);
    if (!fill_pending || abandoned || concurrent) return;
    lk.unlock();
    drain();  // (the candidate chained ahead runs to its end or parks)
    lk.lock();
    running--;
    cv.notify_all();
    if (!bessx_timed_wait(cv, lk, deadline_s, [&] { return !fill_pending || abandoned; })) {
      abandoned = true;  // (a fill that never ended: nobody waits any longer, every chain fails at its next look)
      cv.notify_all();
    }
    running++;
  }

  // a parked chain asks for the cache: 0 = it may fill (every other chain stands still), 1 = another chain filled
  // while this one waited (the right is held all the same: look the columns up again), -1 = the run was abandoned
  // (concurrent rounds: `seen` = the count of completed fills the caller's look-up is known to have run after; 1 is
  // returned whenever a fill has ended since)
  template <class Drain>
  int fill_begin(Drain &&drain, unsigned long long seen = 0) {
    std::unique_lock<std::mutex> lk(mu);
    if (concurrent) {
      const bool ok = bessx_timed_wait(cv, lk, deadline_s, [&] { return !fill_pending || abandoned; });
      if (!ok || abandoned) {
        abandoned = true;
        cv.notify_all();
        return -1;
      }
      fill_pending = true;
      return fills_done != seen ? 1 : 0;
    }
    int waited = 0;
    while (fill_pending && !abandoned) {  // another chain is filling: this one is quiet (parked, its stream drained)
      waited = 1;
      lk.unlock();
      drain();
      lk.lock();
      if (!fill_pending) break;
      running--;
      cv.notify_all();
      bessx_timed_wait(cv, lk, deadline_s, [&] { return !fill_pending || abandoned; });
      running++;
    }
    if (abandoned) return -1;
    fill_pending = true;
    running--;
    const bool ok = bessx_timed_wait(cv, lk, deadline_s, [&] { return running == 0 || abandoned; });
    running++;
    if (!ok || abandoned) {
      abandoned = true;
      fill_pending = false;
      cv.notify_all();
      return -1;
    }
    return waited;
  }

  void fill_end(bool filled = true) {
    std::lock_guard<std::mutex> lk(mu);
    fill_pending = false;
    if (filled) fills_done++;
    cv.notify_all();
  }
};

// PassRendezvous -- chunk chains that SHARE their passes over X (round 6).  Every chain of the streaming forms needs
// X^T v for a vector of its own at every PDAS iteration; instead of one launch per chain -- each streaming the whole of X
// -- the chains hand their vector sets to the next multi-chain launch (k_xtv_mc / k_cox_score1p_mc).  A member that
// submits blocks until the batch it went into has been LAUNCHED (not run: the launch is ordered after every
// participant's earlier work by events); the batch is launched by the member that completes it (arrived == members), by a
// member that leaves while everybody else is waiting, or -- after `timeout_s` -- by a waiting member with whoever is
// there (a straggler then gets a launch of its own: slower, same sums).  What touches the device is handed in:
//   add(i)        under the lock: put the caller's request into place i of the batch
//   launch(n, g)  under the lock: launch the n requests of batch number g
//   after(g)      under the lock, in every participant of batch g after its launch: order the caller's stream behind it
// Invariants: a member has at most one request in flight; arrived <= members at every launch decision; batch numbers
// are handed out in launch order.
struct PassRendezvous {
  std::mutex mu;
  std::condition_variable cv;
  int members = 0, arrived = 0;
  unsigned long long gen = 0;
  double timeout_s = 0.05;
  unsigned long long partial = 0;  // batches launched on a timeout (statistics)

  void reset(int m) {
    std::lock_guard<std::mutex> lk(mu);
    members = m;
    arrived = 0;
  }
  template <class Launch>
  void fire(Launch &launch) {  // (under the lock)
    launch(arrived, gen);
    arrived = 0;
    gen++;
    cv.notify_all();
  }
  template <class Add, class Launch, class After>
  void submit(Add add, Launch launch, After after) {
    std::unique_lock<std::mutex> lk(mu);
    add(arrived);
    arrived++;
    const unsigned long long mine = gen;
    if (arrived >= members) {
      fire(launch);
    } else {
      const bool ok = bessx_timed_wait(cv, lk, timeout_s, [&] { return gen != mine; });
      if (!ok && gen == mine) {
        partial++;
        fire(launch);
      }
    }
    after(mine);
  }
  template <class Launch>
  void leave(Launch launch) {
    std::lock_guard<std::mutex> lk(mu);
    members--;
    if (arrived > 0 && arrived >= members) fire(launch);
  }
  void join() {
    std::lock_guard<std::mutex> lk(mu);
    members++;
  }
};

#endif  // BESSX_SYNC_H
