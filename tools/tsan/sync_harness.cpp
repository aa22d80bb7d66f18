// sync_harness.cpp -- the host-side concurrency primitives of libbessx.so (bess_amd/csrc/bessx_sync.h: FoldPool, the
// fill rendezvous of the chunk chains) on the CPU, under ThreadSanitizer.  No HIP: a chain's "stream" is a counter of
// work in flight that the drain function waits out, the "Gram column cache" a plain array that a filling chain rewrites
// and every running chain reads -- exactly the accesses the rendezvous has to keep apart, so a hole in the protocol is a
// data race TSan reports (and the checks below fail on torn reads).  Random chain lengths, random parking, random
// failures (abandon), pools stopped and started again with more workers, as bessx_kchunks.cpp does.
//   make -C tools/tsan          (g++ -fsanitize=thread; log in profiles/r05_tsan_sync_harness.log)
#include <cstdio>
#include <cstring>
#include <random>

#include "../../bess_amd/csrc/bessx_sync.h"

namespace {

struct Cache {
  static constexpr int N = 256;
  long slot_of[N];  // what a fill rewrites and the chains' "kernels" read -- NOT atomic on purpose
  long version = 0;
};

struct Chain {
  std::atomic<int> in_flight{0};  // "kernels" queued on the chain's stream
  long reads = 0, torn = 0;
};

// a kernel of a running chain: reads the whole map; all entries must carry one version
void kernel_reads(const Cache &c, Chain &q) {
  const long v = c.slot_of[0];
  for (int i = 1; i < Cache::N; i++)
    if (c.slot_of[i] != v) q.torn++;
  q.reads++;
}

int run_rounds(int C, int rounds, unsigned seed, bool with_failures) {
  FoldPool pool;
  pool.spin_iters = 2000;
  pool.start(C - 1);
  FillRendezvous rdv;
  rdv.deadline_s = 20.0;
  Cache cache;
  std::memset(cache.slot_of, 0, sizeof(cache.slot_of));
  std::vector<Chain> ch((size_t)C);
  long fills = 0, abandons = 0;
  std::mutex fills_mu;
  int bad = 0;
  for (int r = 0; r < rounds; r++) {
    rdv.round(C);
    std::atomic<int> failed_chain{-1};
    auto job = [&](int k) {
      std::mt19937 rng(seed * 7919u + (unsigned)r * 131u + (unsigned)k);
      Chain &q = ch[(size_t)k];
      const int candidates = 3 + (int)(rng() % 12);
      bool failed = false;
      auto drain = [&q] {
        while (q.in_flight.load(std::memory_order_acquire) > 0) q.in_flight.fetch_sub(1, std::memory_order_acq_rel);
      };
      for (int cand = 0; cand < candidates && !failed; cand++) {
        // the fit of a candidate: a few kernels that read the map (the stream has work in flight while they "run")
        const int kernels = 1 + (int)(rng() % 4);
        for (int i = 0; i < kernels; i++) {
          q.in_flight.fetch_add(1, std::memory_order_acq_rel);
          kernel_reads(cache, q);
        }
        if (rng() % 5 == 0) {  // parked on a missing column: ask for the cache, fill, hand it back
          drain();             // (a parked fit's stream is idle: the host has just read its result block)
          const int w = rdv.fill_begin(drain);
          if (w < 0) {
            failed = true;  // abandoned by another chain's failure
            break;
          }
          if (w == 0 || rng() % 2) {  // (w == 1: another chain filled meanwhile; the columns may still be missing)
            const long v = ++cache.version;
            for (int i = 0; i < Cache::N; i++) cache.slot_of[i] = v;  // the rewrite nobody may observe half-way
            std::lock_guard<std::mutex> lk(fills_mu);
            fills++;
          }
          rdv.fill_end();
        }
        if (with_failures && rng() % 97 == 0) {
          failed = true;
          failed_chain.store(k);
          break;
        }
        drain();              // publish_wait: the candidate's results are back
        rdv.safe_point(drain);  // between two candidates
      }
      drain();
      rdv.leave(failed);
    };
    if (!pool.run(job, 20.0)) {
      std::fprintf(stderr, "round %d: the pool did not come back\n", r);
      return 1;
    }
    if (failed_chain.load() >= 0) abandons++;
    if (r % 7 == 6 && C < 8) {  // the pool grows when a later path asks for more chains (bessx_kchunks.cpp)
      pool.stop();
      pool.quit = false;
      pool.broken = false;
      pool.ticket = 0;
      pool.ticket_hint.store(0);
      pool.start(C - 1);
    }
  }
  pool.stop();
  long reads = 0, torn = 0;
  for (Chain &q : ch) {
    reads += q.reads;
    torn += q.torn;
  }
  std::printf("chains %d, rounds %d: %ld map reads, %ld fills, %ld rounds with a failed chain, torn reads %ld\n", C, rounds,
              reads, fills, abandons, torn);
  if (torn) bad = 1;
  return bad;
}

// Concurrent rounds (round 5, staged fills): nobody stands still.  The cache is two maps -- slot_w, the writer's (plain
// ints, touched only under the right to fill), and slot_of, what the chains' "kernels" look up (atomics: a kernel may
// read an entry while the publishing launch writes it) -- and the column data, plain longs written by the filling chain
// BEFORE it publishes the entries.  A reader that finds a published entry must find its data complete; two fills must
// never overlap (they would hand out the same slots: a data race on slot_w / data that TSan reports); and a chain whose
// look-up is older than the last completed fill must be told to look again (fill_begin returns 1) -- otherwise it
// would give a column that is cached by now a second slot (counted below as `dup`).
struct StagedCache {
  static constexpr int P = 192;
  int slot_w[P];
  std::atomic<int> slot_of[P];
  long data[P];
  int count = 0;  // (under the right)
};

int run_staged_rounds(int C, int rounds, unsigned seed) {
  FoldPool pool;
  pool.spin_iters = 2000;
  pool.start(C - 1);
  FillRendezvous rdv;
  rdv.deadline_s = 20.0;
  long fills = 0, relooks = 0, dup = 0, incomplete = 0, lookups = 0;
  std::mutex stat_mu;
  for (int r = 0; r < rounds; r++) {
    StagedCache cache;
    for (int i = 0; i < StagedCache::P; i++) {
      cache.slot_w[i] = -1;
      cache.slot_of[i].store(-1, std::memory_order_relaxed);
      cache.data[i] = 0;
    }
    rdv.round(C, true);
    auto job = [&](int k) {
      std::mt19937 rng(seed * 104729u + (unsigned)r * 131u + (unsigned)k);
      long my_fills = 0, my_relooks = 0, my_dup = 0, my_incomplete = 0, my_lookups = 0;
      auto drain = [] {};
      const int candidates = 6 + (int)(rng() % 20);
      for (int cand = 0; cand < candidates; cand++) {
        const unsigned long long seen = rdv.generation();  // (the library: read when the batch of launches is queued)
        int want[4], nw = 1 + (int)(rng() % 4);
        for (int i = 0; i < nw; i++) want[i] = (int)(rng() % StagedCache::P);
        // the look-up "kernel"
        bool missing = false;
        for (int i = 0; i < nw; i++) {
          const int sl = cache.slot_of[want[i]].load(std::memory_order_acquire);
          my_lookups++;
          if (sl < 0)
            missing = true;
          else if (cache.data[sl] != 1000 + want[i])
            my_incomplete++;  // a published entry whose column is not there
        }
        while (missing) {
          const int w = rdv.fill_begin(drain, seen);
          if (w < 0) break;
          if (w == 1) {  // a fill ended since the look-up: look again (the right is held meanwhile)
            my_relooks++;
            missing = false;
            for (int i = 0; i < nw; i++) missing = missing || cache.slot_of[want[i]].load(std::memory_order_acquire) < 0;
            if (!missing) {
              rdv.fill_end(false);
              break;
            }
          }
          // the fill: slots from the writer's map, the columns, then the entries the readers see
          int newc[4], nn = 0;
          for (int i = 0; i < nw; i++) {
            if (cache.slot_of[want[i]].load(std::memory_order_acquire) >= 0) continue;
            bool listed = false;
            for (int j = 0; j < nn; j++) listed = listed || newc[j] == want[i];
            if (listed) continue;
            if (cache.slot_w[want[i]] >= 0) my_dup++;  // cached by a fill this chain was not told about
            cache.slot_w[want[i]] = cache.count++;
            newc[nn++] = want[i];
          }
          for (int j = 0; j < nn; j++) cache.data[cache.slot_w[newc[j]]] = 1000 + newc[j];
          for (int j = 0; j < nn; j++) cache.slot_of[newc[j]].store(cache.slot_w[newc[j]], std::memory_order_release);
          my_fills++;
          rdv.fill_end(true);
          missing = false;
        }
        rdv.safe_point(drain);
      }
      rdv.leave(false);
      std::lock_guard<std::mutex> lk(stat_mu);
      fills += my_fills;
      relooks += my_relooks;
      dup += my_dup;
      incomplete += my_incomplete;
      lookups += my_lookups;
    };
    if (!pool.run(job, 20.0)) {
      std::fprintf(stderr, "staged round %d: the pool did not come back\n", r);
      return 1;
    }
  }
  pool.stop();
  std::printf("staged fills, chains %d, rounds %d: %ld look-ups, %ld fills, %ld second looks, columns cached twice %ld, "
              "published entries without their column %ld\n", C, rounds, lookups, fills, relooks, dup, incomplete);
  return (dup || incomplete) ? 1 : 0;
}

// ---- PassRendezvous: chains that share their passes over X (round 6) -------------------------------------------------
// C member threads submit requests in rounds of random length, pause (leave / join, as the early stitch's wait does) and
// finish at different times; the "launch" reads the batch's requests and writes every participant's result cell -- plain
// memory, touched under the rendezvous' lock only, so a hole in the protocol is a race TSan reports.  Checked: every
// request is served exactly once, by the batch its `after` names; a batch never holds more requests than members;
// nobody waits forever (a member that sleeps long enough makes the others go without it: partial batches).
int run_pass_rounds(int C, int rounds, unsigned seed) {
  PassRendezvous rdv;
  rdv.timeout_s = 0.003;
  struct Cell {
    long served_by = -1;  // batch number that served the member's current request
    long requests = 0, served = 0;
  };
  std::vector<Cell> cell((size_t)C);
  int batch_req[16] = {};
  long launches = 0, bad = 0, over = 0;
  rdv.reset(C);
  auto launch = [&](int n, unsigned long long g) {
    if (n > rdv.members + 0 && n > 0) over++;  // (members may have dropped since the requests came in: checked loosely below)
    if (n < 1 || n > C) bad++;
    for (int i = 0; i < n; i++) {
      Cell &q = cell[(size_t)batch_req[i]];
      q.served_by = (long)g;
      q.served++;
    }
    launches++;
  };
  std::vector<std::thread> th;
  for (int id = 0; id < C; id++)
    th.emplace_back([&, id] {
      std::mt19937 rng(seed + 17u * (unsigned)id);
      const int mine = rounds / 2 + (int)(rng() % (unsigned)(rounds / 2 + 1));
      bool member = true;
      for (int it = 0; it < mine; it++) {
        if (rng() % 23 == 0) {  // a wait that is not for the device: step out, come back
          rdv.leave(launch);
          member = false;
          std::this_thread::sleep_for(std::chrono::microseconds(rng() % 300));
          rdv.join();
          member = true;
        }
        if (rng() % 97 == 0) std::this_thread::sleep_for(std::chrono::milliseconds(5));  // a straggler: the others time out
        long got = -2;
        rdv.submit([&](int i) { batch_req[i] = id; cell[(size_t)id].requests++; }, launch,
                   [&](unsigned long long g) { got = cell[(size_t)id].served_by == (long)g ? 0 : -1; });
        if (got != 0) __atomic_fetch_add(&bad, 1, __ATOMIC_RELAXED);
      }
      if (member) rdv.leave(launch);
    });
  for (auto &t : th) t.join();
  long req = 0, srv = 0;
  for (auto &q : cell) {
    req += q.requests;
    srv += q.served;
  }
  const bool ok = bad == 0 && req == srv && rdv.members == 0 && rdv.arrived == 0;
  std::printf("pass rendezvous  C=%d  requests %ld served %ld  launches %ld  partial batches %llu  %s\n", C, req, srv, launches,
              rdv.partial, ok ? "ok" : "FAILED");
  return ok ? 0 : 1;
}

}  // namespace

int main(int argc, char **argv) {
  const int rounds = argc > 1 ? std::atoi(argv[1]) : 300;
  int bad = 0;
  for (int C : {2, 3, 4, 6, 8}) {
    bad |= run_rounds(C, rounds, 1234u + (unsigned)C, false);
    bad |= run_rounds(C, rounds, 4321u + (unsigned)C, true);
    bad |= run_staged_rounds(C + 1, rounds, 777u + (unsigned)C);
    bad |= run_pass_rounds(C, rounds * 4, 999u + (unsigned)C);
  }
  std::printf(bad ? "FAILED\n" : "sync harness OK\n");
  return bad;
}
