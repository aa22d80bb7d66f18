// bessx_kchunks.cpp -- sequential_path (src/path.cpp:25-132) of ONE LM problem as several chunk chains at once.
//
// Candidate k of the path starts from candidate k-1's model (src/path.cpp:60-64): one chain of dependent fits whose
// kernels are single workgroups -- one compute unit of 256 busy between the passes over X that bring new Gram columns.
// The multi-GPU run cuts the chain into chunks and STITCHES them back into the single chain
// (bess_amd.dist.StitchedKPath); here the same is done inside one device, on ONE Gram column cache:
//   1. a coarse warm-start chain over the sparsity levels in front of the chunks (C - 1 fits): it fills the cache with
//      nearly every column the path will ask for (its fills speculate 64 columns wide) and leaves every chunk the model
//      it starts from;
//   2. the C chunks side by side, each on a fit context of its own (stream, host thread, control block, scores, solve
//      work space) that reads the shared cache.  A chunk that needs a column the cache lacks fills it while every
//      other chain stands still between two candidates (the rendezvous below): nobody reads the slot map while it is
//      rewritten;
//   3. the stitch: chunk r re-fits its first candidates warm from chunk r-1's last model until one coincides with its
//      own (same support, coefficients to 1e-9) -- from there on the two chains are the same chain -- and replaces what
//      was walked before; in rounds, until no chunk's last model changed.
// The candidates returned are the single chain's: same supports, iteration counts and criteria
// (tests/test_kchunks_gpu.py; at full size tests/test_fullsize_gpu.py).
#include "bessx_host.h"

namespace bessx {

// The chains' shared passes over X (round 6; the protocol is bessx_sync.h: PassRendezvous).  The chunk chains of the
// streaming forms -- LM with score_mode 1, logistic, Poisson, Cox -- need X^T v (or the Cox score) for a vector of their own
// at every PDAS iteration.  Issued per chain every one of them streams the whole of X (round 5: 153 / 220 / 241 / 251
// passes per path of the four full-size configs, the passes of three or four chains side by side at 0.44-0.65 of the HBM
// roof each).  Here a chain hands its vector set to the owner's next MULTI-CHAIN launch (k_xtv_mc, k_cox_score1p_mc: X is
// loaded once, every active chain's sums are formed from it, bitwise those of a launch of its own) on the PASS STREAM:
// the launch waits for an event every participant records on its own stream (its vectors are ready), every participant's
// stream waits for the event recorded behind the launch.  A pass then serves every chain that is at its score pass --
// the chains fall into step at the passes, and the count of chains is no longer bounded by what a pass per chain costs.
// Optionally two GROUPS of chains alternate on the pass stream (chain r belongs to group r mod 2; test hook
// kchunks_pass_groups=2; default 1): while one group's pass runs, the other group's chains do what lies between two passes
// -- selection, Gram, solve, residual, the IRLS / Newton steps, the host's read-back and the launches of the next slot.
// Measured slower than the single lock-step (see sequential_path_chunked): kept as a hook.
struct SharedPass {
  static constexpr int GROUPS = 2;
  PassRendezvous rdv[GROUPS];
  int groups = 1;      // groups of this path
  bool on = false;     // this path's chains share their passes
  int kind = 0;        // 0: k_xtv_mc, one vector; 1: k_xtv_mc, two vectors (GLM); 2: k_cox_score1p_mc
  int cap = XTV_MC_MAX;
  hipStream_t st = nullptr;
  std::mutex launch_mu;  // one group's (waits, launch, records) on the pass stream at a time
  hipEvent_t ev_in[GROUPS][XTV_MC_MAX] = {};
  static constexpr int RING = 8;
  hipEvent_t ev_out[GROUPS][RING] = {};
  XtvMc xq[GROUPS] = {};
  CoxMc cq[GROUPS] = {};
  int rc = 0;  // first launch error (reported by the path)
  // statistics: HIP events around every launch on the pass stream, and how many gates were open in it
  bool timing = false;
  std::vector<hipEvent_t> tev;
  size_t tused = 0;
  int *ran = nullptr;  // device, RAN_CAP ints
  static constexpr int RAN_CAP = 8192;
  int launches = 0;
};

struct KChains {
  std::vector<bessx_session *> ctx;
  FoldPool pool;
  bool pool_started = false;
  SharedPass sp;
  FillRendezvous rdv;  // the fill rendezvous (bessx_sync.h: built and hammered under ThreadSanitizer, tools/tsan)
  // candidates a chunk chain has stored so far / whether it has ended (read by its predecessor's early stitch)
  std::atomic<int> progress[10];
  std::atomic<int> ended[10];  // 0 running, 1 ended, 2 failed
  // test hook kchunks_log=1: what every chain did when (to stderr at the end of the path)
  bool log_on = false;
  std::mutex log_mu;
  std::chrono::steady_clock::time_point log_t0;
  struct LogRow {
    double ms;
    int chain;
    const char *what;
    int a, b;
  };
  std::vector<LogRow> log;
  // merged launches over the chains (mc_run_chunks): device copies of the chains' descriptions and states, the chunk's
  // levels, the per-candidate records; pinned status block and its sequence number
  McChain *mc_chains = nullptr;
  McState *mc_states = nullptr;
  int *mc_seq = nullptr, *mc_rec_i = nullptr, *mc_rec_A = nullptr;
  double *mc_rec_d = nullptr, *mc_rec_b = nullptr;
  size_t mc_cap_cand = 0, mc_cap_cells = 0;
  unsigned char *mc_status_h = nullptr;
  unsigned long long *mc_flag = nullptr, mc_seq_no = 0;
};

static void mc_free(KChains *k) {
  if (k->mc_chains) (void)hipFree(k->mc_chains);
  if (k->mc_states) (void)hipFree(k->mc_states);
  if (k->mc_seq) (void)hipFree(k->mc_seq);
  if (k->mc_rec_i) (void)hipFree(k->mc_rec_i);
  if (k->mc_rec_A) (void)hipFree(k->mc_rec_A);
  if (k->mc_rec_d) (void)hipFree(k->mc_rec_d);
  if (k->mc_rec_b) (void)hipFree(k->mc_rec_b);
  if (k->mc_status_h) (void)hipHostFree(k->mc_status_h);
  if (k->mc_flag) (void)hipHostFree(k->mc_flag);
  k->mc_chains = nullptr;
  k->mc_states = nullptr;
  k->mc_seq = k->mc_rec_i = k->mc_rec_A = nullptr;
  k->mc_rec_d = k->mc_rec_b = nullptr;
  k->mc_status_h = nullptr;
  k->mc_flag = nullptr;
  k->mc_cap_cand = k->mc_cap_cells = 0;
}

// between two candidates of a chunk chain: if another chain waits to fill, drain this chain's stream and stand still
void kchains_safe_point(bessx_session *c) {
  KChains *k = c->kch_owner ? c->kch_owner->kch : nullptr;
  if (!k) return;
  k->rdv.safe_point([c] { (void)hipStreamSynchronize(c->st); });
}

// a parked chain asks for the cache: returns 0 when it may fill (every other chain stands still), 1 when another chain
// filled while this one waited (the right is held all the same: look the columns up again), -1 when the run was abandoned
int kchains_fill_begin(bessx_session *c) {
  FillRendezvous &rdv = c->kch_owner->kch->rdv;
  const int waited = rdv.fill_begin([c] { (void)hipStreamSynchronize(c->st); }, c->kch_gen_seen);
  if (waited >= 0) c->kch_gen_seen = rdv.generation();  // (the right is held: no fill ends before this chain's does)
  return waited;
}

void kchains_fill_end(bessx_session *c, bool filled) { c->kch_owner->kch->rdv.fill_end(filled); }

void kchains_log(bessx_session *c, const char *what, int a, int b) {
  bessx_session *o = c->kch_owner ? c->kch_owner : c;
  KChains *k = o->kch;
  if (!k || !k->log_on) return;
  int id = -1;
  for (size_t i = 0; i < k->ctx.size(); i++)
    if (k->ctx[i] == c) id = (int)i;
  const double ms = std::chrono::duration<double>(std::chrono::steady_clock::now() - k->log_t0).count() * 1e3;
  std::lock_guard<std::mutex> lk(k->log_mu);
  k->log.push_back({ms, id, what, a, b});
}

// a chain context has stored its n-th candidate (rows < n of its result arrays are final)
void kchains_progress(bessx_session *c, int n) {
  KChains *k = c->kch_owner ? c->kch_owner->kch : nullptr;
  if (!k || c->kch_index < 0 || c->kch_index >= 10) return;
  if (k->ended[c->kch_index].load(std::memory_order_acquire) != 0) return;  // (a refit on the context of an ended chunk)
  k->progress[c->kch_index].store(n, std::memory_order_release);
}

bool kchains_staged(const bessx_session *c) {
  return c->kch_owner && c->kch_owner->kch && c->kch_owner->kch->rdv.concurrent;  // (set before the round's chains start)
}

unsigned long long kchains_generation(bessx_session *c) { return c->kch_owner->kch->rdv.generation(); }

static void kchains_round(KChains *k, int chains, bool staged = false) { k->rdv.round(chains, staged); }

// ---- shared passes -------------------------------------------------------------------------------------------------
static void sp_free(SharedPass &sp) {
  if (sp.st) ctx_stream_destroy(sp.st);
  for (auto &grp : sp.ev_in)
    for (auto &e : grp)
      if (e) (void)hipEventDestroy(e);
  for (auto &grp : sp.ev_out)
    for (auto &e : grp)
      if (e) (void)hipEventDestroy(e);
  for (auto &e : sp.tev) (void)hipEventDestroy(e);
  if (sp.ran) (void)hipFree(sp.ran);
  sp.st = nullptr;
  sp.ran = nullptr;
  sp.tev.clear();
  for (auto &grp : sp.ev_in)
    for (auto &e : grp) e = nullptr;
  for (auto &grp : sp.ev_out)
    for (auto &e : grp) e = nullptr;
}

// stream, events and the gate counters of the pass stream; false: not to be had (the chains then keep their own passes)
static bool sp_prepare(bessx_session *s, SharedPass &sp) {
  if (sp.st) return true;
  if (!ctx_stream_create(s->device, &sp.st)) {
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess ||
        hipStreamCreateWithPriority(&sp.st, hipStreamNonBlocking, hi) != hipSuccess) {
      (void)hipGetLastError();
      sp.st = nullptr;
      return false;
    }
  }
  bool ok = true;
  for (auto &grp : sp.ev_in)
    for (auto &e : grp) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
  for (auto &grp : sp.ev_out)
    for (auto &e : grp) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipMalloc(reinterpret_cast<void **>(&sp.ran), SharedPass::RAN_CAP * sizeof(int)) == hipSuccess;
  if (!ok) {
    (void)hipGetLastError();
    sp_free(sp);
    return false;
  }
  return true;
}

// (under the rendezvous' lock) launch the n requests collected as batch g
static void sp_launch(bessx_session *o, SharedPass &sp, int grp, int n, unsigned long long g) {
  std::lock_guard<std::mutex> lk(sp.launch_mu);
  hipError_t e = hipSuccess;
  for (int i = 0; i < n && e == hipSuccess; i++) e = hipStreamWaitEvent(sp.st, sp.ev_in[grp][i], 0);
  int *ran = nullptr;
  hipEvent_t ea = nullptr, eb = nullptr;
  if (sp.timing && sp.launches < SharedPass::RAN_CAP) {
    while (sp.tev.size() < sp.tused + 2) {
      hipEvent_t ev = nullptr;
      if (hipEventCreate(&ev) != hipSuccess) break;
      sp.tev.push_back(ev);
    }
    if (sp.tev.size() >= sp.tused + 2) {
      ea = sp.tev[sp.tused];
      eb = sp.tev[sp.tused + 1];
      ran = sp.ran + sp.tused / 2;
      sp.tused += 2;
    }
  }
  if (ea && e == hipSuccess) e = hipEventRecord(ea, sp.st);
  if (e == hipSuccess) {
    if (sp.kind == 2) {
      sp.cq[grp].nc = n;
      sp.cq[grp].ran = ran;
      e = launch_cox_score1p_mc(o->X, o->ld, o->p, o->U, o->nrb, sp.cq[grp], sp.st);
    } else {
      sp.xq[grp].nc = n;
      sp.xq[grp].ran = ran;
      e = launch_xtv_mc(o->X, o->ld, o->p, o->U, sp.xq[grp], sp.kind == 1, sp.st);
    }
  }
  if (eb && e == hipSuccess) e = hipEventRecord(eb, sp.st);
  if (e == hipSuccess) e = hipEventRecord(sp.ev_out[grp][g % SharedPass::RING], sp.st);
  sp.launches++;
  if (e != hipSuccess && sp.rc == 0) {
    sp.rc = (int)e;
    (void)hipGetLastError();
  }
}

bool shared_pass_applies(const bessx_session *c) {
  return c && c->kch_owner && c->kch_owner->kch && c->kch_owner->kch->sp.on && c->kch_sp_member;
}

// The pass over X of slot `slot` of chain context c's fit: its vectors go into the owner's next multi-chain launch.
// Returns when that launch is queued and c's stream is ordered behind it.  v2 / part2: the second accumulator (GLM);
// cox != nullptr: the one-pass Cox score (part = its output planes).
int shared_pass_submit(bessx_session *c, const double *v, const double *v2, double *part, double *part2,
                       const CoxBufs *cox, const FitCtrl *ctrl, int slot) {
  bessx_session *o = c->kch_owner;
  SharedPass &sp = o->kch->sp;
  const int grp = c->kch_sp_group;
  hipError_t e_in = hipSuccess, e_out = hipSuccess;
  sp.rdv[grp].submit(
      [&](int i) {
        if (cox) {
          CoxMc &q = sp.cq[grp];
          q.TH[i] = cox->TH;
          q.CU[i] = cox->CU;
          q.CV[i] = cox->CV;
          q.C2[i] = cox->C2;
          q.out[i] = part;
          q.ctrl[i] = ctrl;
          q.slot[i] = slot;
        } else {
          XtvMc &q = sp.xq[grp];
          q.v[i] = v;
          q.v2[i] = v2;
          q.part[i] = part;
          q.part2[i] = part2;
          q.ctrl[i] = ctrl;
          q.slot[i] = slot;
        }
        e_in = hipEventRecord(sp.ev_in[grp][i], c->st);  // everything this chain has queued so far: its vectors are ready
      },
      [&](int n, unsigned long long g) { sp_launch(o, sp, grp, n, g); },
      [&](unsigned long long g) { e_out = hipStreamWaitEvent(c->st, sp.ev_out[grp][g % SharedPass::RING], 0); });
  if (e_in != hipSuccess || e_out != hipSuccess || sp.rc != 0)
    return fail(BESSX_ERR_HIP, std::string("shared pass over X: ") +
                                   hipGetErrorString(e_in != hipSuccess ? e_in : (e_out != hipSuccess ? e_out : (hipError_t)sp.rc)));
  return 0;
}

// a chain context's thread stops / starts taking part in the shared passes (end of its job; a wait that is not for the
// device).  Leaving while everybody else waits launches their batch.
static void sp_leave(bessx_session *c) {
  if (!c->kch_sp_member) return;
  bessx_session *o = c->kch_owner;
  SharedPass &sp = o->kch->sp;
  c->kch_sp_member = false;
  const int grp = c->kch_sp_group;
  sp.rdv[grp].leave([&](int n, unsigned long long g) { sp_launch(o, sp, grp, n, g); });
}

static void sp_join(bessx_session *c) {
  SharedPass &sp = c->kch_owner->kch->sp;
  if (!sp.on || c->kch_sp_member) return;
  sp.rdv[c->kch_sp_group].join();
  c->kch_sp_member = true;
}

// a round of `members` chain threads starts (before any of them runs)
static void sp_round(KChains *k, const std::vector<bessx_session *> &who) {
  SharedPass &sp = k->sp;
  for (bessx_session *c : k->ctx) c->kch_sp_member = false;
  if (!sp.on) return;
  int cnt[SharedPass::GROUPS] = {};
  for (size_t i = 0; i < who.size(); i++) {
    who[i]->kch_sp_member = true;
    who[i]->kch_sp_group = (int)(i % (size_t)sp.groups);
    cnt[who[i]->kch_sp_group]++;
  }
  for (int g = 0; g < SharedPass::GROUPS; g++) sp.rdv[g].reset(cnt[g]);
}

// after the path (every chain's stream is idle): the launches' times into the session's score-pass statistics
static int sp_collect(bessx_session *s, SharedPass &sp) {
  if (!sp.st) return 0;
  HIPX(hipStreamSynchronize(sp.st));
  const int nl = (int)(sp.tused / 2);
  if (sp.timing && nl > 0) {
    std::vector<int> ran((size_t)nl, 0);
    HIPX(hipMemcpy(ran.data(), sp.ran, (size_t)nl * sizeof(int), hipMemcpyDeviceToHost));
    for (int i = 0; i < nl; i++) {
      if (ran[(size_t)i] <= 0) continue;  // (every gate closed: the launch fell through)
      float ms = 0.f;
      HIPX(hipEventElapsedTime(&ms, sp.tev[(size_t)2 * i], sp.tev[(size_t)2 * i + 1]));
      s->k1_seconds += (double)ms * 1e-3;
      s->k1_launches += 1;
      s->k1_bytes += 8.0 * (double)s->n * (double)s->p;
      s->sp_chain_slots += ran[(size_t)i];
    }
  }
  s->sp_launches += sp.launches;
  for (auto &r : sp.rdv) {
    s->sp_partial += (long long)r.partial;
    r.partial = 0;
  }
  sp.tused = 0;
  sp.launches = 0;
  return 0;
}

static void kchains_leave(KChains *k, bool failed) { k->rdv.leave(failed); }

void kchains_free(bessx_session *s) {
  if (!s || !s->kch) return;
  KChains *k = s->kch;
  if (k->pool_started) k->pool.stop();
  mc_free(k);
  sp_free(k->sp);
  if (s->kch_slot_w) (void)hipFree(s->kch_slot_w);
  s->kch_slot_w = nullptr;
  if (s->kch_fill_st) ctx_stream_destroy(s->kch_fill_st);
  s->kch_fill_st = nullptr;
  for (bessx_session *c : k->ctx) chain_ctx_free(c);
  if (!k->pool.broken) delete k;  // (a broken pool's threads may still touch it: leaked on purpose)
  s->kch = nullptr;
}

// how many chunk chains for this path (1 = the single chain)
static int chains_for(const bessx_session *s, int ns, bool link = false, bool link_warm = false) {
  int C = s->kpath_chains;
  if (C == 0 && s->kch_auto_off) return 1;  // (this session's chunks do not merge: see the stitch's budget)
  if (C == 0 && link && s->model_type == 1 && ns < 96) {
    // a link of a longer chain (a rank's chunk of a multi-GPU k-path): shorter, so fewer chains
    C = (ns >= 40 && s->p >= 2048) ? 2 : 1;
    // ... behind lead fits (bessx_path_chain.lead_levels: the link starts warm on a filled cache) short links pay too
    if (link_warm && s->cov_mode && s->p >= 2048) C = ns >= 40 ? 4 : (ns >= 24 ? 3 : (ns >= 16 ? 2 : 1));
  } else if (C == 0 && link && !(s->model_type == 1 && s->cov_mode) && ns < 48) {
    // a short link of a multi-GPU k-path of the families whose every PDAS iteration is a pass over X: with shared passes
    // (round 6) 12-23 levels pay as 2-3 chains -- rank 4 of 8 of configs[4] (levels 77..95, cold) 208 / 186 / 154 ms
    // with 1 / 2 / 3 chains (tools/cox_link_probe.py); every rank of 8: slowest 358 -> 320 ms.  Longer links do not
    // (the 37-level links of 4 ranks: 267 / 178 / 277 ms either way, and the last one -- levels 114..150, beyond the
    // planted support, where cold chunks wander and the stitch gives up -- 571 -> 740 ms): one chain from 24 levels on
    const char *esh = test_hook("kchunks_shared_pass");
    const bool big = (double)s->n * s->p >= 1e8 && (!esh || std::atoi(esh) != 0) && ctx_streams_own_queue(s->device);
    C = (big && ns < 24) ? (ns >= 18 ? 3 : (ns >= 12 ? 2 : 1)) : 1;
  } else if (C == 0) {
    // automatic: long paths on wide designs.  How many chains pay depends on how many hardware queues the HIP runtime
    // gives the process' streams (GPU_MAX_HW_QUEUES, default 4; read when the runtime starts): configs[1], 18.6 ms as
    // one chain -- 4 queues: 2 chains 16.2 ms, 3 and more lose (21.2 / 19.5 ms: streams share queues and wait for each
    // other); 8 queues: 3 chains 14.9 ms, 4 chains 12.7 ms, 6 chains 16.8 ms (tools/kchunks_bench.py)
    // Round 5: the chains' streams are created with a hardware queue of their own, outside that pool
    // (ctx_stream_create), so the count no longer depends on the variable; where such streams cannot be had the old
    // rule stands.
    const char *q = std::getenv("GPU_MAX_HW_QUEUES");
    const int queues = ctx_streams_own_queue(s->device) ? 8 : (q ? std::atoi(q) : 4);
    if (s->model_type == 1 && s->cov_mode)
      C = (ns >= 96 && s->p >= 2048) ? (queues >= 8 ? 4 : 2) : 1;
    else if (s->model_type == 1)
      // round 5: the STREAMING form of the LM score pass as chunk chains, like the IRLS / Newton families -- every PDAS
      // iteration is a pass over X, and one chain's selection, Gram panel, solve and residual run beside another's pass
      // (tools/streaming_chains_bench.py, configs[1]: 190.1 ms as one chain, 168.7 / 173.5 / 162.3 ms with 2 / 3 / 4)
    {
      C = (ns >= 48 && (double)s->n * s->p >= 1e8) ? (queues >= 8 ? 4 : 2) : 1;
      // round 6, shared passes + the light confirming iteration: 8 chains share a pass at 0.75 ms (4: 0.71) -- configs[1]
      // 79 ms per path with 8 chains, 87 with 4 (tools/shared_pass_sweep.py)
      const char *esh = test_hook("kchunks_shared_pass");
      if (C == 4 && (!esh || std::atoi(esh) != 0) && ns >= 128) C = 8;
    } else {
      // logistic / Poisson / Cox: one chain's IRLS or Newton steps (small kernels) run beside another's pass over X
      // (tools/glm_two_chains_probe.py: logistic at full size 155 -> 120 ms with 3 chains, Cox 1.86 -> 1.60 s)
      C = (ns >= 48 && (double)s->n * s->p >= 1e8) ? (queues >= 8 ? 3 : 2) : 1;
      // round 6, shared passes (a pass serves every chain that is at its score pass; tools/shared_pass_sweep.py, full
      // size): logistic 3 / 4 / 6 / 8 chains 85-90 / 90 / 87.5 / 83 ms, Poisson 92 / 90 / 82 / 84 ms per path -- 6; Cox
      // 2 / 3 / 4 chains 1.21 / 1.04 / 1.11 s -- 3
      const char *esh = test_hook("kchunks_shared_pass");
      if (C == 3 && (!esh || std::atoi(esh) != 0) && s->model_type != 4 && ns >= 96) C = 6;
    }
  }
  // (at least 8 candidates per chain; 6 for the families whose every iteration is a pass over X: a short link of a
  // multi-GPU k-path there still pays to share its passes)
  const bool streaming = !(s->model_type == 1 && s->cov_mode);
  return std::max(1, std::min(std::min(C, 8), ns / (streaming ? 6 : 8)));
}

bool kchunks_apply(const bessx_session *s, const int *seq, int ns, int nl, int is_cv, const bessx_path_chain *chain) {
  if (!s || s->kch_owner || s->parent || is_cv || nl != 1) return false;
  // a link of a longer chain (bessx_session_sequential_path_chain) qualifies when it only starts from a given model
  // -- the stitch refits, which stop at the first candidate equal to the caller's, are short and stay one chain
  if (chain && (chain->stop_support || chain->init_len > s->cap)) return false;
  if (s->grouped || !s->warm_start || s->trace.on || s->cv_shared || !s->publish || s->fill_hook) return false;
  if (s->model_type == 1 && s->cov_mode) {
    if (!s->chain) return false;
    if (s->cov_C < (s->p + 31) / 32 * 32 + COV_R) return false;  // the cache must hold every column: it is never started over
  } else if (s->K > 0) {
    return false;  // (sessions with CV folds keep per-row-set state the chain contexts do not own)
  }
  if (chains_for(s, ns, chain != nullptr, chain && chain->init_len > 0 && chain->keep_caches) < 2) return false;
  int top = 0;
  for (int i = 0; i < ns; i++) {
    if (seq[i] < 1 || (i && seq[i] <= seq[i - 1])) return false;  // ascending levels: every chunk continues its predecessor
    top = std::max(top, seq[i]);
  }
  if (top > 254 || top > s->cap) return false;  // (the register-resident solvers)
  return !(s->model_type == 1 && s->cov_mode) || top + COV_R + s->cov_spec <= s->cov_C;
}

namespace {

struct ChunkRun {  // one chunk's candidates, as sequential_path stores them
  int lo = 0, hi = 0, width = 0;
  std::vector<int> T0, iters, support;
  std::vector<double> lam, loss, ic, coef0, beta, best_beta;
  bessx_path_result res = {};
  bessx_path_chain chain = {};
  std::vector<int> init_idx, last_idx;
  std::vector<double> init_val, last_val;
  double init_coef0 = 0.0, last_coef0 = 0.0;
  int last_len = 0;
  int rc = 0;
  std::string err;  // (the message of a failure on a host thread: bessx_last_error is per thread)
  long long fits = 0;

  void shape(int lo_, int hi_, int width_, int p_full) {
    lo = lo_;
    hi = hi_;
    width = width_;
    const size_t m = (size_t)std::max(1, hi - lo);
    T0.assign(m, 0);
    iters.assign(m, 0);
    lam.assign(m, 0.0);
    loss.assign(m, 0.0);
    ic.assign(m, 0.0);
    coef0.assign(m, 0.0);
    support.assign(m * width, -1);
    beta.assign(m * width, 0.0);
    best_beta.assign((size_t)p_full, 0.0);
    last_idx.assign((size_t)width, 0);
    last_val.assign((size_t)width, 0.0);
  }
  void bind(bessx_path_result *r) {
    *r = bessx_path_result();
    r->beta = best_beta.data();
    r->capacity = hi - lo;
    r->cand_T0 = T0.data();
    r->cand_lambda = lam.data();
    r->cand_iters = iters.data();
    r->cand_train_loss = loss.data();
    r->cand_ic = ic.data();
    r->cand_coef0 = coef0.data();
    r->cand_support = support.data();
    r->cand_beta = beta.data();
    r->max_T0 = width;
  }
};

// the context's own state starts over (the shared cache stays as it is); keep_model: except the model its last fit left
// on the device -- the early stitch continues from exactly that model, without uploading it again
int context_begin(bessx_session *c, bool keep_model = false) {
  if (int rc = settle_device_chain(c)) return rc;
  if (!keep_model) {
    for (auto &q : c->cache) q.valid = q.model_only = false;
    c->dev_state_rs = -1;
  }
  c->trace.clear();
  c->metric_depth = 0;
  c->n_fits = 0;
  c->n_iters = 0;
  return 0;
}

}  // namespace

// every chain context's stream idle (after a failed run a context may still have launches queued)
void kchains_quiesce(bessx_session *s) {
  if (!s || !s->kch) return;
  for (bessx_session *c : s->kch->ctx) (void)hipStreamSynchronize(c->st);
}

// Contexts and host threads for the path's chains.  Non-zero when they cannot be had (no memory for the contexts, host
// threads lost in an earlier call): the caller then runs the path as one chain -- slower, same result.
int kchunks_prepare(bessx_session *s, int ns, bool link, bool link_warm) {
  const int C = chains_for(s, ns, link, link_warm);
  if (hipSetDevice(s->device) != hipSuccess) return 1;
  if (!s->kch) s->kch = new KChains();
  KChains *k = s->kch;
  if (k->pool.broken) return 1;
  k->rdv.deadline_s = s->wait_deadline_s;
  // (LM, covariance form, only with the test hook kchunks_pipeline=1: one more context and host thread -- the coarse chain
  // beside the chunks, see below.  Rounds 4-5 created them for every session: an idle thread woken for every round and
  // a context's buffers and stream held for nothing)
  const char *epl = test_hook("kchunks_pipeline");
  const int extra = (s->model_type == 1 && s->cov_mode && epl && std::atoi(epl) != 0) ? 1 : 0;
  while ((int)k->ctx.size() < C + extra) {
    bessx_session *c = nullptr;
    if (chain_ctx_create(s, &c) != 0) {
      (void)hipGetLastError();
      return 1;
    }
    c->kch_index = (int)k->ctx.size();
    k->ctx.push_back(c);
  }
  if (!k->pool_started || (int)k->pool.th.size() < C - 1 + extra) {  // one host thread per chain; the caller is one of them
    if (k->pool_started) k->pool.stop();
    k->pool.quit = false;  // (a pool started again: no job of the previous threads' numbering is left to run)
    k->pool.ticket = 0;
    k->pool.ticket_hint.store(0);
    k->pool.job = nullptr;
    {
      const int dev = s->device;
      k->pool.start(C - 1 + extra, [dev] { (void)hipSetDevice(dev); });
    }
    k->pool_started = true;
  }
  return 0;
}

// ----------------------------------------------------------------------------------------------------------------
// The chunk phase as merged launches on ONE stream (bessx_dev.h: McChain).  The chains' own streams and host threads are
// not used here; the stitch that follows runs as before.  Returns 0 with every chunk's candidates in run[r] (a chain the
// device could not finish by itself is finished through sequential_path on its context: same candidates), > 0 on a
// failure, < 0 when the engine does not apply to this path (the caller then runs the chunks on their own streams).
// ----------------------------------------------------------------------------------------------------------------
namespace {

struct McHostStatus {
  McState st;
  FitCtrl ctrl;
};
static_assert(sizeof(McHostStatus) == 192, "192 bytes per chain in the pinned status block");

int mc_wait(bessx_session *s, KChains *k, unsigned long long want) {
  volatile unsigned long long *flag = k->mc_flag;
  std::chrono::steady_clock::time_point t0;
  bool timed = false;
  for (unsigned spins = 1;; spins++) {
    if (*flag >= want) break;
    if ((spins & 0x3fff) == 0) {
      const auto now = std::chrono::steady_clock::now();
      if (!timed) {
        t0 = now;
        timed = true;
      } else if (std::chrono::duration<double>(now - t0).count() > s->wait_deadline_s) {
        return fail(BESSX_ERR_HIP, "chunk chains (merged launches): no status block from the device within " +
                                       std::to_string(s->wait_deadline_s) + " s (BESSX_WAIT_TIMEOUT_S)");
      }
    }
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return 0;
}

// |y - X beta|^2 of a model by one pass over its columns (what algorithm_fit does when the solve's loss terms cancel)
int mc_sse_by_residual(bessx_session *c, hipStream_t st, const int *idx, const double *val, int T0, double coef0, double *out) {
  int *st_idx = reinterpret_cast<int *>(c->stage_h);
  double *st_val = reinterpret_cast<double *>(c->stage_h + (size_t)c->capA * sizeof(int));
  for (int i = 0; i < T0; i++) {
    st_idx[i] = idx[i];
    st_val[i] = val[i];
  }
  HIPX(hipMemcpyAsync(c->init_idx_d, st_idx, T0 * sizeof(int), hipMemcpyHostToDevice, st));
  HIPX(hipMemcpyAsync(c->init_val_d, st_val, T0 * sizeof(double), hipMemcpyHostToDevice, st));
  HIPX(launch_resid_lm(c->X, c->ld, c->n, c->y, c->mask[0], c->ctrl, 0, c->init_idx_d, c->init_val_d, c->tmpv, c->sse, st, 3,
                       T0, coef0));
  std::vector<double> part((size_t)2 * c->n_sse_blk);
  HIPX(hipMemcpyAsync(part.data(), c->sse, part.size() * sizeof(double), hipMemcpyDeviceToHost, st));
  HIPX(hipStreamSynchronize(st));
  double tr = 0.0;
  for (int b = 0; b < c->n_sse_blk; b++) tr += part[2 * b];
  *out = tr;
  return 0;
}

int mc_run_chunks(bessx_session *s, KChains *k, const int *seq, int ns, int C, const std::vector<int> &bounds,
                  std::vector<ChunkRun> &run, double lambda, int ic_type, int width) {
  if (s->model_type != 1 || !s->cov_mode || !s->fuse || !s->fuse_sel || !s->cov_cg || !s->cg_by_rows || C > 8) return -1;
  if (!mc_applies(s->p, seq[ns - 1])) return -1;
  // Measured on configs[1] (tools/kchunks_bench.py, profiles/README.md, round 5): NOT faster than the chains on streams
  // of their own -- 12.4-12.6 ms per path against 12.0 with 4 chains, 12.7 with 8 -- because the merged launches run the
  // chains in lock-step: every step costs what the chain with the largest systems needs (48 us for selection, record,
  // selection and a 200-unknown solve) and the chunk phase is (candidates per chain) x that.  Kept behind the test hook
  // kchunks_merged=1, exercised by tests/test_kchunks_gpu.py; the default is the stream per chain.
  const char *mc_on = test_hook("kchunks_merged");
  if (!(mc_on && std::string(mc_on) == "1")) return -1;
  const int p = s->p;
  hipStream_t st = s->st;
  // ---- buffers
  if (!k->mc_chains) {
    HIPX(hipMalloc(reinterpret_cast<void **>(&k->mc_chains), 8 * sizeof(McChain)));
    HIPX(hipMalloc(reinterpret_cast<void **>(&k->mc_states), 8 * sizeof(McState)));
    HIPX(hipHostMalloc(reinterpret_cast<void **>(&k->mc_status_h), 8 * sizeof(McHostStatus)));
    HIPX(hipHostMalloc(reinterpret_cast<void **>(&k->mc_flag), 128));
    k->mc_flag[0] = 0ull;
    k->mc_seq_no = 0;
  }
  if ((size_t)ns > k->mc_cap_cand || (size_t)ns * width > k->mc_cap_cells) {
    if (k->mc_seq) (void)hipFree(k->mc_seq);
    if (k->mc_rec_i) (void)hipFree(k->mc_rec_i);
    if (k->mc_rec_d) (void)hipFree(k->mc_rec_d);
    if (k->mc_rec_A) (void)hipFree(k->mc_rec_A);
    if (k->mc_rec_b) (void)hipFree(k->mc_rec_b);
    k->mc_seq = k->mc_rec_i = k->mc_rec_A = nullptr;
    k->mc_rec_d = k->mc_rec_b = nullptr;
    k->mc_cap_cand = k->mc_cap_cells = 0;
    HIPX(dmalloc(&k->mc_seq, (size_t)ns));
    HIPX(dmalloc(&k->mc_rec_i, (size_t)ns * MC_REC_I));
    HIPX(dmalloc(&k->mc_rec_d, (size_t)ns * MC_REC_D));
    HIPX(dmalloc(&k->mc_rec_A, (size_t)ns * width));
    HIPX(dmalloc(&k->mc_rec_b, (size_t)ns * width));
    k->mc_cap_cand = (size_t)ns;
    k->mc_cap_cells = (size_t)ns * width;
  }
  if (!s->fill_ctrl) {
    HIPX(hipMalloc(reinterpret_cast<void **>(&s->fill_ctrl), sizeof(FitCtrl)));
    HIPX(hipMemset(s->fill_ctrl, 0, sizeof(FitCtrl)));
    HIPX(hipHostMalloc(reinterpret_cast<void **>(&s->fill_ctrl_h), sizeof(FitCtrl)));
  }
  HIPX(hipMemcpyAsync(k->mc_seq, seq, (size_t)ns * sizeof(int), hipMemcpyHostToDevice, st));
  HIPX(hipMemsetAsync(k->mc_rec_i, 0, (size_t)ns * MC_REC_I * sizeof(int), st));
  // ---- the chains: their contexts' buffers, the start of their first fit
  std::vector<McChain> hc((size_t)C);
  std::vector<McState> hs((size_t)C);
  for (int r = 0; r < C; r++) {
    bessx_session *c = k->ctx[r];
    ChunkRun &q = run[r];
    if (int rc = context_begin(c)) return rc;
    c->timing = s->timing;
    const int lo = q.lo, ncand = q.hi - q.lo, k_init = (int)q.init_idx.size();
    if (ncand < 1 || k_init > c->cap) return fail(BESSX_ERR_ARG, "chunk chains: bad chunk");
    bessx_session::CovCache &cv = c->cov[0];
    int *st_idx = reinterpret_cast<int *>(c->stage_h);
    double *st_val = reinterpret_cast<double *>(c->stage_h + (size_t)c->capA * sizeof(int));
    for (int i = 0; i < k_init; i++) {
      st_idx[i] = q.init_idx[i];
      st_val[i] = q.init_val[i];
    }
    if (k_init) {
      HIPX(hipMemcpyAsync(c->init_idx_d, st_idx, k_init * sizeof(int), hipMemcpyHostToDevice, st));
      HIPX(hipMemcpyAsync(c->init_val_d, st_val, k_init * sizeof(double), hipMemcpyHostToDevice, st));
    }
    HIPX(launch_fit_begin(c->ctrl, seq[lo], k_init, c->init_idx_d, c->init_val_d, q.init_coef0, c->A_cur, c->b_cur,
                          c->beta_dense, p, c->hist, st, c->inA));
    McChain &m = hc[(size_t)r];
    m = McChain();
    m.state = k->mc_states + r;
    m.seq = k->mc_seq + lo;
    m.width = width;
    m.max_iter = c->max_iter;
    m.p = p;
    m.G = cv.G;
    m.xty = c->xty[0];
    m.xtx = c->xtx[0];
    m.n_t = (double)c->n_train[0];
    m.lambda = lambda;
    m.always = c->always;
    m.d_out = c->part_rs[0];
    m.bd = c->bd;
    m.bmm = c->cov_bmm;
    TopkNeed nd = {cov_speculates(c) ? c->bd : nullptr, c->bd2, p, cov_C_dev(c), cv.slot_of, cv.meta, c->cov_fcols,
                   c->ctrl, c->A_cur, c->cov_bmm, (p + 31) / 32, c->inA, 0, 0};
    nd.cm_A_cur = c->A_cur;
    nd.cm_b_cur = c->b_cur;
    nd.cm_beta_dense = c->beta_dense;
    nd.cm_hist = c->hist;
    nd.cm_hist_beta = c->hist_beta;
    nd.cm_hist_coef0 = c->hist_coef0;
    nd.cm_hist_stride = c->hist_stride;
    nd.cm_inA = c->inA;
    nd.commit_on = 1;
    nd.no_restart = 1;
    m.nd = nd;
    m.nd1 = nd;
    m.nd1.inc1 = 1;
    m.nd1.bmm_fresh = 1;
    m.fz = cov_fuse_args(c, 0, seq[lo], false, nullptr);
    m.A_new = c->A_new;
    m.sol = c->sol;
    m.tol = c->cg_tol;
    m.maxit = 64;
    m.rec_i = k->mc_rec_i + (size_t)lo * MC_REC_I;
    m.rec_d = k->mc_rec_d + (size_t)lo * MC_REC_D;
    m.rec_A = k->mc_rec_A + (size_t)lo * width;
    m.rec_b = k->mc_rec_b + (size_t)lo * width;
    McState &z = hs[(size_t)r];
    z = McState();
    z.ncand = ncand;
    z.need_d = 1;
    c->dev_state_rs = -1;
  }
  HIPX(hipMemcpyAsync(k->mc_chains, hc.data(), (size_t)C * sizeof(McChain), hipMemcpyHostToDevice, st));
  HIPX(hipMemcpyAsync(k->mc_states, hs.data(), (size_t)C * sizeof(McState), hipMemcpyHostToDevice, st));
  HIPX(hipStreamSynchronize(st));  // (hc / hs / the staging buffers are the host's again)
  // ---- rounds: a batch of (score pass, selection + solve) pairs, then every chain's state
  int longest = 0;
  for (int r = 0; r < C; r++) longest = std::max(longest, run[r].hi - run[r].lo);
  // Batches of 8 pairs: a chain that parks on a missing column waits for the end of its batch before the host sees it
  // (queueing the whole chunk ahead -- about one pair per candidate -- left a parked chain idle for up to 2.6 ms on
  // configs[1]); the read-back between two batches costs the device ~25 us of 500.
  int batch = 8;
  const McHostStatus *hstat = reinterpret_cast<const McHostStatus *>(k->mc_status_h);
  std::vector<int> takeover((size_t)C, 0);
  for (int round = 0;; round++) {
    if (round > 64 + longest) return fail(BESSX_ERR_NUMERIC, "chunk chains (merged launches): the chains do not end");
    for (int b = 0; b < batch; b++) {
      HIPX(launch_mc_cov_d(k->mc_chains, C, p, st));
      HIPX(launch_mc_sel_cgr(k->mc_chains, C, p, st));
    }
    const unsigned long long want = ++k->mc_seq_no;
    HIPX(launch_mc_status(k->mc_chains, C, k->mc_status_h, k->mc_flag, want, st));
    if (int rc = mc_wait(s, k, want)) return rc;
    bool all = true;
    std::vector<int> parked;
    for (int r = 0; r < C; r++) {
      const McHostStatus &h = hstat[r];
      if (h.st.finished == 1) continue;
      if (h.st.finished == 2) {
        takeover[(size_t)r] = 1;
        continue;
      }
      if (h.st.parked == 1) {
        parked.push_back(r);
        all = false;
      } else if (h.st.parked) {  // a tie, a solve for the Cholesky kernel, a full cache: the proven path finishes this chain
        HIPX(launch_mc_stop(k->mc_chains, r, st));
        takeover[(size_t)r] = 1;
      } else {
        all = false;
      }
    }
    if (!parked.empty()) {
      // ONE fill for every chain parked on missing columns; nobody else runs (one stream): fold_fits_side_by_side's rule
      CovUnion u = {};
      bessx_session *spec_src = nullptr;
      int ub = 0;
      for (int r : parked) {
        bessx_session *c = k->ctx[r];
        u.list[u.nf] = c->A_new;
        u.len[u.nf++] = hstat[r].ctrl.T0;
        ub += hstat[r].ctrl.cov_nmiss;
        if (!spec_src && cov_speculates(c)) spec_src = c;
      }
      if (spec_src)
        HIPX(launch_topk(spec_src->bd2, p, s->cov_spec, spec_src->cov_extras, spec_src->cand, nullptr, 0, st));
      HIPX(launch_cov_fill_union(u, 0, spec_src ? spec_src->cov_extras : nullptr, spec_src ? spec_src->bd2 : nullptr,
                                 s->cov_spec, s->cov_spec / 2, s->cov[0].slot_of, s->cov[0].meta, p, s->cov_fcols,
                                 s->fill_ctrl, st));
      HIPX(hipMemcpyAsync(s->fill_ctrl_h, s->fill_ctrl, sizeof(FitCtrl), hipMemcpyDeviceToHost, st));
      HIPX(hipStreamSynchronize(st));
      s->cov_panel_groups += s->fill_ctrl_h->cov_groups - s->fill_groups_seen;
      s->fill_groups_seen = s->fill_ctrl_h->cov_groups;
      const int ngroups = s->fill_ctrl_h->cov_nfill / COV_R;
      if (ngroups > (ub + s->cov_spec + 2 * COV_R - 1) / COV_R)
        return fail(BESSX_ERR_NUMERIC, "internal error: union fill list longer than its bound (chunk chains)");
      if (int rc = enqueue_cov_fill(s, 0, ngroups, 1, s->fill_ctrl)) return rc;
      if (s->timing) {
        HIPX(hipStreamSynchronize(st));
        if (int rc = cov_collect(s, s->fill_ctrl_h->cov_nfill)) return rc;
      }
      for (int r : parked) HIPX(launch_mc_resume(k->mc_chains, r, st));
      s->kch_chunk_fills++;
    }
    if (all) break;
  }
  // ---- the records
  std::vector<int> rec_i((size_t)ns * MC_REC_I), rec_A((size_t)ns * width);
  std::vector<double> rec_d((size_t)ns * MC_REC_D), rec_b((size_t)ns * width);
  HIPX(hipMemcpyAsync(rec_i.data(), k->mc_rec_i, rec_i.size() * sizeof(int), hipMemcpyDeviceToHost, st));
  HIPX(hipMemcpyAsync(rec_d.data(), k->mc_rec_d, rec_d.size() * sizeof(double), hipMemcpyDeviceToHost, st));
  HIPX(hipMemcpyAsync(rec_A.data(), k->mc_rec_A, rec_A.size() * sizeof(int), hipMemcpyDeviceToHost, st));
  HIPX(hipMemcpyAsync(rec_b.data(), k->mc_rec_b, rec_b.size() * sizeof(double), hipMemcpyDeviceToHost, st));
  HIPX(hipStreamSynchronize(st));
  for (int r = 0; r < C; r++) {
    bessx_session *c = k->ctx[r];
    ChunkRun &q = run[r];
    q.bind(&q.res);
    const int ncand = q.hi - q.lo;
    int good = 0;
    SparseVec last;
    double last_c0 = q.init_coef0;
    last.idx = q.init_idx;
    last.val = q.init_val;
    for (int i = 0; i < ncand; i++) {
      const int g = q.lo + i;
      if (!rec_i[(size_t)g * MC_REC_I + 3]) break;
      const int T0 = rec_i[(size_t)g * MC_REC_I + 0];
      if (T0 != seq[g]) return fail(BESSX_ERR_NUMERIC, "internal error: chunk chain record out of order");
      Candidate cand;
      cand.T0 = T0;
      cand.lambda = lambda;
      cand.beta.idx.assign(rec_A.begin() + (size_t)g * width, rec_A.begin() + (size_t)g * width + T0);
      cand.beta.val.assign(rec_b.begin() + (size_t)g * width, rec_b.begin() + (size_t)g * width + T0);
      cand.coef0 = rec_d[(size_t)g * MC_REC_D + 0];
      cand.iters = rec_i[(size_t)g * MC_REC_I + 1];
      // the loss from the solved system (algorithm_fit, all rows, covariance form), by a pass over the active columns
      // where those terms cancel
      const double yy = c->yy_h[0];
      double tr = yy - rec_d[(size_t)g * MC_REC_D + 1] - lambda * rec_d[(size_t)g * MC_REC_D + 2];
      if (!rec_i[(size_t)g * MC_REC_I + 2] || !(tr > 1e-6 * yy))
        if (int rc = mc_sse_by_residual(c, st, cand.beta.idx.data(), cand.beta.val.data(), T0, cand.coef0, &tr)) return rc;
      c->sparsity_level = T0;
      c->lambda_level = lambda;
      c->beta = cand.beta;
      c->coef0 = cand.coef0;
      c->l = cand.iters;
      c->sse_train = tr;
      c->sse_test = 0.0;
      if (int rc = metric_train_loss(c, &cand.loss)) return rc;
      if (int rc = metric_ic(c, ic_type, 0, &cand.ic)) return rc;
      store_candidate(c, &q.res, cand, false);
      last = cand.beta;
      last_c0 = cand.coef0;
      good++;
    }
    q.fits += good;
    if (good < ncand) {
      // the device stopped in this chunk (takeover): the rest through sequential_path on the context, warm from the last
      // recorded model -- the same chain, by the proven code
      if (!takeover[(size_t)r] && hstat[r].st.finished != 2)
        return fail(BESSX_ERR_NUMERIC, "internal error: chunk chain ended short of its candidates");
      s->kch_takeovers++;
      ChunkRun rest;
      rest.shape(q.lo + good, q.hi, width, s->p_full);
      rest.bind(&rest.res);
      rest.chain = bessx_path_chain();
      rest.chain.init_idx = last.idx.data();
      rest.chain.init_val = last.val.data();
      rest.chain.init_len = (int)last.idx.size();
      rest.chain.init_coef0 = last_c0;
      rest.chain.keep_caches = 1;
      rest.chain.last_idx = rest.last_idx.data();
      rest.chain.last_val = rest.last_val.data();
      rest.chain.last_cap = width;
      if (int rc = context_begin(c)) return rc;
      KChains *own = c->kch_owner ? c->kch_owner->kch : nullptr;
      (void)own;
      kchains_round(k, 1);  // (one chain runs: a fill of its own finds everybody else standing still)
      int rc = sequential_path(c, seq + q.lo + good, ncand - good, &lambda, 1, ic_type, 0, &rest.res, &rest.chain);
      if (rc == 0 && hipStreamSynchronize(c->st) != hipSuccess) rc = fail(BESSX_ERR_HIP, "chunk chain: stream");
      kchains_leave(k, rc != 0);
      if (rc) return rc;
      for (int i = 0; i < ncand - good; i++) {
        const int gi = good + i;  // (copied field by field into the chunk's arrays: they are already de-normalised)
        q.T0[gi] = rest.T0[i];
        q.iters[gi] = rest.iters[i];
        q.lam[gi] = rest.lam[i];
        q.loss[gi] = rest.loss[i];
        q.ic[gi] = rest.ic[i];
        q.coef0[gi] = rest.coef0[i];
        std::copy(rest.support.begin() + (size_t)i * width, rest.support.begin() + (size_t)(i + 1) * width,
                  q.support.begin() + (size_t)gi * width);
        std::copy(rest.beta.begin() + (size_t)i * width, rest.beta.begin() + (size_t)(i + 1) * width,
                  q.beta.begin() + (size_t)gi * width);
      }
      q.res.n_candidates = ncand;
      q.fits += c->n_fits;
      q.last_idx = rest.last_idx;
      q.last_val = rest.last_val;
      q.last_len = rest.chain.last_len;
      q.last_coef0 = rest.chain.last_coef0;
    } else {
      q.last_len = (int)last.idx.size();
      for (int i = 0; i < q.last_len && i < width; i++) {
        q.last_idx[(size_t)i] = last.idx[(size_t)i];
        q.last_val[(size_t)i] = last.val[(size_t)i];
      }
      q.last_coef0 = last_c0;
    }
    for (auto &cc : c->cache) cc.valid = cc.model_only = false;  // (the device state of the context is the engine's, not a fit's of its own)
    c->dev_state_rs = -1;
  }
  s->kch_merged++;
  return 0;
}

}  // namespace

int sequential_path_chunked(bessx_session *s, const int *seq, int ns, double lambda, int ic_type,
                            bessx_path_result *res, bessx_path_chain *link) {
  const int C = chains_for(s, ns, link != nullptr, link && link->init_len > 0 && link->keep_caches);
  HIPX(hipSetDevice(s->device));
  KChains *k = s->kch;
  if (!k || (int)k->ctx.size() < C || k->pool.broken)
    return fail(BESSX_ERR_HIP, "chunk chains: not prepared (kchunks_prepare)");
  // chunk r = candidates [bounds[r], bounds[r + 1]): equal lengths.  (Lengths weighted by the level -- a candidate costs
  // more the larger its solve -- were measured on configs[1]: 1 + k / 40 ... 1 + k / 400 gave 15.0 ... 12.5 ms per path
  // against 12.1 for equal chunks; where the boundaries fall relative to the fills matters more than the balance.)
  // The IRLS / Newton families: the later chunks are shorter (their sub-model systems grow with the level: logistic
  // 123 ms with lengths weighted 1 + k / 160, 129 ms with equal ones).
  std::vector<int> bounds((size_t)C + 1, 0);
  if (s->model_type == 1) {
    for (int r = 1; r <= C; r++) bounds[r] = (int)((long)ns * r / C);
  } else {
    // (chains that share their passes advance in step, one PDAS iteration per shared pass: equal lengths -- with the
    // weights of round 5 the first chunk, the longest, ended 30 % after the others; test hook kchunks_len_div)
    const char *esh = test_hook("kchunks_shared_pass");
    double wdiv = (!esh || std::atoi(esh) != 0) ? 1e9 : 160.0;
    if (const char *ew = test_hook("kchunks_len_div")) wdiv = std::max(1.0, std::atof(ew));
    double tot = 0.0;
    std::vector<double> cum((size_t)ns + 1, 0.0);
    for (int i = 0; i < ns; i++) {
      tot += 1.0 + (double)seq[i] / wdiv;
      cum[(size_t)i + 1] = tot;
    }
    for (int r = 1; r < C; r++) {
      const int b = (int)(std::lower_bound(cum.begin(), cum.end(), tot * r / C) - cum.begin());
      bounds[r] = std::min(std::max(b, bounds[r - 1] + 1), ns - (C - r));
    }
    bounds[C] = ns;
  }
  const int width = res->max_T0 > 0 ? res->max_T0 : seq[ns - 1];
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double>(b - a).count();
  };
  const bool lm_cov = s->model_type == 1 && s->cov_mode;
  // Staged fills (bessx_sync.h: FillRendezvous, concurrent rounds; test hook kchunks_staged=0/1): a chain's fill writes
  // slots no other chain can see until its last launch publishes them, so nobody stands still for a fill.
  // The pipeline (test hook kchunks_pipeline=0/1; needs staged fills and the extra context): the coarse chain runs on a
  // context of its own BESIDE the chunks -- chunk 0 starts with it, chunk r as soon as the coarse fit at its lower
  // boundary is done -- instead of in front of them.
  bool staged = false, pipeline = false;
  if (lm_cov) {
    const char *ev = test_hook("kchunks_staged");
    staged = !ev || std::atoi(ev) != 0;  // (default; 0: round 4's rendezvous -- every chain stands still for a fill)
    const char *ep = test_hook("kchunks_pipeline");
    // (not for a link of a longer chain: the columns of the model it starts from are formed by the coarse chain's
    // first fit, which must then be over before chunk 0 looks them up)
    pipeline = ep && std::atoi(ep) != 0 && (int)k->ctx.size() > C && (int)k->pool.th.size() >= C && link == nullptr;
    if (pipeline) staged = true;
  }
  if (staged && !s->kch_slot_w && dmalloc(&s->kch_slot_w, (size_t)s->p) != hipSuccess) {
    (void)hipGetLastError();
    s->kch_slot_w = nullptr;
    staged = pipeline = false;
  }
  if (staged && !s->kch_fill_st && !s->kch_fill_tried) {
    s->kch_fill_tried = true;  // (decided once per session: no unit to leave out, or no such stream to be had)
    // the fills' stream: every compute unit but a few (test hook kchunks_reserve=count[:stride of the mask bits]; 0: the
    // fills run on the filling chain's own stream)
    int leave = 16, stride = 1;
    if (s->panel_variant == 5) {
      // k_cov_panel_dp runs ONE workgroup per compute unit: the units left out must be ones its rounds of workgroups
      // leave idle anyway (configs[1]: 237 workgroups, one round), or a pass on the fill stream would take a round more
      hipDeviceProp_t prop;
      const int cus = hipGetDeviceProperties(&prop, s->device) == hipSuccess ? prop.multiProcessorCount : 0;
      // (the most units that can go without a round more: rounds = ceil(blocks / units) must stay what it is)
      const int blocks = s->cov_panel_blocks;
      const int rounds = (cus > 0 && blocks > 0) ? (blocks + cus - 1) / cus : 0;
      const int idle = rounds > 0 ? cus - (blocks + rounds - 1) / rounds : 0;
      leave = idle >= 4 ? std::min(16, idle) : 0;
    }
    if (const char *er = test_hook("kchunks_reserve")) {
      leave = std::max(0, std::atoi(er));
      if (const char *c2 = std::strchr(er, ':')) stride = std::max(1, std::atoi(c2 + 1));
    }
    if (leave > 0 && !ctx_stream_create(s->device, &s->kch_fill_st, leave, stride)) s->kch_fill_st = nullptr;
  }
  if (lm_cov && (pipeline || test_hook("kchunks_weights"))) {
    // chunk lengths by weight (test hook kchunks_weights=w0:w1:...).  Pipeline: chunk r starts when coarse fit r is
    // done -- the later a chunk starts the shorter it is (default: falling linearly to a third)
    std::vector<double> w((size_t)C, 1.0);
    for (int r = 0; r < C && pipeline; r++) w[(size_t)r] = 1.0 - (2.0 / 3.0) * r / std::max(1, C - 1);
    if (const char *ew = test_hook("kchunks_weights")) {
      int r = 0;
      for (const char *q = ew; *q && r < C; r++) {
        w[(size_t)r] = std::max(0.05, std::atof(q));
        while (*q && *q != ':') q++;
        if (*q == ':') q++;
      }
    }
    double tot = 0.0, acc = 0.0;
    for (double v : w) tot += v;
    for (int r = 1; r < C; r++) {
      acc += w[(size_t)r - 1];
      bounds[(size_t)r] = std::min(std::max((int)std::lround(ns * acc / tot), bounds[(size_t)r - 1] + 1), ns - (C - r));
    }
    bounds[(size_t)C] = ns;
  }
  const auto t_begin = now();
  {
    const char *el = test_hook("kchunks_log");
    k->log_on = el && std::atoi(el) != 0;
    k->log_t0 = t_begin;
    k->log.clear();
  }
  std::vector<ChunkRun> run((size_t)C);
  SparseVec link_init;
  double link_c0 = 0.0;
  if (link) {  // the model the whole link starts from: chunk 0's start, and the coarse chain's
    for (int i = 0; i < link->init_len; i++) {
      if (link->init_idx[i] < 0 || link->init_idx[i] >= s->p) return fail(BESSX_ERR_ARG, "chain: init index out of range");
      link_init.idx.push_back(link->init_idx[i]);
      link_init.val.push_back(link->init_val[i]);
    }
    link_c0 = link->init_coef0;
    run[0].init_idx = link_init.idx;
    run[0].init_val = link_init.val;
    run[0].init_coef0 = link_c0;
    link->stopped_at = -1;
  }
  s->hint.on = false;
  // The coarse chain.  (LM only: it is what fills the shared cache.  The other families keep no cache: their chunks start
  // cold, side by side -- a restricted fit is cold-started anyway, only the active set is warm -- and are stitched)
  // In front of the chunks: at most COARSE_MAX coarse fits, at chunk boundaries spread evenly over the path: with more
  // chunks than that a chunk starts from the nearest coarse model below it (or cold, in front of the first one) -- its
  // own first fits bridge the gap side by side with the other chunks, where a coarse fit per chunk (7 for 8 chunks:
  // + 1.2 ms on configs[1]) would run before any of them.  Beside the chunks (pipeline): one per boundary.
  // Starting points only: the stitch makes the path the single chain's.
  constexpr int COARSE_MAX = 3;
  int M = lm_cov ? (pipeline ? C - 1 : std::min(C - 1, COARSE_MAX)) : 0;  // (no cache to fill otherwise)
  if (const char *ev = test_hook("kchunks_coarse")) M = std::max(0, std::min(M, std::atoi(ev)));  // (0: every chunk starts cold)
  std::vector<int> coarse_at;  // coarse fit m (1-based) ends at the lower boundary of chunk coarse_at[m - 1]
  {
    int done_r = 0;
    for (int j = 1; j <= M; j++) {
      const int r = std::max(done_r + 1, (int)((long)C * j / (M + 1)));
      if (r >= C) break;
      coarse_at.push_back(r);
      done_r = r;
    }
  }
  std::vector<int> start_of((size_t)C, 0);  // chunk r starts from coarse fit start_of[r] (0: cold, or the link's model)
  for (size_t m = 0; m < coarse_at.size(); m++)
    for (int q = coarse_at[m]; q < C; q++) start_of[(size_t)q] = (int)m + 1;
  struct CoarseModel {
    std::vector<int> idx;
    std::vector<double> val;
    double c0 = 0.0;
  };
  std::vector<CoarseModel> coarse_model(coarse_at.size() + 1);
  long long coarse_fits = 0;
  auto t_coarse = t_begin;
  if (!pipeline) {
    // ---- 1. the coarse chain on the session's own state (the caller has just started the caches over)
    SparseVec init = link_init;
    double c0 = link_c0;
    for (size_t m = 0; m < coarse_at.size(); m++) {
      if (int rc = run_fit(s, seq[bounds[(size_t)coarse_at[m]] - 1], lambda, init, c0)) return rc;
      init = s->beta;
      c0 = s->coef0;
      coarse_model[m + 1] = {init.idx, init.val, c0};
    }
    if (int rc = settle_device_chain(s)) return rc;
    HIPX(hipStreamSynchronize(s->st));  // the cache is complete before any chunk chain reads it
    coarse_fits = s->n_fits;
    t_coarse = now();
    for (int r = 0; r < C; r++)
      if (start_of[(size_t)r] > 0) {
        const CoarseModel &cm = coarse_model[(size_t)start_of[(size_t)r]];
        run[(size_t)r].init_idx = cm.idx;
        run[(size_t)r].init_val = cm.val;
        run[(size_t)r].init_coef0 = cm.c0;
      }
  }
  // The writer's slot map is a copy of the readers' whenever a round of chains starts (no fill is in flight between two
  // rounds; what ran in between -- the coarse chain, a chunk phase as merged launches -- filled through the readers' map)
  auto writer_map_in_step = [&]() -> int {
    if (!staged) return 0;
    HIPX(hipMemcpyAsync(s->kch_slot_w, s->cov[0].slot_of, (size_t)s->p * sizeof(int), hipMemcpyDeviceToDevice, s->st));
    HIPX(hipStreamSynchronize(s->st));
    return 0;
  };
  if (int rc = writer_map_in_step()) return rc;
  if (lm_cov)
    for (bessx_session *c : k->ctx) c->cov[0].slot_w = staged ? s->kch_slot_w : nullptr;
  // ---- 2. the chunks side by side: as merged launches on the session's stream where that applies (LM, the fused
  // selection + solve kernels' range), else on a stream and a host thread each
  for (int r = 0; r < C; r++) run[r].shape(bounds[r], bounds[r + 1], width, s->p_full);
  const int merged = pipeline ? -1 : mc_run_chunks(s, k, seq, ns, C, bounds, run, lambda, ic_type, width);
  if (merged > 0) return merged;
  kchains_round(k, C + (pipeline ? 1 : 0), staged);
  {
    // the chains of the streaming forms share their passes over X (SharedPass above; test hook kchunks_shared_pass=0:
    // a pass per chain, as in round 5)
    SharedPass &sp = k->sp;
    const char *eh = test_hook("kchunks_shared_pass");
    sp.kind = s->model_type == 1 ? 0 : (s->model_type == 4 ? 2 : 1);
    sp.cap = sp.kind == 2 ? COX_MC_MAX : XTV_MC_MAX;
    // (one group: two alternating groups -- one group's selection / solve / IRLS steps under the other group's pass --
    // were built and measured SLOWER, 131 against 91 ms on the streaming configs[1] path with 4 chains, 95 against 84 ms on
    // logistic with 8: every pass then serves half the chains, and the small kernels run 3-8 x slower beside a pass)
    sp.groups = 1;
    if (const char *eg = test_hook("kchunks_pass_groups")) sp.groups = std::max(1, std::min(SharedPass::GROUPS, std::atoi(eg)));
    sp.on = !lm_cov && C >= 2 && (C + sp.groups - 1) / sp.groups <= sp.cap && (!eh || std::atoi(eh) != 0) &&
            (sp.kind != 2 || s->cox.one_pass) && sp_prepare(s, sp);
    sp.timing = s->timing;
    sp.rc = 0;
    for (auto &r : sp.rdv) r.timeout_s = std::min(0.05, s->wait_deadline_s);
    sp_round(k, std::vector<bessx_session *>(k->ctx.begin(), k->ctx.begin() + C));
  }
  struct Starts {  // (pipeline) which coarse fits are done
    std::mutex mu;
    std::condition_variable cv;
    int done = 0;
    bool failed = false;
  } starts;
  int coarse_rc = 0;
  std::string coarse_err;
  // The stitch's first round, early (test hook kchunks_early_stitch=0 switches it off): the thread of chunk r - 1, when its
  // chunk is walked, re-fits the first candidates of chunk r from its own last model on its own (now idle) context while
  // the later chunks are still running -- what round 1 of the stitch below would do after ALL chunks have ended.  It
  // compares with the rows chunk r has stored by then (kchains_progress); a refit that could not see `budget` rows is
  // thrown away and done again in round 1.
  bool early_on = !pipeline && C <= 9;
  if (const char *ee = test_hook("kchunks_early_stitch")) early_on = early_on && std::atoi(ee) != 0;
  std::vector<ChunkRun> early_st((size_t)C);
  std::vector<char> early_ok((size_t)C, 0);
  for (int r = 0; r < 10; r++) {
    k->progress[r].store(0, std::memory_order_relaxed);
    k->ended[r].store(0, std::memory_order_relaxed);
  }
  auto budget_of = [&](int r) { return std::max(8, (run[r].hi - run[r].lo) / 6); };
  auto early_stitch = [&](int r) {  // on the thread and the context of chunk r - 1, which has just ended without error
    bessx_session *c = k->ctx[(size_t)r - 1];
    ChunkRun &q = run[(size_t)r], &t = early_st[(size_t)r];
    const int want = std::min(q.hi - q.lo, budget_of(r));
    const auto t0 = std::chrono::steady_clock::now();
    sp_leave(c);  // (this wait is not for the device: the other chains' shared passes must not wait for this thread)
    while (k->progress[r].load(std::memory_order_acquire) < want && k->ended[r].load(std::memory_order_acquire) == 0) {
      kchains_safe_point(c);  // (a chain that waits for every other chain to stand still must not wait for this one)
      std::this_thread::sleep_for(std::chrono::microseconds(20));
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > s->wait_deadline_s) return;
    }
    if (k->ended[r].load(std::memory_order_acquire) == 2) return;
    const int rows = k->progress[r].load(std::memory_order_acquire);
    if (rows < want) return;
    const ChunkRun &pr = run[(size_t)r - 1];
    std::vector<int> pidx(pr.last_idx.begin(), pr.last_idx.begin() + pr.last_len);
    std::vector<double> pval(pr.last_val.begin(), pr.last_val.begin() + pr.last_len);
    t.shape(q.lo, q.hi, width, s->p_full);
    t.bind(&t.res);
    t.chain = bessx_path_chain();
    t.chain.init_idx = pidx.data();
    t.chain.init_val = pval.data();
    t.chain.init_len = (int)pidx.size();
    t.chain.init_coef0 = pr.last_coef0;
    t.chain.keep_caches = 1;
    t.chain.stop_support = q.support.data();
    t.chain.stop_beta = q.beta.data();
    t.chain.stop_rows = rows;  // (the rows chunk r has stored so far: final, it only appends)
    t.chain.stop_row_len = width;
    t.chain.stop_rtol = 1e-9;
    t.chain.last_idx = t.last_idx.data();
    t.chain.last_val = t.last_val.data();
    t.chain.last_cap = width;
    kchains_log(c, "early stitch of the next chunk", q.lo, rows);
    sp_join(c);
    t.rc = context_begin(c, true);  // (its last fit's model IS the model the refit starts from)
    if (t.rc == 0) t.rc = sequential_path(c, seq + q.lo, want, &lambda, 1, ic_type, 0, &t.res, &t.chain);
    if (t.rc == 0 && hipStreamSynchronize(c->st) != hipSuccess) t.rc = fail(BESSX_ERR_HIP, "chunk chain: stream");
    t.fits = c->n_fits;
    t.chain.init_idx = nullptr;  // (locals of this function)
    t.chain.init_val = nullptr;
    // a failure here is not the path's: round 1 of the stitch does this refit again
    if (t.rc == 0) early_ok[(size_t)r] = 1;
    else (void)hipGetLastError();
    kchains_log(c, "early stitch done", t.res.n_candidates, t.chain.stopped_at);
  };
  auto coarse_job = [&] {
    bessx_session *w = k->ctx[(size_t)C];
    int rc = hipSetDevice(s->device) == hipSuccess ? context_begin(w) : fail(BESSX_ERR_HIP, "hipSetDevice");
    w->timing = s->timing;
    w->hint.on = false;
    SparseVec init = link_init;
    double c0 = link_c0;
    for (size_t m = 0; m < coarse_at.size() && rc == 0; m++) {
      rc = run_fit(w, seq[bounds[(size_t)coarse_at[m]] - 1], lambda, init, c0);
      if (rc) break;
      init = w->beta;
      c0 = w->coef0;
      {
        std::lock_guard<std::mutex> lk(starts.mu);
        coarse_model[m + 1] = {init.idx, init.val, c0};
        starts.done = (int)m + 1;
      }
      kchains_log(w, "coarse fit done", seq[bounds[(size_t)coarse_at[m]] - 1], w->l);
      starts.cv.notify_all();
    }
    if (rc == 0) rc = settle_device_chain(w);
    if (rc == 0 && hipStreamSynchronize(w->st) != hipSuccess) rc = fail(BESSX_ERR_HIP, "coarse chain: stream");
    if (rc) {
      coarse_err = g_err;
      {
        std::lock_guard<std::mutex> lk(starts.mu);
        starts.failed = true;
      }
      starts.cv.notify_all();
    }
    coarse_rc = rc;
    coarse_fits = w->n_fits;
    t_coarse = now();
    kchains_leave(k, rc != 0);
  };
  auto chunk_job = [&](int r) {
    if (pipeline && r == C) {
      coarse_job();
      return;
    }
    if (r >= C) return;
    bessx_session *c = k->ctx[r];
    ChunkRun &q = run[r];
    q.rc = 0;
    if (pipeline && start_of[(size_t)r] > 0) {  // its starting model: the coarse fit at its lower boundary
      std::unique_lock<std::mutex> lk(starts.mu);
      const bool ok = bessx_timed_wait(starts.cv, lk, s->wait_deadline_s,
                                       [&] { return starts.done >= start_of[(size_t)r] || starts.failed; });
      if (!ok || starts.failed) {
        q.rc = fail(BESSX_ERR_HIP, "chunk chain: the coarse chain did not deliver its starting model");
      } else {
        const CoarseModel &cm = coarse_model[(size_t)start_of[(size_t)r]];
        q.init_idx = cm.idx;
        q.init_val = cm.val;
        q.init_coef0 = cm.c0;
      }
    }
    if (q.rc == 0) q.rc = hipSetDevice(s->device) == hipSuccess ? context_begin(c) : fail(BESSX_ERR_HIP, "hipSetDevice");
    c->timing = s->timing;  // (its fills count in the session's score-pass statistics)
    kchains_log(c, "chunk starts", q.lo, q.hi);
    if (q.rc == 0) {
      q.bind(&q.res);
      q.chain = bessx_path_chain();
      q.chain.init_idx = q.init_idx.data();
      q.chain.init_val = q.init_val.data();
      q.chain.init_len = (int)q.init_idx.size();
      q.chain.init_coef0 = q.init_coef0;
      q.chain.keep_caches = 1;
      q.chain.last_idx = q.last_idx.data();
      q.chain.last_val = q.last_val.data();
      q.chain.last_cap = width;
      q.rc = sequential_path(c, seq + q.lo, q.hi - q.lo, &lambda, 1, ic_type, 0, &q.res, &q.chain);
      q.last_len = q.chain.last_len;
      q.last_coef0 = q.chain.last_coef0;
      q.fits += c->n_fits;
    }
    if (q.rc == 0 && hipStreamSynchronize(c->st) != hipSuccess) q.rc = fail(BESSX_ERR_HIP, "chunk chain: stream");
    if (q.rc) q.err = g_err;
    kchains_log(c, "chunk ends", q.lo, q.hi);
    if (r < 10) k->ended[r].store(q.rc ? 2 : 1, std::memory_order_release);
    if (early_on && q.rc == 0 && r + 1 < C) early_stitch(r + 1);
    sp_leave(c);
    kchains_leave(k, q.rc != 0);
  };
  if (merged < 0) {
    if (!k->pool.run(chunk_job, s->wait_deadline_s)) return fail(BESSX_ERR_HIP, "chunk chains: a host thread did not come back");
    if (pipeline && coarse_rc) return fail(coarse_rc, "coarse chain: " + coarse_err);
    for (int r = 0; r < C; r++)
      if (run[r].rc) return fail(run[r].rc, "chunk chain: " + run[r].err);
  }
  const auto t_chunks = now();
  // ---- 3. the stitch, in rounds until no chunk's last model changed (bess_amd.dist.StitchedKPath.step)
  // A chunk whose refit has not met its own chain after `budget` candidates is on another trajectory than the warm
  // chain (designs with many near-equivalent supports: correlated columns, weak signal) -- re-walking it, and then its
  // successors round by round, would cost more than the one chain.  The stitch then gives up: everything from the first
  // unsettled chunk on is walked as ONE chain from its predecessor's last model, and the session's automatic choice
  // becomes one chain.
  std::vector<char> need((size_t)C, 1), changed((size_t)C, 0);
  need[0] = 0;
  long long refits = 0;
  int give_up_from = -1;
  for (int round = 1;; round++) {
    std::vector<ChunkRun> st((size_t)C);
    std::vector<char> have((size_t)C, 0);  // refits of this round that are there already (the early stitch)
    if (round == 1 && merged < 0)
      for (int r = 1; r < C; r++)
        if (early_ok[(size_t)r]) {
          st[(size_t)r] = std::move(early_st[(size_t)r]);
          have[(size_t)r] = 1;
        }
    std::vector<std::vector<int>> pred_idx((size_t)C);
    std::vector<std::vector<double>> pred_val((size_t)C);
    std::vector<double> pred_c0((size_t)C, 0.0);
    for (int r = 1; r < C; r++)
      if (need[r]) {  // the predecessor's last model as it stands before this round
        pred_idx[r].assign(run[r - 1].last_idx.begin(), run[r - 1].last_idx.begin() + run[r - 1].last_len);
        pred_val[r].assign(run[r - 1].last_val.begin(), run[r - 1].last_val.begin() + run[r - 1].last_len);
        pred_c0[r] = run[r - 1].last_coef0;
      }
    auto stitch_job = [&](int r) {
      if (r >= C || !need[r] || have[(size_t)r]) return;
      bessx_session *c = k->ctx[r];
      ChunkRun &q = run[r], &t = st[r];
      t.shape(q.lo, q.hi, width, s->p_full);
      t.bind(&t.res);
      t.chain = bessx_path_chain();
      t.chain.init_idx = pred_idx[r].data();
      t.chain.init_val = pred_val[r].data();
      t.chain.init_len = (int)pred_idx[r].size();
      t.chain.init_coef0 = pred_c0[r];
      t.chain.keep_caches = 1;
      t.chain.stop_support = q.support.data();
      t.chain.stop_beta = q.beta.data();
      t.chain.stop_rows = q.hi - q.lo;
      t.chain.stop_row_len = width;
      t.chain.stop_rtol = 1e-9;
      t.chain.last_idx = t.last_idx.data();
      t.chain.last_val = t.last_val.data();
      t.chain.last_cap = width;
      t.rc = hipSetDevice(s->device) == hipSuccess ? context_begin(c) : fail(BESSX_ERR_HIP, "hipSetDevice");
      c->timing = s->timing;
      if (t.rc == 0)
        t.rc = sequential_path(c, seq + q.lo, std::min(q.hi - q.lo, budget_of(r)), &lambda, 1, ic_type, 0, &t.res, &t.chain);
      if (t.rc == 0 && hipStreamSynchronize(c->st) != hipSuccess) t.rc = fail(BESSX_ERR_HIP, "chunk chain: stream");
      if (t.rc) t.err = g_err;
      t.fits = c->n_fits;
      sp_leave(c);
      kchains_leave(k, t.rc != 0);
    };
    int active = 0;
    for (int r = 1; r < C; r++) active += (need[r] && !have[(size_t)r]) ? 1 : 0;
    if (active > 0) {
      if (int rc = writer_map_in_step()) return rc;
      kchains_round(k, active, staged);
      {
        std::vector<bessx_session *> who;
        for (int r = 1; r < C; r++)
          if (need[r] && !have[(size_t)r]) who.push_back(k->ctx[r]);
        sp_round(k, who);
      }
      if (!k->pool.run(stitch_job, s->wait_deadline_s)) return fail(BESSX_ERR_HIP, "chunk chains: a host thread did not come back");
    }
    bool any = false;
    for (int r = 1; r < C; r++) {
      changed[r] = 0;
      if (!need[r]) continue;
      ChunkRun &q = run[r], &t = st[r];
      if (t.rc) return fail(t.rc, "chunk chain (stitch): " + t.err);
      const int m = std::min(t.res.n_candidates, q.hi - q.lo);
      for (int i = 0; i < m; i++) {  // what was walked before the merge point is replaced
        q.T0[i] = t.T0[i];
        q.iters[i] = t.iters[i];
        q.lam[i] = t.lam[i];
        q.loss[i] = t.loss[i];
        q.ic[i] = t.ic[i];
        q.coef0[i] = t.coef0[i];
        std::copy(t.support.begin() + (size_t)i * width, t.support.begin() + (size_t)(i + 1) * width,
                  q.support.begin() + (size_t)i * width);
        std::copy(t.beta.begin() + (size_t)i * width, t.beta.begin() + (size_t)(i + 1) * width,
                  q.beta.begin() + (size_t)i * width);
      }
      refits += m;
      q.fits += t.fits;
      if (t.chain.stopped_at < 0 && m < q.hi - q.lo) {
        // out of budget without meeting the chunk's own chain
        if (give_up_from < 0 || r < give_up_from) give_up_from = r;
      } else if (t.chain.stopped_at < 0) {  // the whole (short) chunk was replaced: its last model is a new one
        changed[r] = 1;
        any = true;
        q.last_idx = t.last_idx;
        q.last_val = t.last_val;
        q.last_len = t.chain.last_len;
        q.last_coef0 = t.chain.last_coef0;
      }
    }
    if (give_up_from >= 0) {
      // a chunk behind one whose last model changed in this round was refitted from the old model: unsettled too
      for (int r = 1; r < give_up_from; r++)
        if (changed[r] && r + 1 < give_up_from) give_up_from = r + 1;
      break;
    }
    if (!any) break;
    if (round > C) return fail(BESSX_ERR_NUMERIC, "chunk chains: the stitching did not settle");
    for (int r = 1; r < C; r++) need[r] = changed[r - 1];
  }
  if (give_up_from >= 1) {
    // the rest as ONE chain on the session's own state, from the last settled chunk's last model, on the warm cache
    const int f = give_up_from;
    ChunkRun tail;
    tail.shape(bounds[f], ns, width, s->p_full);
    tail.bind(&tail.res);
    tail.chain = bessx_path_chain();
    tail.chain.init_idx = run[f - 1].last_idx.data();
    tail.chain.init_val = run[f - 1].last_val.data();
    tail.chain.init_len = run[f - 1].last_len;
    tail.chain.init_coef0 = run[f - 1].last_coef0;
    tail.chain.keep_caches = 1;
    tail.chain.last_idx = tail.last_idx.data();
    tail.chain.last_val = tail.last_val.data();
    tail.chain.last_cap = width;
    kchains_quiesce(s);
    if (int rc = settle_device_chain(s)) return rc;
    for (auto &q : s->cache) q.valid = q.model_only = false;
    s->dev_state_rs = -1;
    const long long fits_before = s->n_fits;
    if (int rc = sequential_path(s, seq + bounds[f], ns - bounds[f], &lambda, 1, ic_type, 0, &tail.res, &tail.chain)) return rc;
    tail.last_len = tail.chain.last_len;
    tail.last_coef0 = tail.chain.last_coef0;
    tail.fits = s->n_fits - fits_before;
    run.resize((size_t)f);
    run.push_back(std::move(tail));
    s->kch_giveups++;
    if (s->kpath_chains == 0) s->kch_auto_off = true;
  }
  for (bessx_session *c : k->ctx) c->kch_sp_member = false;
  k->sp.on = false;
  if (int rc = sp_collect(s, k->sp)) return rc;
  if (k->sp.rc) return fail(BESSX_ERR_HIP, std::string("shared pass over X: ") + hipGetErrorString((hipError_t)k->sp.rc));
  const int R = (int)run.size();  // chunks of the result (fewer than C after a give-up)
  if (k->log_on) {
    kchains_log(s, "path ends", 0, 0);
    for (const auto &r : k->log) std::fprintf(stderr, "kchunks %8.3f ms  chain %2d  %-28s %d %d\n", r.ms, r.chain, r.what, r.a, r.b);
  }
  s->kch_t[0] += secs(t_begin, t_coarse);
  s->kch_t[1] += secs(t_coarse, t_chunks);
  s->kch_t[2] += secs(t_chunks, now());
  // ---- the path's result: the candidates in order, the best of them by the criterion (first minimum, :113)
  res->n_candidates = 0;
  int best_r = 0, best_i = 0;
  long long iters_path = 0, fits_all = coarse_fits;
  for (int r = 0; r < R; r++) {
    const ChunkRun &q = run[r];
    fits_all += q.fits;
    for (int i = 0; i < q.hi - q.lo; i++) {
      const int g = res->n_candidates++;
      iters_path += std::min(q.iters[i], s->max_iter);  // (a fit that ran out of iterations reports max_iter + 1)
      if (q.ic[i] < run[best_r].ic[best_i]) {
        best_r = r;
        best_i = i;
      }
      if (g >= res->capacity) continue;
      if (res->cand_T0) res->cand_T0[g] = q.T0[i];
      if (res->cand_lambda) res->cand_lambda[g] = q.lam[i];
      if (res->cand_iters) res->cand_iters[g] = q.iters[i];
      if (res->cand_train_loss) res->cand_train_loss[g] = q.loss[i];
      if (res->cand_ic) res->cand_ic[g] = q.ic[i];
      if (res->cand_coef0) res->cand_coef0[g] = q.coef0[i];
      for (int j = 0; j < res->max_T0; j++) {
        if (res->cand_support) res->cand_support[(size_t)g * res->max_T0 + j] = j < width ? q.support[(size_t)i * width + j] : -1;
        if (res->cand_beta) res->cand_beta[(size_t)g * res->max_T0 + j] = j < width ? q.beta[(size_t)i * width + j] : 0.0;
      }
    }
  }
  {
    const ChunkRun &q = run[best_r];
    if (res->beta) {
      std::fill(res->beta, res->beta + s->p_full, 0.0);
      for (int j = 0; j < width; j++) {
        const int col = q.support[(size_t)best_i * width + j];
        if (col >= 0) res->beta[col] = q.beta[(size_t)best_i * width + j];
      }
    }
    res->coef0 = q.coef0[best_i];
    res->train_loss = q.loss[best_i];
    res->ic = q.ic[best_i];
    res->lambda = q.lam[best_i];
    res->best_T0 = q.T0[best_i];
    res->best_iters = q.iters[best_i];
  }
  for (bessx_session *c : k->ctx) {  // statistics of the contexts belong to the session
    s->cov_panel_groups += c->cov_panel_groups;
    s->cov_cg_fallbacks += c->cov_cg_fallbacks;
    s->cov_tie_rescues += c->cov_tie_rescues;
    s->dbg_waits += c->dbg_waits;
    s->dbg_waits_ready += c->dbg_waits_ready;
    s->dbg_enq_s += c->dbg_enq_s;
    c->dbg_waits = c->dbg_waits_ready = 0;
    c->dbg_enq_s = 0.0;
    s->chain_queued += c->chain_queued;
    s->chain_hits += c->chain_hits;
    c->cov_panel_groups = c->cov_cg_fallbacks = c->cov_tie_rescues = 0;
    c->chain_queued = c->chain_hits = 0;
    s->n_submodel_steps += c->n_submodel_steps;
    c->n_submodel_steps = 0;
    s->k1_seconds += c->k1_seconds;
    s->k1_bytes += c->k1_bytes;
    s->k1_launches += c->k1_launches;
    c->k1_seconds = c->k1_bytes = 0.0;
    c->k1_launches = 0;
    for (int w = 0; w < 2; w++) {
      s->panel_w_seconds[w] += c->panel_w_seconds[w];
      s->panel_w_launches[w] += c->panel_w_launches[w];
      c->panel_w_seconds[w] = 0.0;
      c->panel_w_launches[w] = 0;
    }
  }
  // fits and get_A calls as the single chain counts them; the work really done (coarse chain, replaced candidates) is in
  // bessx_session_counter 14-16
  (void)fits_all;
  s->n_fits = ns;
  s->n_iters = iters_path;
  s->kch_paths++;
  s->kch_last_chains = C;
  s->kch_refits += refits;
  // the session's own device state is the coarse chain's last fit, not the path's last candidate
  for (auto &q : s->cache) q.valid = q.model_only = false;
  s->dev_state_rs = -1;
  if (link) {  // the model the link's successor starts from
    const ChunkRun &q = run[R - 1];
    link->last_len = q.last_len;
    link->last_coef0 = q.last_coef0;
    for (int i = 0; i < q.last_len && i < link->last_cap; i++) {
      if (link->last_idx) link->last_idx[i] = q.last_idx[i];
      if (link->last_val) link->last_val[i] = q.last_val[i];
    }
  }
  // Algorithm state as the path leaves it: the last candidate's model (normalised scale)
  {
    const ChunkRun &q = run[R - 1];
    s->beta.idx.assign(q.last_idx.begin(), q.last_idx.begin() + q.last_len);
    s->beta.val.assign(q.last_val.begin(), q.last_val.begin() + q.last_len);
    s->coef0 = q.last_coef0;
  }
  return 0;
}

}  // namespace bessx
