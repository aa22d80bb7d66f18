"""Multi-GPU plumbing for the paths that shard: one process per GPU, units split
across ranks with no data-path collective, and ONE small collective at the end -- the all-gather of the
per-candidate IC / CV curve (8 bytes per candidate).  torch.distributed is plumbing only: backend "nccl"
(= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
"""
import numpy as np


def partition(n_units, world, rank):
    """Contiguous, balanced split of unit indices 0..n_units-1: rank r gets units[lo:hi].  Contiguity matters
    for the k-path: each chunk is one warm-start chain (src/path.cpp:60-64)."""
    base, extra = divmod(n_units, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def rebalance_bounds(bounds, seconds, damping=0.5, fixed_seconds=0.0, deadband=0.12):
    """Chunk boundaries of a k-path (bounds[r] .. bounds[r+1] = rank r's candidates) moved towards equal TIME per rank:
    the candidates of a chunk are priced at its measured seconds / its length (candidates at large sparsity levels
    cost more: larger solves, more Gram columns still to form), the ideal boundaries cut the cumulated price into
    equal parts, and every boundary moves `damping` of the way there (the fills are lumpy: no jumping after one
    measurement).  Every chunk keeps at least one candidate; the same input gives the same output on every rank.
    A time above three times the median is taken for an outlier (a kernel instance loaded at its first launch costs
    tens of ms once) and clipped; nothing moves while the ranks' step times -- `seconds` plus the part every rank spends
    alike, `fixed_seconds` -- are within `deadband` of each other."""
    b = [int(v) for v in bounds]
    world = len(b) - 1
    n = b[-1]
    if world < 2 or n < world or len(seconds) != world:
        return b
    seconds = np.minimum(np.asarray(seconds, dtype=np.float64), 3.0 * float(np.median(seconds)))
    tot = seconds + float(fixed_seconds)
    if not float(np.mean(tot)) > 0.0 or (float(np.max(tot)) - float(np.min(tot))) <= deadband * float(np.mean(tot)):
        return b
    price = np.zeros(n)
    for r in range(world):
        if b[r + 1] > b[r]:
            price[b[r]:b[r + 1]] = max(float(seconds[r]), 0.0) / (b[r + 1] - b[r])
    cum = np.concatenate([[0.0], np.cumsum(price)])
    if not cum[-1] > 0.0:
        return b
    new = [0]
    for r in range(1, world):
        ideal = int(np.searchsorted(cum, cum[-1] * r / world, side="left"))
        v = int(round(b[r] + damping * (ideal - b[r])))
        new.append(min(max(v, new[-1] + 1), n - (world - r)))
    new.append(n)
    return new


def gather_curve(local_values, n_units, world, rank, device=None):
    """All-gather the per-unit scalars of every rank into one curve of length n_units (rank order = unit
    order under partition()).  Works for unequal chunk lengths by padding to the longest chunk."""
    import torch
    import torch.distributed as dist
    local_values = np.asarray(local_values, dtype=np.float64)
    longest = -(-n_units // world)
    buf = torch.full((longest,), float("nan"), dtype=torch.float64, device=device)
    buf[:local_values.size] = torch.as_tensor(local_values, dtype=torch.float64, device=device)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    curve = np.empty(n_units)
    for r in range(world):
        lo, hi = partition(n_units, world, r)
        curve[lo:hi] = out[r][:hi - lo].cpu().numpy()
    return curve


def gather_rows(local_values, world, device=None):
    """All-gather one equally long vector per rank (weak scaling: every rank solved its own full path);
    returns a (world x len) array."""
    import torch
    import torch.distributed as dist
    mine = torch.as_tensor(np.asarray(local_values, dtype=np.float64), device=device)
    out = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    return np.stack([o.cpu().numpy() for o in out])


def select_best(curve):
    """argmin with the reference's tie rule: the first minimum wins (Eigen minCoeff, src/path.cpp:113)."""
    return int(np.argmin(curve))


class _NoComm:
    """world = 1: the all-gather is the identity."""

    def all_gather(self, mine, world):
        return [mine]


class _TorchComm:
    def __init__(self, device=None):
        self.device = device

    def all_gather(self, mine, world):
        import torch
        import torch.distributed as dist
        t = torch.as_tensor(mine, dtype=torch.float64, device=self.device)
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return [o.cpu().numpy() for o in out]

    def exchange_blocks(self, session, world, rank, ng):
        """The data-path collective of the cooperative prefill: every rank's p x 32 Gram column blocks to every rank.
        Backend nccl (= RCCL over xGMI): the blocks never leave device memory -- the library copies them out of and into
        its cache through the tensors' device pointers.  gloo (CPU tests, one-device rehearsal): through host arrays."""
        import torch
        import torch.distributed as dist
        p = session.p_kept
        longest = -(-ng // world)
        lo, hi = partition(ng, world, rank)
        on_dev = self.device is not None and str(self.device).startswith("cuda")
        if on_dev:
            buf = torch.zeros(longest * 32 * p, dtype=torch.float64, device=self.device)
            if hi > lo:
                session.cov_prefill_export(lo, hi - lo, device_ptr=buf.data_ptr())
            out = [torch.empty_like(buf) for _ in range(world)]
            dist.all_gather(out, buf)
            torch.cuda.synchronize()
            for r in range(world):
                a, b = partition(ng, world, r)
                if r != rank and b > a:
                    session.cov_prefill_import(a, b - a, device_ptr=out[r].data_ptr())
            return
        mine = np.zeros(longest * 32 * p)
        if hi > lo:
            mine[:(hi - lo) * 32 * p] = session.cov_prefill_export(lo, hi - lo)
        for r, blk in enumerate(self.all_gather(mine, world)):
            a, b = partition(ng, world, r)
            if r != rank and b > a:
                session.cov_prefill_import(a, b - a, blk[:(b - a) * 32 * p])


class BessxComm:
    """The library's own communicator (bessx_comm_*, include/bessx.h section 5: RCCL directly, no torch.distributed in
    the data path) with the all_gather interface the sharded paths use -- what a C or R host has.  `unique_id`: the 128
    bytes of rank 0's BessxComm.unique_id(), handed to the other ranks by the host (here: any broadcast the caller has;
    bench.py --comm bessx uses torch's store once, at start-up).  Collective: every rank constructs it."""

    def __init__(self, rank, world, unique_id, device=0):
        import ctypes
        from . import capi
        self._capi, self._ct = capi, ctypes
        self.rank, self.world = int(rank), int(world)
        h = ctypes.c_void_p()
        capi._check(capi.lib().bessx_comm_init(ctypes.byref(h), self.rank, self.world, bytes(unique_id), int(device)))
        self._h = h

    @staticmethod
    def unique_id():
        import ctypes
        from . import capi
        buf = ctypes.create_string_buffer(128)
        capi._check(capi.lib().bessx_comm_unique_id(buf))
        return buf.raw

    def all_gather(self, mine, world):
        assert world == self.world
        mine = np.ascontiguousarray(mine, dtype=np.float64).ravel()
        out = np.empty(self.world * mine.size)
        D = self._ct.POINTER(self._ct.c_double)
        self._capi._check(self._capi.lib().bessx_comm_allgather_f64(self._h, mine.ctypes.data_as(D), int(mine.size),
                                                                    out.ctypes.data_as(D)))
        return [out[r * mine.size:(r + 1) * mine.size].copy() for r in range(self.world)]

    def exchange_blocks(self, session, world, rank, ng):
        """The cooperative prefill's Gram column blocks through host arrays (the opt-in variant; the blocks could stay on
        the device, as _TorchComm keeps them -- not needed by the default partition, which has no data-path collective)."""
        p = session.p_kept
        longest = -(-ng // world)
        lo, hi = partition(ng, world, rank)
        mine = np.zeros(longest * 32 * p)
        if hi > lo:
            mine[:(hi - lo) * 32 * p] = session.cov_prefill_export(lo, hi - lo)
        for r, blk in enumerate(self.all_gather(mine, world)):
            a, b = partition(ng, world, r)
            if r != rank and b > a:
                session.cov_prefill_import(a, b - a, blk[:(b - a) * 32 * p])

    def close(self):
        if self._h:
            self._capi.lib().bessx_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _share_and_exchange(session, world, rank, comm, ng):
    """This rank's contiguous share of the ng listed 32-column groups (one pass over X per group, two groups per pass with
    the pair kernel), then everybody's blocks to everybody."""
    lo, hi = partition(ng, world, rank)
    session.cov_prefill_compute(lo, hi - lo)
    if world > 1:
        if hasattr(comm, "exchange_blocks"):
            comm.exchange_blocks(session, world, rank, ng)
        else:
            p = session.p_kept
            longest = -(-ng // world)
            mine = np.zeros(longest * 32 * p)
            if hi > lo:
                mine[:(hi - lo) * 32 * p] = session.cov_prefill_export(lo, hi - lo)
            for r, blk in enumerate(comm.all_gather(mine, world)):
                a, b = partition(ng, world, r)
                if r != rank and b > a:
                    session.cov_prefill_import(a, b - a, blk[:(b - a) * 32 * p])
    session.cov_prefill_end()
    return hi - lo


def cooperative_prefill(session, world, rank, comm, n_cols):
    """Before the chunks of a k-path start cold: the ranks share the passes over X their cold starts would all repeat.
    The list is the same on every rank -- the n_cols columns with the largest sacrifice scores at beta = 0, what the
    first PDAS iteration of any cold fit ranks (src/Algorithm.h:1109-1128) --; rank r forms the Gram columns of its
    contiguous share of the 32-column groups, the p x 32 blocks are all-gathered (2.56 MB each at p = 10000: the ONE
    data-path collective of this mode, which north_star's replicas-only partitioning does not have -- keep n_cols = 0
    for that), and every rank's cache then holds all of them under the same slots.  Columns a chunk needs beyond the
    list are formed by the rank itself, as without the prefill.  Cache contents only: every result is what it is
    without it."""
    ng = int(n_cols) // 32
    if ng < 1:
        return 0
    scores = session.marginal_scores()
    cols = np.argsort(-scores, kind="stable")[:ng * 32].astype(np.int32)
    session.cov_prefill_begin(cols)
    return _share_and_exchange(session, world, rank, comm, ng)


def pilot_prefill(session, world, rank, comm, n_first, k_pilot, n_second, ic_type=3, wide=0):
    """Two shared fills around a PILOT fit every rank runs identically.  The marginal list of cooperative_prefill covers
    the first PDAS iteration of a cold fit only: the iterations after it rank the noise columns by the residual of the
    FITTED model, which the marginal ranking does not predict (tools/coop_prefill.py: the slowest of 8 chunks still spent
    8.5 of its 9.5 ms on ~9 private fills).  So: (1) the n_first columns with the largest marginal scores, shared;
    (2) the same fit of sparsity level k_pilot on every rank (same data, same cache, deterministic kernels: bit-identical
    results and cache states, no communication); (3) the n_second uncached columns the pilot's final scores rank highest
    -- the columns the chain is about to visit -- appended to the cache in the same slots everywhere, again one share
    per rank.  Returns the pilot's model (normalised scale): the chunks beyond k_pilot start warm from it instead of cold.
    Two data-path all-gathers of p x 32 blocks; cache contents and starting points only -- the stitching makes the
    gathered path the single chain's whatever the chunks started from."""
    cooperative_prefill(session, world, rank, comm, n_first)
    if wide >= 32 and world > 1:
        # (2'): the pilot fit's OWN fills shared as well.  Every rank's pilot parks at the same iteration on the same
        # missing columns; instead of each forming them privately (a pass over X per 32-64 columns, repeated on every
        # rank) the library lists `wide` columns -- the missing ones, then the best uncached ones by the scores of that
        # very iteration -- and the ranks form one group each (bessx_session_set_fill_hook)
        session.set_fill_hook(lambda ng: _share_and_exchange(session, world, rank, comm, ng), int(wide))
    try:
        pilot = session.sequential_path_chain([int(k_pilot)], ic_type=ic_type, keep_caches=n_first >= 32)
    except Exception:
        if getattr(session, "_hook_error", None) is not None:
            raise session._hook_error
        raise
    finally:
        if wide >= 32 and world > 1:
            session.set_fill_hook(None)
    bd, slot = session.cov_state()
    score = np.where(slot >= 0, -np.inf, bd)
    ng = min(int(n_second) // 32, int(np.sum(slot < 0)) // 32)
    if ng >= 1:
        cols = np.argsort(-score, kind="stable")[:ng * 32].astype(np.int32)
        session.cov_prefill_extend(cols)
        _share_and_exchange(session, world, rank, comm, ng)
    return pilot["last_idx"], pilot["last_val"], pilot["last_coef0"]


# ------------------------------------------------------------------------------------------------------------
# The k-path in contiguous chunks whose gathered candidates EQUAL the single warm-start chain's for every k
# (option (c) carried through; reference chain: src/path.cpp:60-64).
#
# Candidate k of sequential_path starts from candidate k-1's model.  Rank r walks its chunk [lo_r, hi_r) as a chain
# of its own, whose first candidate has no predecessor (cold or ladder start): that chain may differ from the single
# chain's for a few candidates (PDAS is a local fixed-point iteration).  After the chunks -- they are the parallel
# part -- the chains are STITCHED: rank r takes rank r-1's last model (normalised coefficients, bit for bit), re-fits
# its first candidate warm from it and keeps walking warm until a candidate coincides with its own chunk's (same
# support, coefficients to the solver's tolerance); from there on the two chains are the same chain, and the
# candidates in front of that point are replaced.  Rank r-1's last model is only final once ITS stitch has ended
# without touching its last candidate, so the stitching runs in rounds until no last model changed (one round
# unless a whole chunk had to be replaced; at most world-1).  X stays replicated; per step: the chunk, then per
# round one all-gather of (flag, last model) -- a few KB -- and the all-gather of the IC curve.
# ------------------------------------------------------------------------------------------------------------
class StitchedKPath:
    """`session` needs sequential_path_chain(sequence, ic_type=, init_idx=, init_val=, init_coef0=, keep_caches=,
    stop_support=, stop_beta=, stop_rtol=) -> path result + stopped_at, last_idx, last_val, last_coef0
    (bess_amd.capi.Session).  step() returns this rank's chunk of the single chain plus the gathered IC curve."""

    def __init__(self, session, sequence, world=1, rank=0, ic_type=3, lead=(), device=None, stop_rtol=1e-9, comm=None,
                 prefill=0, pilot=None, rebalance=False, coarse_lead=False):
        self.s, self.world, self.rank = session, int(world), int(rank)
        self.full_seq = np.asarray(sequence, dtype=np.int32)
        self.kmax = int(self.full_seq.size)
        self.ic_type = ic_type
        self.bounds = [partition(self.kmax, self.world, r)[0] for r in range(self.world)] + [self.kmax]
        self.lo, self.hi = self.bounds[self.rank], self.bounds[self.rank + 1]
        self.seq = self.full_seq[self.lo:self.hi]
        self.lead = np.asarray([k for k in lead if self.lo > 0 and 1 <= k < (self.seq[0] if self.seq.size else 0)],
                               dtype=np.int32)  # ladder start: sparsity levels walked in front of the chunk, discarded
        # rebalance: after every step the chunk boundaries move towards equal chunk + stitch time per rank
        # (rebalance_bounds on the all-gathered times of that step: the same new boundaries on every rank).  Where a
        # chunk starts changes nothing in the stitched path.  Not with a ladder start (its rungs are tied to k0).
        # (every rank must decide alike: the flag is the caller's; a ladder on ANY rank rules it out)
        if rebalance and len(lead):
            raise ValueError("StitchedKPath: rebalance moves the chunk boundaries, a ladder start is tied to them")
        self.rebalance = bool(rebalance) and self.world > 1
        self.comm = comm if comm is not None else (_TorchComm(device) if world > 1 else _NoComm())
        self.stop_rtol = stop_rtol
        self.prefill = int(prefill)  # columns of the cooperative prefill in front of the chunks (0: replicas only)
        self.pilot = pilot           # (k_pilot, n_second[, wide]) of pilot_prefill, or None: the marginal list alone
        self.width = int(self.full_seq.max()) if self.kmax else 1  # longest support of the path (singleton groups)
        # coarse_lead (round 6, LM): every rank walks, in front of its chunk, the coarse warm-start chain the ONE-GPU path
        # walks in front of its chunk chains (bessx_kchunks.cpp: the levels at the quarter points of s.list) as far as it
        # lies below the chunk, then the level just below the chunk -- bessx_path_chain.lead_levels.  No communication: the
        # ranks repeat each other's coarse fits (deterministic kernels, replicated X); the chunk then starts warm on a
        # cache that holds nearly every column it will ask for, where a cold start at k0 = 176 paid 14 passes over X.
        self.coarse_lead = bool(coarse_lead)

    KEYS = ("cand_T0", "cand_iters", "cand_train_loss", "cand_ic", "cand_coef0", "cand_support", "cand_beta")

    def lead_levels(self):
        """Sparsity levels of this rank's lead fits (coarse_lead): the one-GPU path's coarse levels below the chunk, then
        the level just below the chunk."""
        if not (self.coarse_lead and self.seq.size and self.lo > 0):
            return np.zeros(0, dtype=np.int32)
        first = int(self.seq[0])
        coarse = [int(self.full_seq[self.kmax * j // 4 - 1]) for j in (1, 2, 3) if self.kmax >= 8]
        lv = sorted({k for k in coarse + [int(self.full_seq[self.lo - 1])] if 1 <= k < first})
        return np.asarray(lv, dtype=np.int32)

    def _model_record(self, changed, idx, val, coef0):
        rec = np.zeros(3 + 2 * self.width)
        rec[0], rec[1], rec[2] = float(changed), len(idx), coef0
        rec[3:3 + len(idx)] = idx
        rec[3 + self.width:3 + self.width + len(idx)] = val
        return rec

    def _model_of(self, rec):
        k = int(rec[1])
        return rec[3:3 + k].astype(np.int32), rec[3 + self.width:3 + self.width + k].copy(), float(rec[2])

    def step(self):
        import time
        t0 = time.time()
        nl = int(self.lead.size)
        mine = None
        last = (np.zeros(0, np.int32), np.zeros(0), 0.0)
        t_pre = 0.0
        init = None
        if (self.prefill >= 32 or (self.pilot and len(self.pilot) > 2 and self.pilot[2] >= 32)) and self.world > 1:
            if self.pilot:
                model = pilot_prefill(self.s, self.world, self.rank, self.comm, self.prefill, self.pilot[0], self.pilot[1],
                                      ic_type=self.ic_type, wide=self.pilot[2] if len(self.pilot) > 2 else 0)
                if self.seq.size and int(self.seq[0]) > int(self.pilot[0]):
                    init = model  # (chunks at or below the pilot's level start cold: cheap there)
            else:
                cooperative_prefill(self.s, self.world, self.rank, self.comm, self.prefill)
            t_pre = time.time() - t0
        if self.seq.size and init is not None:
            nl = 0
            out = self.s.sequential_path_chain(self.seq, ic_type=self.ic_type, keep_caches=True, init_idx=init[0],
                                               init_val=init[1], init_coef0=init[2])
        elif self.seq.size and self.coarse_lead and self.lead_levels().size:
            nl = 0
            out = self.s.sequential_path_chain(self.seq, ic_type=self.ic_type, keep_caches=t_pre > 0.0,
                                               lead_levels=self.lead_levels())
        elif self.seq.size:
            out = self.s.sequential_path_chain(np.concatenate([self.lead, self.seq]), ic_type=self.ic_type,
                                               keep_caches=t_pre > 0.0)
        if self.seq.size:
            mine = {k: np.array(out[k][nl:]) for k in self.KEYS}
            last = (out["last_idx"], out["last_val"], out["last_coef0"])
        t_chunk = time.time() - t0
        refits, rounds = 0, 0
        models = self.comm.all_gather(self._model_record(True, *last), self.world)
        need = self.rank >= 1 and self.seq.size > 0  # round 1: every chunk but the first has a predecessor to meet
        while True:
            rounds += 1
            changed = False
            if need:
                pi, pv, pc = self._model_of(models[self.rank - 1])
                res = self.s.sequential_path_chain(self.seq, ic_type=self.ic_type, init_idx=pi, init_val=pv,
                                                   init_coef0=pc, keep_caches=True, stop_support=mine["cand_support"],
                                                   stop_beta=mine["cand_beta"], stop_rtol=self.stop_rtol)
                m = int(res["n_candidates"])
                for k in self.KEYS:
                    a, b = mine[k], res[k]
                    if a.ndim == 2:
                        a[:m, :] = -1 if a.dtype.kind == "i" else 0.0
                        a[:m, :b.shape[1]] = b[:m]
                    else:
                        a[:m] = b[:m]
                refits += m
                if res["stopped_at"] < 0:  # the whole chunk was replaced: its last model is a new one
                    changed = True
                    last = (res["last_idx"], res["last_val"], res["last_coef0"])
            models = self.comm.all_gather(self._model_record(changed, *last), self.world)
            flags = [bool(m[0]) for m in models]
            if not any(flags):
                break
            need = self.rank >= 1 and self.seq.size > 0 and flags[self.rank - 1]
            if rounds > self.world:
                raise RuntimeError("stitching did not settle in world rounds")
        t_stitch = time.time() - t0 - t_chunk
        # the one result collective: the IC curve (and, for the report, this step's stitch statistics)
        longest = max(self.bounds[r + 1] - self.bounds[r] for r in range(self.world))
        buf = np.full(longest + 4, np.nan)
        if mine is not None:
            buf[:self.seq.size] = mine["cand_ic"]
        buf[longest:] = (refits, t_chunk - t_pre, t_stitch, t_pre)
        curve = np.empty(self.kmax)
        stats = []
        for r, b in enumerate(self.comm.all_gather(buf, self.world)):
            lo, hi = self.bounds[r], self.bounds[r + 1]
            curve[lo:hi] = b[:hi - lo]
            stats.append(b[longest:])
        stats = np.asarray(stats)
        used = list(self.bounds)
        if self.rebalance:
            self.bounds = rebalance_bounds(used, stats[:, 1] + stats[:, 2], fixed_seconds=float(np.median(stats[:, 3])))
            self.lo, self.hi = self.bounds[self.rank], self.bounds[self.rank + 1]
            self.seq = self.full_seq[self.lo:self.hi]
        return {"chunk": mine, "ic_curve": curve, "best_k": int(self.full_seq[select_best(curve)]), "bounds": used,
                "stitch_refits": int(stats[:, 0].sum()), "stitch_refits_per_rank": [int(v) for v in stats[:, 0]],
                "stitch_rounds": rounds, "chunk_seconds_per_rank": [float(v) for v in stats[:, 1]],
                "stitch_seconds_per_rank": [float(v) for v in stats[:, 2]],
                "prefill_seconds_per_rank": [float(v) for v in stats[:, 3]]}


# ------------------------------------------------------------------------------------------------------------
# Cross-validated paths with the fold fits dealt to ranks (BASELINE configs[3]).
#
# Under CV one candidate (s, lambda) = the full-data Algorithm::fit + K fold fits (Metric::test_loss,
# src/Metric.h:150-195).  Every fold is its own warm-start chain (cv_initial_model_param.row(k), :177-188) and the
# full-data fits are another one (src/path.cpp:60-64, :173-177); a fold fit reads nothing of the full-data fit of
# the same candidate (its coef0_init is the one the path handed to that fit, i.e. the PREVIOUS candidate's).  So the
# K + 1 chains are the independent units: unit u lives on rank u % world for the whole path, which keeps every
# chain exactly as the single-process path runs it -- the results do not depend on the number of ranks.  X is
# replicated; the only communication is one all-gather of the K + 1 small fit records per evaluation.
#
# (fold x s) pairs (the final sweep of gs_path, src/path.cpp:301-328): under the default warm start fold k
# at size s + 1 starts from fold k at size s (cv_initial_model_param.row(k)), so the pairs of one fold ARE a chain and
# at most K + 1 ranks have work; dealing them out singly would change the starting points and with them, possibly,
# the supports.  With is_warm_start = False every (candidate, unit) pair is independent: whole batches of candidates
# -- the sweep of gs_path, every size of sequential_path -- are dealt pair by pair to ALL ranks (_round_cold), with
# results identical to the single-process path for any number of ranks.
# ------------------------------------------------------------------------------------------------------------
class FoldShardedCV:
    """sequential_path / gs_path under cross-validation (src/path.cpp:25-389 with Metric::is_cv) over `world`
    ranks.  `session` is this rank's solver on the replicated data with the folds already set
    (bess_amd.capi.Session after set_cv): it needs fit(T0, lam, fold, init_idx, init_val, init_coef0) ->
    {support, beta, coef0, iters, train_loss, test_loss}, normalization(), n, p.  Returns the same dictionary
    keys as Session.gs_path / Session.sequential_path on every rank."""

    HEAD = 5  # record = [T0, iters, coef0, train_loss, test_loss, support[T0 ...], beta[T0 ...]]

    def __init__(self, session, K, world=1, rank=0, is_warm_start=True, data_type=1, is_normal=True, device=None,
                 comm=None):
        self.s, self.K, self.world, self.rank = session, int(K), int(world), int(rank)
        self.warm = bool(is_warm_start)
        self.data_type, self.is_normal = data_type, is_normal
        self.comm = comm if comm is not None else (_TorchComm(device) if world > 1 else _NoComm())
        self.units = [u for u in range(self.K + 1) if u % self.world == self.rank]  # unit K = the full-data chain
        self.per_rank = -(-(self.K + 1) // self.world)
        self.cv_init = {k: (np.zeros(0, np.int32), np.zeros(0)) for k in range(self.K)}  # cv_initial_model_param
        self.n_fits = 0
        self.n_pdas_iters = 0
        self.evaluations = 0
        self._norm = None
        if hasattr(session, "reset_caches"):
            session.reset_caches()  # a path starts cold, like a bessCpp call

    # -- one evaluation round: the units of this rank, then the all-gather of the records ------------------
    def _width(self, T0):
        """Columns a record reserves for a fit of sparsity level T0: T0, or -- sessions with groups of size > 1, where T0
        counts groups -- the columns of the T0 widest groups (unused entries: support -1)."""
        return int(self.s.fit_width(T0)) if hasattr(self.s, "fit_width") else int(T0)

    @staticmethod
    def _unpack(sup, beta):
        sup = np.asarray(sup)
        keep = sup >= 0
        return sup[keep].astype(np.int32), np.asarray(beta)[keep].copy()

    def _round(self, T0, lam, want_full, want_folds, full_init, coef0_init):
        W = self._width(T0)
        reclen = self.HEAD + 2 * W
        mine = np.full((self.per_rank, reclen), np.nan)
        row = 0
        if self.warm and getattr(self.s, "is_warm_start", False) and hasattr(self.s, "cv_eval"):
            # the library evaluates this rank's share of the candidate in ONE call: the full-data fit (if this rank owns
            # that chain) and its folds' fits -- side by side on their own streams with union fills where that applies,
            # from the session's own per-fold warm starts (bessx_session_cv_eval)
            own_full = bool(want_full) and self.K in self.units
            folds = [u for u in self.units if u != self.K] if want_folds else []
            init = full_init if full_init is not None else (np.zeros(0, np.int32), np.zeros(0))
            got = self.s.cv_eval(T0, lam, own_full, init[0], init[1], coef0_init, folds) if (own_full or folds) else []
            if self.world == 1:  # nothing to exchange: the records are the result
                recs = dict(zip(([self.K] if own_full else []) + folds, got))
                self.evaluations += 1
                self.n_fits += len(recs)
                self.n_pdas_iters += sum(r["iters"] for r in recs.values())
                return recs
            for u, r in zip(([self.K] if own_full else []) + folds, got):
                mine[row, :self.HEAD] = (u, r["iters"], r["coef0"], r["train_loss"], r["test_loss"])
                k = len(r["support"])
                mine[row, self.HEAD:self.HEAD + W] = -1
                mine[row, self.HEAD + W:] = 0.0
                mine[row, self.HEAD:self.HEAD + k] = r["support"]
                mine[row, self.HEAD + W:self.HEAD + W + k] = r["beta"]
                row += 1
            return self._gather_round(mine, W)
        for u in self.units:
            if u == self.K:
                if not want_full:
                    continue
                r = self.s.fit(T0, lam, -1, full_init[0], full_init[1], coef0_init)
            else:
                if not want_folds:
                    continue
                init = self.cv_init[u] if self.warm else (np.zeros(0, np.int32), np.zeros(0))
                r = self.s.fit(T0, lam, u, init[0], init[1], coef0_init)
                if self.warm:
                    self.cv_init[u] = (r["support"].copy(), r["beta"].copy())
            mine[row, :self.HEAD] = (u, r["iters"], r["coef0"], r["train_loss"], r["test_loss"])
            k = len(r["support"])
            mine[row, self.HEAD:self.HEAD + W] = -1
            mine[row, self.HEAD + W:] = 0.0
            mine[row, self.HEAD:self.HEAD + k] = r["support"]
            mine[row, self.HEAD + W:self.HEAD + W + k] = r["beta"]
            row += 1
        return self._gather_round(mine, W)

    def _gather_round(self, mine, W):
        recs = {}
        for block in self.comm.all_gather(mine, self.world):
            for rec in block:
                if not np.isnan(rec[0]):
                    sup, beta = self._unpack(rec[self.HEAD:self.HEAD + W], rec[self.HEAD + W:])
                    recs[int(rec[0])] = {"iters": int(rec[1]), "coef0": float(rec[2]), "train_loss": float(rec[3]),
                                         "test_loss": float(rec[4]), "support": sup, "beta": beta}
        self.evaluations += 1
        self.n_fits += len(recs)
        self.n_pdas_iters += sum(r["iters"] for r in recs.values())
        return recs

    # -- is_warm_start = False: every (candidate, unit) pair is independent (Algorithm::fit starts from zero,
    # Metric::test_loss does not read cv_initial_model_param) -- a batch of candidates is dealt pair by pair to ALL
    # ranks, so more than K + 1 ranks have work ((fold x s) pairs of the final sweep of gs_path) ----------
    def _round_cold(self, T0s, lam):
        pairs = [(i, u) for i in range(len(T0s)) for u in range(self.K + 1)]
        tmax = max(self._width(T0) for T0 in T0s)
        reclen = self.HEAD + 1 + 2 * tmax
        per_rank = -(-len(pairs) // self.world)
        mine = np.full((per_rank, reclen), np.nan)
        row = 0
        empty = (np.zeros(0, np.int32), np.zeros(0))
        for q, (i, u) in enumerate(pairs):
            if q % self.world != self.rank:
                continue
            T0 = T0s[i]
            r = self.s.fit(T0, lam, -1 if u == self.K else u, empty[0], empty[1], 0.0)
            mine[row, :self.HEAD + 1] = (u, r["iters"], r["coef0"], r["train_loss"], r["test_loss"], i)
            k = len(r["support"])
            mine[row, self.HEAD + 1:self.HEAD + 1 + tmax] = -1
            mine[row, self.HEAD + 1 + tmax:] = 0.0
            mine[row, self.HEAD + 1:self.HEAD + 1 + k] = r["support"]
            mine[row, self.HEAD + 1 + tmax:self.HEAD + 1 + tmax + k] = r["beta"]
            row += 1
        out = [dict() for _ in T0s]
        for block in self.comm.all_gather(mine, self.world):
            for rec in block:
                if np.isnan(rec[0]):
                    continue
                i = int(rec[self.HEAD])
                sup, beta = self._unpack(rec[self.HEAD + 1:self.HEAD + 1 + tmax], rec[self.HEAD + 1 + tmax:])
                out[i][int(rec[0])] = {"iters": int(rec[1]), "coef0": float(rec[2]), "train_loss": float(rec[3]),
                                       "test_loss": float(rec[4]), "support": sup, "beta": beta}
        self.evaluations += 1
        self.n_fits += len(pairs)
        self.n_pdas_iters += sum(r["iters"] for recs in out for r in recs.values())
        return out

    @staticmethod
    def _cv_loss(recs, K):
        acc = 0.0
        for k in range(K):  # fold order, like the single-process path
            acc += recs[k]["test_loss"]
        return acc / K

    def _denorm(self, sup, beta, coef0, gs_variant):
        """src/path.cpp:76-110 (sequential) / :330-342 (golden section)."""
        beta = np.array(beta, dtype=np.float64)
        if not self.is_normal:
            return beta, coef0
        if self._norm is None:
            self._norm = self.s.normalization()  # (p-vectors: fetched once per driver, not per candidate)
        xm, xn, ym = self._norm
        beta = np.sqrt(float(self.s.n)) * beta / xn[sup]
        dot = float(np.dot(beta, xm[sup]))
        if self.data_type == 1:
            coef0 = ym - dot
        elif self.data_type == 2 or gs_variant:
            coef0 = coef0 - dot
        return beta, coef0

    def _result(self, cands, best, gs_variant):
        out = {k: [] for k in ("cand_T0", "cand_lambda", "cand_iters", "cand_train_loss", "cand_ic", "cand_coef0",
                               "cand_support", "cand_beta")}
        for c in cands:
            b, c0 = self._denorm(c["support"], c["beta"], c["coef0"], gs_variant)
            out["cand_T0"].append(c["T0"])
            out["cand_lambda"].append(c["lambda"])
            out["cand_iters"].append(c["iters"])
            out["cand_train_loss"].append(c["loss"])
            out["cand_ic"].append(c["ic"])
            out["cand_coef0"].append(c0)
            out["cand_support"].append(np.asarray(c["support"]))
            out["cand_beta"].append(b)
        b, c0 = self._denorm(best["support"], best["beta"], best["coef0"], gs_variant)
        beta = np.zeros(self.s.p)
        beta[best["support"]] = b
        out.update({"beta": beta, "coef0": c0, "train_loss": best["loss"], "ic": best["ic"],
                    "lambda": best["lambda"], "best_T0": best["T0"], "best_iters": best["iters"],
                    "n_candidates": len(cands), "n_fits": self.n_fits, "n_pdas_iters": self.n_pdas_iters,
                    "evaluations": self.evaluations})
        for k in ("cand_T0", "cand_iters"):
            out[k] = np.asarray(out[k], dtype=np.int32)
        for k in ("cand_lambda", "cand_train_loss", "cand_ic", "cand_coef0"):
            out[k] = np.asarray(out[k], dtype=np.float64)
        return out

    # -- sequential_path, src/path.cpp:25-132 --------------------------------------------------------------
    def sequential_path(self, sequence, lambda_seq=(0.0,)):
        seq, lam = [int(v) for v in sequence], [float(v) for v in lambda_seq]
        ns, nl = len(seq), len(lam)
        full_init, coef0_init = (np.zeros(0, np.int32), np.zeros(0)), 0.0
        grid, cands = {}, []
        cold = {}
        if not self.warm:  # independent candidates: one batch per lambda, (s x unit) pairs over all ranks
            for j in range(nl):
                for i, recs in enumerate(self._round_cold(seq, lam[j])):
                    cold[(i, j)] = recs
        for i in range(ns):
            order = range(nl) if i % 2 == 0 else range(nl - 1, -1, -1)  # snake order, :50
            for j in order:
                recs = cold[(i, j)] if cold else self._round(seq[i], lam[j], True, True, full_init, coef0_init)
                full = recs[self.K]
                if self.warm:
                    full_init, coef0_init = (full["support"], full["beta"]), full["coef0"]
                c = {"T0": seq[i], "lambda": lam[j], "support": full["support"], "beta": full["beta"],
                     "coef0": full["coef0"], "iters": full["iters"], "loss": full["train_loss"],
                     "ic": self._cv_loss(recs, self.K)}
                grid[j * ns + i] = c
                cands.append(c)
        best = grid[0]
        for q in range(ns * nl):  # minCoeff over the column-major (ns x nl) matrix: first minimum, :113
            if grid[q]["ic"] < best["ic"]:
                best = grid[q]
        return self._result(cands, best, False)

    # -- gs_path, src/path.cpp:134-389 ---------------------------------------------------------------------
    def gs_path(self, s_min, s_max):
        st = {"full_init": (np.zeros(0, np.int32), np.zeros(0)), "coef0_init": 0.0}
        cands = []

        def fit_point(T, twice):
            # the full-data fit and the first ic() are independent: one round; the second ic() (:204+:210,
            # :245+:253, :286+:294) continues the fold chains, a second round without the full-data unit
            if not self.warm:
                # cold starts: the second ic() repeats the first one's K fits exactly (same start, same rows); it is
                # counted (the reference runs it) but not recomputed
                recs = self._round_cold([T], 0.0)[0]
                full = recs[self.K]
                first = self._cv_loss(recs, self.K)
                cands.append({"T0": T, "lambda": 0.0, "support": full["support"], "beta": full["beta"],
                              "coef0": full["coef0"], "iters": full["iters"], "loss": full["train_loss"],
                              "ic": first})
                if twice:
                    self.n_fits += self.K
                    self.n_pdas_iters += sum(recs[k]["iters"] for k in range(self.K))
                return first, (first if twice else None)
            coef0_prev = st["coef0_init"]
            recs = self._round(T, 0.0, True, True, st["full_init"], coef0_prev)
            full = recs[self.K]
            if self.warm:
                st["full_init"], st["coef0_init"] = (full["support"], full["beta"]), full["coef0"]
            first = self._cv_loss(recs, self.K)
            cands.append({"T0": T, "lambda": 0.0, "support": full["support"], "beta": full["beta"],
                          "coef0": full["coef0"], "iters": full["iters"], "loss": full["train_loss"], "ic": first})
            second = None
            if twice:
                second = self._cv_loss(self._round(T, 0.0, False, True, None, coef0_prev), self.K)
            return first, second

        rnd = lambda v: int(np.floor(v + 0.5)) if v >= 0 else -int(np.floor(-v + 0.5))  # std::round
        Tmin, Tmax = int(s_min), int(s_max)
        T1, T2 = rnd(0.618 * Tmin + 0.382 * Tmax), rnd(0.382 * Tmin + 0.618 * Tmax)
        ic1, _ = fit_point(T1, False)
        icT1 = ic1
        ic2, icT2 = fit_point(T2, True)
        while T1 != T2:
            if icT1 < icT2:
                Tmax, T2, ic2, icT2 = T2, T1, ic1, ic1
                T1 = rnd(0.618 * Tmin + 0.382 * Tmax)
                ic1, icT1 = fit_point(T1, True)
            else:
                Tmin, T1, ic1, icT1 = T1, T2, ic2, ic2
                T2 = rnd(0.382 * Tmin + 0.618 * Tmax)
                ic2, icT2 = fit_point(T2, True)
        best = {"T0": 0, "lambda": 0.0, "support": np.zeros(0, np.int32), "beta": np.zeros(0), "coef0": 0.0,
                "loss": 0.0, "ic": np.finfo(np.float64).max, "iters": 0}
        sweep = None if self.warm else self._round_cold(list(range(Tmin, Tmax + 1)), 0.0)  # (fold x s) pairs
        for T in range(Tmin, Tmax + 1):
            coef0_prev = st["coef0_init"]
            recs = sweep[T - Tmin] if sweep is not None else \
                self._round(T, 0.0, True, True, st["full_init"], coef0_prev)
            full = recs[self.K]
            if self.warm:
                st["full_init"], st["coef0_init"] = (full["support"], full["beta"]), full["coef0"]
            v = self._cv_loss(recs, self.K)
            if v < best["ic"]:
                last = recs[self.K - 1]  # read AFTER ic(): the last fold's fit, src/path.cpp:314-319
                best = {"T0": T, "lambda": 0.0, "support": last["support"], "beta": last["beta"],
                        "coef0": last["coef0"], "loss": last["train_loss"], "ic": v, "iters": full["iters"]}
                cands.append(best)
        return self._result(cands, best, True)
