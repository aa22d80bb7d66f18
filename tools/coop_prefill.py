"""One-GPU rehearsal of the 8-rank k-path of configs[1] with the cooperative prefill (bess_amd/dist.py): every rank's
step -- its share of the prefill passes, the import of the other ranks' Gram column blocks (device-to-device copies
stand in for the RCCL all-gather, which one GPU cannot run), its chunk from a cold / ladder start on the prefilled
cache, and its stitch onto the predecessor's last model -- timed one rank after another on the same device.  The slowest
rank bounds the step of an 8-GPU run.  Prints one JSON line per setting.

    python tools/coop_prefill.py [world] [n p kmax]
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from bess_amd import capi, synth  # noqa: E402
from bess_amd import dist as bdist  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n, p, kmax = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (50000, 10000, 200)
X, y, _, _ = synth.make_lm(n, p, min(100, kmax // 2))
seq = np.arange(1, kmax + 1)
with capi.Session(X, y) as s:
    del X
    single = s.sequential_path(seq, ic_type=3)
    t0 = time.time()
    single = s.sequential_path(seq, ic_type=3)
    t_single = time.time() - t0
    # the model every chunk's predecessor ends with (normalised scale), from the single chain walked in links
    last_at = {}

    def last_model(lo):  # the single chain's model in front of candidate index lo
        if lo not in last_at:
            h = s.sequential_path_chain(seq[:lo], ic_type=3)
            last_at[lo] = (h["last_idx"], h["last_val"], h["last_coef0"])
        return last_at[lo]

    REB = int(os.environ.get("COOP_REBALANCE", "0"))  # rounds of dist.rebalance_bounds on the measured chunk times
    scores = s.marginal_scores()
    # (columns of the marginal list, (pilot level, columns of the list behind the pilot[, width of the shared wide fills
    # INSIDE the pilot fit]), chunk start)
    settings = [(0, None, "cold"), (0, None, "ladder"), (320, None, "cold"), (320, None, "ladder"), (384, None, "cold")]
    settings += [(128, (96, 256), "pilot"), (128, (96, 384), "pilot"), (160, (128, 256), "pilot"), (128, (96, 512), "pilot")]
    W = int(os.environ.get("COOP_W", 32 * world))
    settings += [(128, (128, 0, W), "pilot"), (160, (128, 0, W), "pilot"), (0, (128, 0, W), "pilot"),
                 (0, (128, 128, W), "pilot"), (160, (128, 256, W), "pilot")]
    if os.environ.get("COOP_ONLY_WIDE") == "1":
        settings = [q for q in settings if q[1] and len(q[1]) > 2]
    for M, pilot, start in settings:
        ng = M // 32
        blocks = blocks2 = cols = cols2 = None
        pmodel = None
        wide = pilot[2] if pilot and len(pilot) > 2 else 0
        recorded = []  # wide fills of the pilot fit, in the order they happen: the blocks of ALL groups of each

        def rec_hook(n_g):
            s.cov_prefill_compute(0, n_g)
            blk = torch.empty(n_g * 32 * p, dtype=torch.float64, device="cuda")
            s.cov_prefill_export(0, n_g, device_ptr=blk.data_ptr())
            s.cov_prefill_end()
            recorded.append(blk)
        if ng or wide:  # every rank's share computed once up front (untimed): what the all-gather would deliver
            if ng:
                cols = np.argsort(-scores, kind="stable")[:M].astype(np.int32)
                s.cov_prefill_begin(cols)
                s.cov_prefill_compute(0, ng)
                blocks = torch.empty(ng * 32 * p, dtype=torch.float64, device="cuda")
                s.cov_prefill_export(0, ng, device_ptr=blocks.data_ptr())
                s.cov_prefill_end()
            if pilot:
                if wide:
                    s.set_fill_hook(rec_hook, wide)
                pm = s.sequential_path_chain([pilot[0]], ic_type=3, keep_caches=bool(ng))
                s.set_fill_hook(None)
                pmodel = (pm["last_idx"], pm["last_val"], pm["last_coef0"])
                bd, slot = s.cov_state()
                ng2 = pilot[1] // 32
                if ng2:
                    cols2 = np.argsort(-np.where(slot >= 0, -np.inf, bd), kind="stable")[:ng2 * 32].astype(np.int32)
                    s.cov_prefill_extend(cols2)
                    s.cov_prefill_compute(0, ng2)
                    blocks2 = torch.empty(ng2 * 32 * p, dtype=torch.float64, device="cuda")
                    s.cov_prefill_export(0, ng2, device_ptr=blocks2.data_ptr())
                    s.cov_prefill_end()
        bounds = [bdist.partition(kmax, world, r)[0] for r in range(world)] + [kmax]
        for reb in range(REB + 1):
            last = {r: last_model(bounds[r]) for r in range(1, world)}
            per_rank = []
            for r in range(world):
                lo, hi = bounds[r], bounds[r + 1]
                k0 = int(seq[lo])
                lead = sorted({k for k in (k0 // 8, k0 // 4, k0 // 2) if 1 <= k < k0}) if (start == "ladder" and lo > 0) else []
                torch.cuda.synchronize()
                t0 = time.time()

                def shared_fill(cl, blk, first):
                    n_g = len(cl) // 32
                    (s.cov_prefill_begin if first else s.cov_prefill_extend)(cl)
                    a, b = bdist.partition(n_g, world, r)
                    s.cov_prefill_compute(a, b - a)
                    for q in range(world):
                        c, d = bdist.partition(n_g, world, q)
                        if q != r and d > c:
                            s.cov_prefill_import(c, d - c, device_ptr=blk.data_ptr() + c * 32 * p * 8)
                    s.cov_prefill_end()

                init = None
                if ng or wide:
                    if ng:
                        shared_fill(cols, blocks, True)
                    if pilot:
                        calls = [0]

                        def replay_hook(n_g):  # this rank's share for real, the others' blocks as the all-gather would bring them
                            a, b = bdist.partition(n_g, world, r)
                            s.cov_prefill_compute(a, b - a)
                            blk = recorded[calls[0]]
                            calls[0] += 1
                            for q in range(world):
                                c, d = bdist.partition(n_g, world, q)
                                if q != r and d > c:
                                    s.cov_prefill_import(c, d - c, device_ptr=blk.data_ptr() + c * 32 * p * 8)
                            s.cov_prefill_end()

                        if wide:
                            s.set_fill_hook(replay_hook, wide)
                        pm = s.sequential_path_chain([pilot[0]], ic_type=3, keep_caches=bool(ng))
                        s.set_fill_hook(None)
                        if wide:
                            assert calls[0] == len(recorded)
                        if cols2 is not None:
                            bd, slot = s.cov_state()
                            c2 = np.argsort(-np.where(slot >= 0, -np.inf, bd), kind="stable")[:len(cols2)].astype(np.int32)
                            assert np.array_equal(c2, cols2)  # the pilot is the same fit on every rank
                            shared_fill(cols2, blocks2, False)
                        if k0 > pilot[0]:
                            init = (pm["last_idx"], pm["last_val"], pm["last_coef0"])
                t_pre = time.time() - t0
                if init is not None:
                    lead = []
                    out = s.sequential_path_chain(seq[lo:hi], ic_type=3, keep_caches=True, init_idx=init[0], init_val=init[1],
                                                  init_coef0=init[2])
                else:
                    out = s.sequential_path_chain(np.concatenate([np.array(lead, dtype=seq.dtype), seq[lo:hi]]), ic_type=3,
                                                  keep_caches=bool(ng or wide))
                t_chunk = time.time() - t0 - t_pre
                refits = 0
                nl = len(lead)
                if r > 0:
                    res = s.sequential_path_chain(seq[lo:hi], ic_type=3, init_idx=last[r][0], init_val=last[r][1],
                                                  init_coef0=last[r][2], keep_caches=True,
                                                  stop_support=out["cand_support"][nl:], stop_beta=out["cand_beta"][nl:])
                    refits = int(res["n_candidates"])
                    sup = out["cand_support"][nl:].copy()
                    sup[:refits] = -1
                    sup[:refits, :res["cand_support"].shape[1]] = res["cand_support"][:refits]
                else:
                    sup = out["cand_support"]
                t_all = time.time() - t0
                same = int(np.sum([np.array_equal(sup[i, :lo + i + 1], single["cand_support"][lo + i, :lo + i + 1])
                                   for i in range(hi - lo)]))
                per_rank.append({"rank": r, "k": [lo + 1, hi], "ms": round(1e3 * t_all, 3), "prefill_ms": round(1e3 * t_pre, 3),
                                 "chunk_ms": round(1e3 * t_chunk, 3), "stitch_ms": round(1e3 * (t_all - t_pre - t_chunk), 3),
                                 "stitch_refits": refits, "equal_to_single_chain": same, "of": hi - lo})
            used = list(bounds)
            bounds = bdist.rebalance_bounds(bounds, [q["chunk_ms"] + q["stitch_ms"] for q in per_rank],
                                            fixed_seconds=float(np.median([q["prefill_ms"] for q in per_rank])))
            slow = max(q["ms"] for q in per_rank)
            nbytes = ng * 32 * p * 8 + (pilot[1] * p * 8 if pilot else 0) + sum(int(b.numel()) * 8 for b in recorded)
            print(json.dumps({"world": world, "bounds": used, "prefill_columns": M, "pilot": list(pilot) if pilot else None, "wide_fills_in_pilot": len(recorded), "chunk_start": start,
                              "single_chain_ms": round(1e3 * t_single, 3), "slowest_rank_ms": slow,
                              "speedup_estimate": round(1e3 * t_single / slow, 2), "all_gather_bytes_per_rank": nbytes,
                              "supports_equal": sum(q["equal_to_single_chain"] for q in per_rank), "of": kmax,
                              "per_rank": per_rank}))
            sys.stdout.flush()
