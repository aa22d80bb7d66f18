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
    last = {}
    for r in range(1, world):
        lo = bdist.partition(kmax, world, r)[0]
        h = s.sequential_path_chain(seq[:lo], ic_type=3)
        last[r] = (h["last_idx"], h["last_val"], h["last_coef0"])
    scores = s.marginal_scores()
    settings = [(0, None, "cold"), (0, None, "ladder"), (320, None, "cold"), (320, None, "ladder"), (384, None, "cold")]
    settings += [(128, (96, 256), "pilot"), (128, (96, 384), "pilot"), (160, (128, 256), "pilot"), (128, (96, 512), "pilot")]
    for M, pilot, start in settings:
        ng = M // 32
        blocks = blocks2 = cols = cols2 = None
        pmodel = None
        if ng:  # every rank's share computed once up front (untimed): what the all-gather would deliver
            cols = np.argsort(-scores, kind="stable")[:M].astype(np.int32)
            s.cov_prefill_begin(cols)
            s.cov_prefill_compute(0, ng)
            blocks = torch.empty(ng * 32 * p, dtype=torch.float64, device="cuda")
            s.cov_prefill_export(0, ng, device_ptr=blocks.data_ptr())
            s.cov_prefill_end()
            if pilot:
                pm = s.sequential_path_chain([pilot[0]], ic_type=3, keep_caches=True)
                pmodel = (pm["last_idx"], pm["last_val"], pm["last_coef0"])
                bd, slot = s.cov_state()
                ng2 = pilot[1] // 32
                cols2 = np.argsort(-np.where(slot >= 0, -np.inf, bd), kind="stable")[:ng2 * 32].astype(np.int32)
                s.cov_prefill_extend(cols2)
                s.cov_prefill_compute(0, ng2)
                blocks2 = torch.empty(ng2 * 32 * p, dtype=torch.float64, device="cuda")
                s.cov_prefill_export(0, ng2, device_ptr=blocks2.data_ptr())
                s.cov_prefill_end()
        per_rank = []
        for r in range(world):
            lo, hi = bdist.partition(kmax, world, r)
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
            if ng:
                shared_fill(cols, blocks, True)
                if pilot:
                    pm = s.sequential_path_chain([pilot[0]], ic_type=3, keep_caches=True)
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
                                              keep_caches=bool(ng))
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
        slow = max(q["ms"] for q in per_rank)
        nbytes = ng * 32 * p * 8 + (pilot[1] * p * 8 if pilot else 0)
        print(json.dumps({"world": world, "prefill_columns": M, "pilot": list(pilot) if pilot else None, "chunk_start": start,
                          "single_chain_ms": round(1e3 * t_single, 3), "slowest_rank_ms": slow,
                          "speedup_estimate": round(1e3 * t_single / slow, 2), "all_gather_bytes_per_rank": nbytes,
                          "supports_equal": sum(q["equal_to_single_chain"] for q in per_rank), "of": kmax,
                          "per_rank": per_rank}))
        sys.stdout.flush()
