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
