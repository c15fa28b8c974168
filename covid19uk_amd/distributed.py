"""Multi-GPU layout: independent chains shard over ranks (one process per GPU,
`torch.distributed`; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for tests).

The sweep needs no data-path collective: chain c of the job always draws from the
Philox stream keyed by (seed, c), wherever it runs (first_chain_id of the sampler).
The only exchange is the optional pooling of the adapted HMC step size across all
chains at window boundaries (one float64 per chain), and the max-over-ranks of a
timing.  The reference itself is single-chain, single-process
(covid19uk/inference/inference.py:563-576); pooling is off unless asked for.
"""
from __future__ import annotations

import numpy as np


def world():
    """(rank, world_size) of the initialised process group, or (0, 1)."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except ImportError:                                   # pragma: no cover
        pass
    return 0, 1


def shard_chains(total_chains: int, world_size: int, rank: int):
    """Contiguous block of global chain ids for `rank`: (first_chain_id, count).
    Blocks differ by at most one chain; with total = k * world every rank gets k."""
    if not (0 <= rank < world_size) or total_chains < 0:
        raise ValueError("bad rank/world/total")
    base, extra = divmod(total_chains, world_size)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def _device_for_collectives(device=None):
    import torch
    import torch.distributed as dist
    if dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device() if device is None else device)
    return torch.device("cpu")


def gather_chain_values(local_values, device=None):
    """All-gather one float64 per local chain -> array over all chains of the job, in global
    chain order (ranks may hold different counts)."""
    local = np.ascontiguousarray(local_values, dtype=np.float64).reshape(-1)
    rank, ws = world()
    if ws == 1:
        return local.copy()
    import torch
    import torch.distributed as dist
    dev = _device_for_collectives(device)
    counts = torch.zeros(ws, dtype=torch.int64, device=dev)
    counts[rank] = local.size
    dist.all_reduce(counts)
    n_max = int(counts.max())
    buf = torch.zeros(n_max, dtype=torch.float64, device=dev)
    buf[:local.size] = torch.from_numpy(local).to(dev)
    out = [torch.empty_like(buf) for _ in range(ws)]
    dist.all_gather(out, buf)
    return np.concatenate([o[:int(c)].cpu().numpy() for o, c in zip(out, counts.cpu())])


def pool_step_sizes(local_step_sizes, device=None):
    """Geometric mean of the step sizes of ALL chains of the job (mean of log eps)."""
    eps = gather_chain_values(np.log(np.asarray(local_step_sizes, dtype=np.float64)), device)
    return float(np.exp(eps.mean()))


def max_over_ranks(value: float, device=None) -> float:
    rank, ws = world()
    if ws == 1:
        return float(value)
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=_device_for_collectives(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.cpu())
