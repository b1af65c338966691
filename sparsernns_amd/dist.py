"""Multi-GPU use of the fixed-point S5 path: one process per GPU, ``torch.distributed`` (RCCL on ROCm).

The reference has no distributed code at all.  Batches of independent
sequences shard across ranks; weights (< 1 MB) are replicated.  The only coupling between
sequences is the data-dependent exponent of the five ``compute_best`` ops per layer
(fxparray.py:420-448, 601-609; fxpmodel.py:892-933, 1147-1152), whose float32 maxima span the batch:

* mode B "per-shard" (default): every rank treats its shard as one reference batch -- exactly what the
  reference computes when it is handed that shard (its recipe batch size is 32, recipes/ndns.json).
  No collective on the data path.
* mode A "global": the maxima are combined with ``all_reduce(MAX)`` so N ranks reproduce, bit for bit, one
  reference run over the concatenated batch.  What the fast path exchanges per layer: the 2H per-channel
  extremes of the layer input (every BatchNorm stage is monotone per channel, so the four BatchNorm maxima
  follow from them -- csrc/mfma_bn.hpp) and the 3 maxima of the residual add: two collectives per layer.
  The generic path exchanges the <= 3 maxima of each of its five compute_best ops.

Outputs are gathered with one ``all_gather`` (RCCL over xGMI) when the caller wants them on every rank.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced split of `total` sequences: the first (total % world) ranks get one more."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world: {rank}/{world}")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def make_exponent_allreduce(group: Optional[dist.ProcessGroup] = None, via_host: bool = False) -> Callable[[torch.Tensor], None]:
    """Hook for ``Engine.enqueue(..., allreduce=...)``: element-wise MAX over ranks of the float32 maxima
    of one compute_best op.  The tensor is a view into the engine's workspace; the collective is
    stream-ordered behind the reduction kernel that produced it.

    via_host: for backends that cannot reduce device memory (gloo rehearsals of several ranks on one GPU):
    the values make a round trip through host memory, which synchronises the stream once per exchange."""

    def hook(maxima: torch.Tensor) -> None:
        if via_host:
            t = maxima.cpu()
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            maxima.copy_(t)
        else:
            dist.all_reduce(maxima, op=dist.ReduceOp.MAX, group=group)

    return hook


def gather_outputs(y_local: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """all_gather of equally sized (B_local, L, d_out) int32 outputs -> (world*B_local, L, d_out)."""
    world = dist.get_world_size(group)
    out = torch.empty((world * y_local.shape[0],) + tuple(y_local.shape[1:]), dtype=y_local.dtype, device=y_local.device)
    dist.all_gather_into_tensor(out, y_local.contiguous(), group=group)
    return out
