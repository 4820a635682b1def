"""Data-parallel sharding of utterances over the GPUs of one node.

Utterances are independent (weights replicated, one Philox row block per *global* utterance index),
so the batch is split contiguously over ranks with no data-path collective; the only exchange is one
all-gather of the generated int32 token grid at the end (RCCL over xGMI when the backend is "nccl";
gloo in the CPU tests).  Results are independent of the number of ranks by construction.
The reference has no multi-GPU inference (its only distributed code is DeepSpeed DP training,
/root/reference/vall_e/train.py:29-31); this module is new in the build.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import torch
import torch.distributed as dist


def shard_bounds(n_items: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous split; the first n_items % world ranks get one extra utterance."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def generate_audio_dp(model, text_list: Sequence[torch.Tensor], proms_list: Sequence[torch.Tensor], *, seed: int,
                      group=None, generate_fn: Optional[Callable] = None, **kw) -> torch.Tensor:
    """Every rank passes the same global lists and gets back the same int64 [B, canvas] grid."""
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    B = len(text_list)
    lo, hi = shard_bounds(B, world, rank)
    fn = generate_fn or model.generate_audio
    if hi > lo:
        local = fn(list(text_list[lo:hi]), list(proms_list[lo:hi]), seed=seed, utt0=lo, **kw)
        local = local.reshape(hi - lo, -1).to(torch.int32)
    else:
        local = None
    if world == 1:
        return local.long()
    canvas = model.cfg.canvas
    dev = local.device if local is not None else model.device
    per = -(-B // world)                                   # padded shard so one fixed-size all-gather suffices
    send = torch.zeros((per, canvas), dtype=torch.int32, device=dev)
    if local is not None:
        send[: hi - lo] = local
    recv = torch.empty((world * per, canvas), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(recv, send, group=group)
    parts = []
    for r in range(world):
        a, b = shard_bounds(B, world, r)
        parts.append(recv[r * per: r * per + (b - a)])
    return torch.cat(parts).long()
