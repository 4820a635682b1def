"""Data-parallel sharding of utterances over the GPUs of one node.

Utterances are independent (weights replicated, one Philox row block per *global* utterance index),
so the batch is split contiguously over ranks with no data-path collective; the only exchange is one
all-gather of the generated int32 token grid at the end (RCCL over xGMI when the backend is "nccl";
gloo in the CPU tests).  Results are independent of the number of ranks: the noise rows are keyed by the global utterance
index, every GEMM schedule is bit-compatible with every other, and the one choice that is not -- the instruction shape of the
attention kernels, picked by batch size -- is made for the GLOBAL batch on every rank (`global_batch=`, d3pm_tuning.regime_batch).
tests/test_gpu_batch_sweep.py holds a shard against the unsplit batch bit for bit.
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
    """Every rank passes the same global lists and gets back the same int64 [B, canvas] grid ([B, canvas, n_q] for a model
    built with n_q > 1)."""
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    B = len(text_list)
    lo, hi = shard_bounds(B, world, rank)
    fn = generate_fn or model.generate_audio
    if generate_fn is None:
        kw = dict(kw, global_batch=B)          # a shard takes the attention kernels of the unsplit batch: same ids at any rank count
    if hi > lo:
        local = fn(list(text_list[lo:hi]), list(proms_list[lo:hi]), seed=seed, utt0=lo, **kw)
        local = local.reshape(hi - lo, -1).to(torch.int32)
    else:
        local = None
    n_q = max(1, getattr(model.cfg, "n_q", 1))
    shape = (-1, model.cfg.canvas) if n_q == 1 else (-1, model.cfg.canvas, n_q)
    if world == 1:
        return local.long().reshape(shape)
    canvas = model.cfg.canvas * n_q
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
    return torch.cat(parts).long().reshape(shape)


def generate_codes_dp(ar, nar, text_list: Sequence[torch.Tensor], proms_list: Sequence[torch.Tensor], *, seed: int,
                      group=None, ar_fn: Optional[Callable] = None, nar_fn: Optional[Callable] = None,
                      **kw) -> torch.Tensor:
    """Both stages of the reference's inference script (/root/reference/vall_e/__main__.py:60-71) sharded the same way:
    every rank runs the D3PM stage and then the NAR stage (levels 1..7) for its contiguous slice of the utterances --
    noise keyed by the global utterance index in both -- and ONE all-gather returns the int64 [B, n_frames, 8] codes
    to every rank.  No collective sits between the stages: a rank's NAR input is its own D3PM output."""
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    B = len(text_list)
    lo, hi = shard_bounds(B, world, rank)
    n_frames, levels = ar.cfg.n_frames, nar.n_resp_levels + 1
    gen = ar_fn or ar.generate_audio
    fill = nar_fn or nar
    if ar_fn is None:
        kw = dict(kw, global_batch=B)
    local = None
    if hi > lo:
        texts, proms = list(text_list[lo:hi]), list(proms_list[lo:hi])
        lvl0 = gen(texts, proms, seed=seed, utt0=lo, **kw).reshape(hi - lo, -1)[:, :n_frames]
        resps = [lvl0[b].clamp(max=nar.n_tokens - 1).reshape(-1, 1) for b in range(hi - lo)]
        full = fill(texts, proms, resps, seed=seed, utt0=lo)
        local = torch.stack([f.reshape(n_frames, levels) for f in full]).to(torch.int32)
    if world == 1:
        return local.long()
    dev = local.device if local is not None else ar.device
    per = -(-B // world)
    send = torch.zeros((per, n_frames, levels), dtype=torch.int32, device=dev)
    if local is not None:
        send[: hi - lo] = local
    recv = torch.empty((world * per, n_frames, levels), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(recv, send, group=group)
    parts = []
    for r in range(world):
        a, b = shard_bounds(B, world, r)
        parts.append(recv[r * per: r * per + (b - a)])
    return torch.cat(parts).long()
