"""TEST INFRASTRUCTURE -- CPU restatement of the build's n_q > 1 EXTENSION of the D3PM sampler.  Not product code.

Parity status: NOT a reference oracle.  The reference generates quantizer level 0 only
(/root/reference/vall_e/vall_e/ar_discrete.py:699-709); BASELINE.json configs[1] ("x 8 quantizers")
and SURVEY.md section 8d config 2 ask for an n_q = 8 extension and state that no reference oracle exists for it.  This
file is the build's own DEFINITION of that extension written as plain PyTorch-CPU code on top of oracle/d3pm_oracle.py --
every operation of the denoiser, the posterior and the draw is the pinned oracle's, applied per level -- so that the HIP
path can be checked against something that is not itself:

  * input embedding of a frame = sum over its n_q levels of `resps_emb.weight[l][x_t[frame, l]]`, accumulated in fp32 in
    level order and rounded once (what MultiEmbedding does for the prompt levels, base.py:244-274; d3pm_oracle.prompt_embedding);
  * the DiT blocks are unchanged (rows = frames);
  * `final` has n_q * K outputs: logits[frame, l, :] = final.weight[l K : (l + 1) K] . h + final.bias[l K : (l + 1) K];
  * every (frame, level) token is sampled by the level-0 rule (posterior_logits_closed + gumbel_argmax) from its own K logits
    and its own x_t; level 0 of a frame draws the uniforms the level-0-only path draws (Philox stream 0, row = utt * canvas +
    frame), level l > 0 the same counter on stream 16 + l.

With n_q = 1 every function here reduces to d3pm_oracle's (tests/test_oracle_golden.py checks it), which is the parity
statement VERDICT round 2 asks for: "the n_q = 1 slice bit-identical to today's path".
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F

from . import d3pm_oracle as O
from . import philox

K = O.K_CLASSES


def stream_of(level: int) -> int:
    return 0 if level == 0 else 16 + level


def embed(sd, x_t: torch.Tensor) -> torch.Tensor:
    """x_t int [T, n_q] -> [T, d]: fp32 sum over the levels, one rounding.  n_q = 1 with a [K, d] table: the plain gather."""
    w = sd["resps_emb.weight"]
    if w.dim() == 2:
        return F.embedding(x_t.reshape(-1).long(), w)
    acc = torch.zeros(x_t.shape[0], w.shape[-1], dtype=torch.float32)
    for l in range(w.shape[0]):
        acc += F.embedding(x_t[:, l].long(), w[l]).float()
    return acc.to(w.dtype)


def logits(sd, shape: O.Shape, x_t, t: int, cond_prompt, cond_text, mask) -> torch.Tensor:
    """-> x0-logits [T, n_q, K] in the dtype of sd."""
    n_q = sd["final.weight"].shape[0] // K
    t_emb = F.embedding(torch.tensor([t]), sd["time_emb.weight"])
    x = embed(sd, x_t.reshape(x_t.shape[0], -1))[None]
    for i in range(shape.n_layers):
        x = O.dit_block(sd, i, x, cond_prompt[None], cond_text[None], t_emb, mask, shape)
    h = x * mask[None, :, None]
    out = []
    for l in range(n_q):      # one Linear per level, like the product: the K-dot products are what they are in either form
        out.append(F.linear(h, sd["final.weight"][l * K:(l + 1) * K], sd["final.bias"][l * K:(l + 1) * K])[0])
    return torch.stack(out, dim=1)


def step(sd, shape: O.Shape, tabs, x_t, t: int, cond_prompt, cond_text, mask, seed: int, utt: int = 0, greedy: bool = False):
    """One reverse step for one utterance: x_t int [T, n_q] -> x_{t-1} int64 [T, n_q]."""
    lg = logits(sd, shape, x_t, t, cond_prompt, cond_text, mask)
    T, n_q = lg.shape[0], lg.shape[1]
    out = torch.empty(T, n_q, dtype=torch.int64)
    for l in range(n_q):
        post = O.posterior_logits_closed(lg[:, l].to(torch.float16), x_t[:, l], t, tabs)
        if greedy:
            out[:, l] = torch.argmax(post, dim=-1)
        else:
            u = torch.from_numpy(philox.uniform_rows(seed, t, utt * T, T, K, stream_of(l)))
            out[:, l] = O.gumbel_argmax(post, u, t)
    return out


def generate(sd, shape: O.Shape, text, prompt, seed: int, utt: int = 0, t_start: Optional[int] = None, t_stop: int = 0,
             trace: Optional[list] = None):
    n_q = sd["final.weight"].shape[0] // K
    betas = O.cosine_betas(shape.timesteps)
    tabs = O.scalar_tables(betas, shape.timesteps)
    with torch.no_grad():
        x = torch.zeros(shape.canvas, n_q, dtype=torch.int64)
        x[: shape.n_frames] = O.MASK_ID
        mask = x[:, 0] != 0
        cp, ct = O.encode_conditions(sd, shape, text, prompt)
        t_start = shape.timesteps - 1 if t_start is None else t_start
        for t in range(t_start, t_stop, -1):
            x = step(sd, shape, tabs, x, t, cp, ct, mask, seed, utt)
            if trace is not None:
                trace.append(x.clone())
    return x
