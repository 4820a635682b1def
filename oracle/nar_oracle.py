"""TEST INFRASTRUCTURE -- CPU oracle for the stock NAR model that fills quantizer levels 1..7 after the D3PM
sampler.  Not product code; same import rules as d3pm_oracle.py.

Functional PyTorch-CPU restatement of the inference branch of the reference's
/root/reference/vall_e/vall_e/nar.py:76-101 (`NAR.forward` with one given level) and of
/root/reference/vall_e/vall_e/base.py (`Base.forward` :403-499, `Block`/`PrenormResidual` :161-234,
`AdaLN` :136-158, `Attention` :92-133, `SinusodialEmbedding` :38-89, `MultiEmbedding` :244-274,
`_join`/`list_to_tensor` :19-35,277-286).

Parity status: PINNED -- tests/golden/make_golden.py (gen_nar) runs the reference NAR and this oracle on the same
weights / inputs / torch seed and asserts bit-equality of logits and of the Categorical samples; the reference's
outputs are committed in tests/golden/nar_small.npz.

Sampling: the reference draws `Categorical(logits=h/T).sample()` from torch's global generator, whose stream
cannot be reproduced on a GPU.  `sample_gumbel` is the build-defined equivalent (Gumbel-max over the Philox
stream 2 of oracle/philox.py -- same distribution, different random numbers); `sample_torch` is the reference's own
call, used only to pin this oracle to the reference.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch.distributions import Categorical

from . import philox

STREAM_NAR = 2
ADALN_K, ADALN_C, ADALN_EPS = 0.1, 2, 1e-5


def sinusoid_pe(n: int, d_model: int, dtype: torch.dtype) -> torch.Tensor:
    """base.py:38-89 -- omega is computed in fp32 here (fp16 in the D3PM file) and follows the module dtype."""
    half = d_model // 2
    omega = torch.exp(-math.log(1e4) * (torch.arange(half, dtype=torch.float32) / half)).to(dtype)
    ang = omega[None, :] * torch.arange(n)[:, None]
    return torch.cat([ang.sin(), ang.cos()], dim=-1)


def level_sum_embedding(w: torch.Tensor, codes: torch.Tensor) -> torch.Tensor:
    """MultiEmbedding: codes [t, l<=L], w [L, K, d] -> [t, d] via the one-hot contraction upstream uses."""
    oh = F.one_hot(codes, num_classes=w.shape[1])
    oh = F.pad(oh, (0, 0, 0, w.shape[0] - oh.shape[1])).to(w)
    return torch.einsum("l k d, n l k -> n d", w, oh)


def merged_sequence(sd, text, prom, resp):
    """[text | sep | prompt | sep | response] embeddings of one utterance (base.py:277-286,441-446)."""
    sep = sd["sep"][None]
    return torch.cat([F.embedding(text, sd["text_emb.weight"]), sep, level_sum_embedding(sd["proms_emb.weight"], prom), sep,
                      level_sum_embedding(sd["resps_emb.weight"], resp)], dim=0)


def adaln(x, emb_row):
    """AdaLN.forward with one level for the whole batch: emb_row [2d] = (log gamma | beta)."""
    d = x.shape[-1]
    log_g, beta = emb_row[:d], emb_row[d:]
    h = F.layer_norm(x, x.shape[-1:], eps=ADALN_EPS)
    h = ADALN_C * (1 - (ADALN_K * h)) * h
    return log_g.exp() * h + beta


def attention(sd, p, x, m, n_heads):
    """Attention.forward (non-causal): x [b,t,c], m [b,t,1]."""
    b, t, c = x.shape
    hd = c // n_heads
    q, k, v = F.linear(x, sd[p + ".to_qkv.weight"]).chunk(3, dim=-1)
    q, k, v = (z.reshape(b, t, n_heads, hd) for z in (q, k, v))
    e = torch.einsum("b i h d, b j h d -> b i j h", q, k) * (hd ** -0.5)
    kpm = m.unsqueeze(1) * m.unsqueeze(2)
    e = e.masked_fill(kpm == 0, -torch.finfo(e.dtype).max)
    a = e.softmax(dim=2)
    o = torch.einsum("b i j h, b j h d -> b i h d", a, v).flatten(-2)
    return F.linear(o, sd[p + ".to_out.weight"], sd[p + ".to_out.bias"]) * m


def level_logits(sd, n_heads: int, n_layers: int, text_list, proms_list, resps_list, level: int):
    """One `Base.forward` pass at quantizer `level`: returns the classifier logits of the response rows of every
    utterance (list of [t_resp, n_tokens]) -- everything before the Categorical draw."""
    dtype = sd["classifier.weight"].dtype
    xs = [merged_sequence(sd, t, p, r) for t, p, r in zip(text_list, proms_list, resps_list)]
    lens = [len(x) for x in xs]
    T = max(lens)
    # upstream pads sequence-first and *views* the result batch-first (list_to_tensor, base.py:19-35): the tensor
    # stays (t, b, c)-ordered in memory, which sends every biased nn.Linear down torch's non-fused matmul + add_
    # path (two roundings in fp16).  Reproduce the layout, not just the values.
    x = torch.stack([F.pad(z, (0, 0, 0, T - len(z))) for z in xs], dim=1).permute(1, 0, 2)        # [b, T, d] view
    m = torch.stack([(torch.arange(T) < n) for n in lens]).float().t().unsqueeze(-1).permute(1, 0, 2).to(dtype)
    x = x + sinusoid_pe(T, x.shape[-1], dtype)[None]
    for i in range(n_layers):
        pa, pf = f"blocks.{i}.attn", f"blocks.{i}.ffn"
        x = (x + attention(sd, pa + ".block", adaln(x, sd[pa + ".norm.emb.weight"][level]) * m, m, n_heads)) * m
        h = adaln(x, sd[pf + ".norm.emb.weight"][level]) * m
        h = F.linear(F.gelu(F.linear(h, sd[pf + ".block.0.weight"], sd[pf + ".block.0.bias"])),
                     sd[pf + ".block.3.weight"], sd[pf + ".block.3.bias"])
        x = (x + h) * m
    h = F.linear(x, sd["classifier.weight"], sd["classifier.bias"]) * m
    return [h[b, lens[b] - len(r): lens[b]] for b, r in enumerate(resps_list)]


def sample_torch(logits_list, temperature: float):
    """The reference's own draw (torch global generator), base.py:491-494."""
    return [Categorical(logits=h / temperature).sample() for h in logits_list]


def sample_gumbel(logits_list, temperature: float, seed: int, level: int, utt0: int = 0, greedy: bool = False):
    """Gumbel-max over the Philox stream: counter row = (utt0 + b) * 65536 + frame, t = level, stream 2."""
    out = []
    for b, h in enumerate(logits_list):
        z = h.float() / temperature
        if not greedy:
            u = torch.from_numpy(philox.uniform_rows(seed, level, (utt0 + b) * 65536, h.shape[0], h.shape[1], STREAM_NAR))
            u = torch.clamp(u, min=torch.finfo(torch.float32).tiny, max=1.0)
            z = z - torch.log(-torch.log(u))
        out.append(torch.argmax(z, dim=-1))
    return out


def generate(sd, n_heads, n_layers, text_list, proms_list, level0_list, temperature=0.2, sampler="torch", seed=0,
             n_levels: int = 7):
    """NAR.forward inference loop (nar.py:76-101): level0_list = [codes[t]] -> [codes[t, 8]]."""
    prev = [r.reshape(-1, 1) if r.dim() == 1 else r for r in level0_list]
    with torch.no_grad():
        while prev[0].shape[-1] - 1 < n_levels:
            level = prev[0].shape[-1] - 1
            lg = level_logits(sd, n_heads, n_layers, text_list, proms_list, prev, level)
            new = sample_torch(lg, temperature) if sampler == "torch" else \
                sample_gumbel(lg, temperature, seed, level, greedy=(sampler == "greedy"))
            prev = [torch.cat([rs, r.unsqueeze(-1)], dim=-1) for rs, r in zip(prev, new)]
    return prev
