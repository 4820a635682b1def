"""Training step of the D3PM model on MI355X: loss AND parameter gradients of `AR.forward` from hand-written HIP kernels,
plus the data-parallel gradient all-reduce.

Reference: the training forward /root/reference/vall_e/vall_e/ar_discrete.py:588-694 (for t = 1 .. timesteps-1: q_sample ->
DiT blocks :126-161 -> final Linear :776 -> masked cross-entropy :683-690, summed over t and divided by mask.sum()), its
gradient by torch autograd inside `engine.backward`, and DeepSpeed's data-parallel all-reduce of the gradients
(/root/reference/vall_e/utils/engines.py:144-147; /root/reference/vall_e/train.py:29-31 initialises the NCCL group).

Here: `forward_backward` replays the forward op by op through the C ABI with a stash of every op input, then walks the stash
backwards through the `d3pm_op_*_bwd_f32` kernels (csrc/d3pm_train.hip); gradients are accumulated into `param.grad`
(fp32), exactly what an optimizer from torch.optim consumes.  `all_reduce_gradients` is one bucketed all-reduce over
`torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box; "gloo" in the CPU tests).

Scope: the F32 precision mode (fp32 weights and activations).  Gradients flow into every parameter the denoiser forward
reads -- the DiT blocks (cross_attn2 and token_emb are dead upstream and stay gradient-free), `final`, `resps_emb`, `time_emb`
via `timestep_fc` -- and, through the cached cross-attention K / V, into the two condition encoders (ar_discrete.py:216-230:
two post-norm TransformerEncoder layers + a SiLU Mlp each) and their embeddings `text_emb` (padding row 0 excluded, as
nn.Embedding(padding_idx=0) does) and `proms_emb`.

Dropout is NOT applied: the step is the eval-mode forward (`model.eval()`), and the gradient fixtures were generated that
way (tests/golden/make_golden.py gen_grads).  The reference trains with dropout active -- p = 0.1 inside both condition
encoders' TransformerEncoderLayers and drop = 0.01 in their Mlp (ar_discrete.py:216-230) -- drawn from torch's global
generator, a stream this build cannot reproduce; the DiT blocks themselves have dropout 0 (:109,123).  A training run
that needs the regulariser has to add it outside this step.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch
import torch.nn.functional as F

from . import _hip

GELU, RELU, SILU = 1, 2, 3


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _f32(t: torch.Tensor, name: str):
    if t.dtype != torch.float32 or not t.is_cuda:
        raise _hip.D3PMError(f"{name}: the training ops take fp32 device tensors")
    return t


def matmul(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, *, ta=False, tb=False, beta=0.0, row_mask=None, period=1):
    """out[M, N] = beta * out + op(a) @ op(b) through d3pm_op_matmul_f32 (2-D fp32 views, any strides)."""
    for t_, n_ in ((a, "a"), (b, "b"), (out, "out")):
        _f32(t_, n_)
    A = a.t() if ta else a
    Bm = b.t() if tb else b
    M, K = A.shape
    K2, N = Bm.shape
    assert K == K2 and tuple(out.shape) == (M, N) and out.stride(1) == 1
    _hip.check(_hip.lib().d3pm_op_matmul_f32(_ptr(A), A.stride(0), A.stride(1), _ptr(Bm), Bm.stride(0), Bm.stride(1), _ptr(out),
                                             out.stride(0), M, N, K, float(beta), _ptr(row_mask), period, _hip.stream_ptr()),
               "d3pm_op_matmul_f32")
    return out


def colsum(x: torch.Tensor, out: torch.Tensor, beta=1.0):
    assert x.stride(1) == 1 and out.numel() == x.shape[1]
    _hip.check(_hip.lib().d3pm_op_colsum_f32(_ptr(_f32(x, "x")), x.stride(0), x.shape[0], x.shape[1], _ptr(_f32(out, "out")), float(beta),
                                             _hip.stream_ptr()), "d3pm_op_colsum_f32")


def linear_bwd(x, w, dy, dw, db, dx=None, dx_beta=0.0):
    """y = x @ w.T + b:  dw += dy.T @ x, db += colsum(dy), dx = dx_beta * dx + dy @ w."""
    matmul(dy, x, dw, ta=True, beta=1.0)
    if db is not None:
        colsum(dy, db, 1.0)
    if dx is not None:
        matmul(dy, w, dx, beta=dx_beta)
    return dx


def act_bwd(u, dm, act=GELU):
    du = torch.empty_like(u)
    _hip.check(_hip.lib().d3pm_op_act_bwd_f32(_ptr(_f32(u, "u")), _ptr(_f32(dm, "dm")), _ptr(du), u.numel(), act, _hip.stream_ptr()),
               "d3pm_op_act_bwd_f32")
    return du


def mask_rows(x, mask):
    _hip.check(_hip.lib().d3pm_op_mask_rows_f32(_ptr(_f32(x, "x")), x.stride(0), x.shape[0], x.shape[1], _ptr(mask), mask.numel(),
                                                _hip.stream_ptr()), "d3pm_op_mask_rows_f32")


def layernorm_bwd(x, dout, w, b, dx, dw, db, film=None, dfilm=None, eps=1e-6, accumulate=True):
    M, d = x.shape
    _hip.check(_hip.lib().d3pm_op_layernorm_bwd_f32(_ptr(_f32(x, "x")), _ptr(_f32(dout, "dout")), _ptr(w), _ptr(b), _ptr(film), float(eps), M,
                                                    d, _ptr(_f32(dx, "dx")), 1 if accumulate else 0, _ptr(dw), _ptr(db), _ptr(dfilm),
                                                    _hip.stream_ptr()), "d3pm_op_layernorm_bwd_f32")


def attention_bwd(q, k, v, do, dq, dk, dv, n_heads, scale, beta_kv=0.0):
    """q [B,Tq,d], k / v [B,S,d] views (row strides free), do [B,Tq,d]; dq / dk / dv views of the same shapes."""
    B, Tq, d = q.shape
    S = k.shape[1]
    hd = d // n_heads
    stats = torch.empty(B * n_heads * Tq * 2, dtype=torch.float32, device=q.device)
    assert k.stride(1) == v.stride(1) and dk.stride(1) == dv.stride(1)
    _hip.check(_hip.lib().d3pm_op_attention_bwd_f32(_ptr(q), q.stride(1), _ptr(k), _ptr(v), k.stride(1), _ptr(do), do.stride(1), _ptr(dq),
                                                    dq.stride(1), _ptr(dk), _ptr(dv), dk.stride(1), _ptr(stats), B, Tq, S, n_heads, hd,
                                                    float(scale), float(beta_kv), _hip.stream_ptr()), "d3pm_op_attention_bwd_f32")


def ce_bwd(logits, targets, frame_mask, gscale):
    rows, K = logits.shape
    out = torch.empty_like(logits)
    _hip.check(_hip.lib().d3pm_op_ce_bwd_f32(_ptr(_f32(logits, "logits")), logits.stride(0), _ptr(targets), _ptr(frame_mask),
                                             frame_mask.numel(), rows, K, float(gscale), _ptr(out), out.stride(0), _hip.stream_ptr()),
               "d3pm_op_ce_bwd_f32")
    return out


def embed(tok, mask, table):
    rows, d = tok.numel(), table.shape[1]
    y = torch.empty((rows, d), dtype=torch.float32, device=table.device)
    _hip.check(_hip.lib().d3pm_op_embed_f32(_ptr(tok), _ptr(mask), 0 if mask is None else mask.numel(), _ptr(_f32(table, "table")), _ptr(y),
                                            rows, d, table.shape[0], _hip.stream_ptr()), "d3pm_op_embed_f32")
    return y


def embed_bwd(tok, mask, dy, dtable, padding_idx=0):
    _hip.check(_hip.lib().d3pm_op_embed_bwd_f32(_ptr(tok), _ptr(mask), 0 if mask is None else mask.numel(), _ptr(_f32(dy, "dy")),
                                                _ptr(_f32(dtable, "dtable")), tok.numel(), dy.shape[1], dtable.shape[0], padding_idx,
                                                _hip.stream_ptr()), "d3pm_op_embed_bwd_f32")


def _grad(p: torch.nn.Parameter) -> torch.Tensor:
    if p.grad is None:
        p.grad = torch.zeros_like(p, dtype=torch.float32)
    return p.grad


class D3PMTrainer:
    """Loss + gradients of AR.forward for a model in fp32 on a HIP device."""

    def __init__(self, model):
        if model.dtype != torch.float32 or model.device.type != "cuda":
            raise _hip.D3PMError("training runs in the F32 precision mode on a HIP device: model.float().to('cuda')")
        self.model = model

    # ---- condition encoders (once per utterance) with a stash ------------------------------------------------
    def _encode(self, which, tokens):
        """which 0: text int32 [S_t]; 1: prompt int32 [S_p, n_levels] -> (cond [S, d], stash)."""
        m, cfg = self.model, self.model.cfg
        d = cfg.d_model
        G = _hip.FAMILY_GENERIC
        lin = lambda x, w, b, **kw: _hip.op_linear(x, w, b, family=G, **kw)
        pe_text0, pe_prompt = m._pe()
        enc = m.encodertext if which == 0 else m.encoder2
        rows = tokens.shape[0]
        x = torch.empty((rows, d), dtype=torch.float32, device=m.device)
        table = m.text_emb.weight if which == 0 else m.proms_emb.weight
        pe = pe_text0 if which == 0 else pe_prompt
        _hip.check(_hip.lib().d3pm_op_cond_embed(_hip.F32, which, _ptr(tokens), cfg.n_levels, _ptr(table), _ptr(pe.contiguous()), _ptr(x), rows,
                                                 cfg.s_prompt, d, cfg.n_classes, _hip.stream_ptr()), "d3pm_op_cond_embed")
        H = cfg.cond_heads
        scale = (1.0 / (d // H)) ** 0.5
        layers = []
        for layer in enc[0].layers:
            s = {"x": x}
            s["qkv"] = lin(x, layer.self_attn.in_proj_weight, layer.self_attn.in_proj_bias)
            qkv = s["qkv"].view(1, rows, 3 * d)
            s["att"] = _hip.op_attention(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], H, scale, family=G).view(rows, d)
            s["t1"] = lin(s["att"], layer.self_attn.out_proj.weight, layer.self_attn.out_proj.bias, r1=x)
            s["x1"] = _hip.op_layernorm(s["t1"], layer.norm1.weight, layer.norm1.bias, eps=1e-5)
            s["ff"] = lin(s["x1"], layer.linear1.weight, layer.linear1.bias, act=RELU)
            s["t2"] = lin(s["ff"], layer.linear2.weight, layer.linear2.bias, r1=s["x1"])
            x = _hip.op_layernorm(s["t2"], layer.norm2.weight, layer.norm2.bias, eps=1e-5)
            layers.append(s)
        mlp = enc[1]
        top = {"x": x, "u": lin(x, mlp.fc1.weight, mlp.fc1.bias), "h": lin(x, mlp.fc1.weight, mlp.fc1.bias, act=SILU)}
        cond = lin(top["h"], mlp.fc2.weight, mlp.fc2.bias)
        return cond, {"which": which, "tokens": tokens, "layers": layers, "top": top}

    def _encode_backward(self, stash, dcond):
        m, cfg = self.model, self.model.cfg
        d, H = cfg.d_model, cfg.cond_heads
        scale = (1.0 / (d // H)) ** 0.5
        which, tokens, top = stash["which"], stash["tokens"], stash["top"]
        enc = m.encodertext if which == 0 else m.encoder2
        rows = tokens.shape[0]
        dev = dcond.device
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        mlp = enc[1]
        dh = new(rows, top["h"].shape[1])
        linear_bwd(top["h"], mlp.fc2.weight, dcond, _grad(mlp.fc2.weight), _grad(mlp.fc2.bias), dh)
        du = act_bwd(top["u"], dh, SILU)
        dx = new(rows, d)
        linear_bwd(top["x"], mlp.fc1.weight, du, _grad(mlp.fc1.weight), _grad(mlp.fc1.bias), dx)
        for layer, s in reversed(list(zip(enc[0].layers, stash["layers"]))):
            dt2 = new(rows, d)                                   # x = LN(t2; norm2), t2 = x1 + linear2(relu(linear1(x1)))
            layernorm_bwd(s["t2"], dx, layer.norm2.weight, layer.norm2.bias, dt2, _grad(layer.norm2.weight), _grad(layer.norm2.bias),
                          eps=1e-5, accumulate=False)
            dff = new(rows, s["ff"].shape[1])
            linear_bwd(s["ff"], layer.linear2.weight, dt2, _grad(layer.linear2.weight), _grad(layer.linear2.bias), dff)
            dpre = act_bwd(s["ff"], dff, RELU)                    # relu'(u) = [u > 0] = [relu(u) > 0]
            linear_bwd(s["x1"], layer.linear1.weight, dpre, _grad(layer.linear1.weight), _grad(layer.linear1.bias), dt2, dx_beta=1.0)
            dt1 = new(rows, d)                                   # x1 = LN(t1; norm1), t1 = x + out_proj(attention)
            layernorm_bwd(s["t1"], dt2, layer.norm1.weight, layer.norm1.bias, dt1, _grad(layer.norm1.weight), _grad(layer.norm1.bias),
                          eps=1e-5, accumulate=False)
            datt = new(rows, d)
            linear_bwd(s["att"], layer.self_attn.out_proj.weight, dt1, _grad(layer.self_attn.out_proj.weight),
                       _grad(layer.self_attn.out_proj.bias), datt)
            qkv, dqkv = s["qkv"].view(1, rows, 3 * d), new(1, rows, 3 * d)
            attention_bwd(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], datt.view(1, rows, d), dqkv[..., :d], dqkv[..., d:2 * d],
                          dqkv[..., 2 * d:], H, scale)
            dx = dt1                                             # residual branch; the in-projection adds its part
            linear_bwd(s["x"], layer.self_attn.in_proj_weight, dqkv.view(rows, 3 * d), _grad(layer.self_attn.in_proj_weight),
                       _grad(layer.self_attn.in_proj_bias), dx, dx_beta=1.0)
        if which == 0:
            embed_bwd(tokens, None, dx, _grad(m.text_emb.weight), padding_idx=0)
        else:
            gp = _grad(m.proms_emb.weight)
            for lvl in range(tokens.shape[1]):
                embed_bwd(tokens[:, lvl].contiguous(), None, dx, gp[lvl], padding_idx=-1)

    # ---- one denoiser evaluation with a stash ---------------------------------------------------------------
    def _forward(self, x_t, fm, t, kv_t, kv_p, film):
        m, cfg = self.model, self.model.cfg
        d, H = cfg.d_model, cfg.n_heads
        scale = (1.0 / (d // H)) ** 0.5
        G = _hip.FAMILY_GENERIC
        lin = lambda x, w, b, **kw: _hip.op_linear(x, w, b, family=G, **kw)
        x = embed(x_t.reshape(-1), fm, m.resps_emb.weight)
        stash = []
        for l, blk in enumerate(m.blocks):
            s = {"xin": x}
            s["h1"] = _hip.op_layernorm(x, blk.norm1.weight, blk.norm1.bias)
            s["qkv"] = lin(s["h1"], blk.attn.in_proj_weight, blk.attn.in_proj_bias)
            qkv = s["qkv"].view(1, -1, 3 * d)
            s["att"] = _hip.op_attention(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], H, scale, family=G).view(-1, d)
            s["x1"] = lin(s["att"], blk.attn.out_proj.weight, blk.attn.out_proj.bias, r1=x)
            s["h2"] = _hip.op_layernorm(s["x1"], blk.norm2.weight, blk.norm2.bias)
            s["h22"] = _hip.op_layernorm(s["x1"], blk.norm22.weight, blk.norm22.bias)
            wq, bq = blk.cross_attn.in_proj_weight[:d], blk.cross_attn.in_proj_bias[:d]
            s["qt"], s["qp"] = lin(s["h2"], wq, bq), lin(s["h22"], wq, bq)
            s["at"] = _hip.op_attention(s["qt"].view(1, -1, d), kv_t[l][..., :d], kv_t[l][..., d:], H, scale, family=G).view(-1, d)
            s["ap"] = _hip.op_attention(s["qp"].view(1, -1, d), kv_p[l][..., :d], kv_p[l][..., d:], H, scale, family=G).view(-1, d)
            wo, bo = blk.cross_attn.out_proj.weight, blk.cross_attn.out_proj.bias
            ot = lin(s["at"], wo, bo)
            s["x2"] = lin(s["ap"], wo, bo, r1=s["x1"], r2=ot)
            s["film"] = film[t, l].contiguous()
            s["h3"] = _hip.op_layernorm(s["x2"], blk.norm3.weight, blk.norm3.bias, film=s["film"])
            s["u"] = lin(s["h3"], blk.mlp.fc1.weight, blk.mlp.fc1.bias)
            s["g"] = lin(s["h3"], blk.mlp.fc1.weight, blk.mlp.fc1.bias, act=GELU)
            x = lin(s["g"], blk.mlp.fc2.weight, blk.mlp.fc2.bias, r1=s["x2"], row_mask=fm, mask_period=fm.numel())
            stash.append(s)
        logits = lin(x, m.final.weight, m.final.bias)
        return x, logits, stash

    def _backward(self, x_t, fm, t, kv_t, kv_p, cond_t, cond_p, dcond_t, dcond_p, x_last, logits, stash, targets, gscale):
        m, cfg = self.model, self.model.cfg
        d, H = cfg.d_model, cfg.n_heads
        scale = (1.0 / (d // H)) ** 0.5
        n = x_last.shape[0]
        dev = x_last.device
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        dlogits = ce_bwd(logits, targets, fm, gscale)
        dx = new(n, d)
        linear_bwd(x_last, m.final.weight, dlogits, _grad(m.final.weight), _grad(m.final.bias), dx)
        temb = m.time_emb.weight[t:t + 1]
        for l in reversed(range(len(m.blocks))):
            blk, s = m.blocks[l], stash[l]
            mask_rows(dx, fm)                                                     # x = (...) * mask at the end of the block
            # ---- x3 = x2 + fc2(gelu(fc1(LN3 + FiLM)))
            dg = new(n, 4 * d)
            linear_bwd(s["g"], blk.mlp.fc2.weight, dx, _grad(blk.mlp.fc2.weight), _grad(blk.mlp.fc2.bias), dg)
            du = act_bwd(s["u"], dg, GELU)
            dh3 = new(n, d)
            linear_bwd(s["h3"], blk.mlp.fc1.weight, du, _grad(blk.mlp.fc1.weight), _grad(blk.mlp.fc1.bias), dh3)
            dfilm = torch.zeros(2 * d, dtype=torch.float32, device=dev)
            layernorm_bwd(s["x2"], dh3, blk.norm3.weight, blk.norm3.bias, dx, _grad(blk.norm3.weight), _grad(blk.norm3.bias),
                          film=s["film"], dfilm=dfilm)
            # film = timestep_fc(time_emb[t])
            dfilm2 = dfilm.view(1, 2 * d)
            linear_bwd(temb, blk.timestep_fc.weight, dfilm2, _grad(blk.timestep_fc.weight), _grad(blk.timestep_fc.bias),
                       _grad(m.time_emb.weight)[t:t + 1], dx_beta=1.0)
            # ---- x2 = x1 + out(at) + out(ap), both through cross_attn.out_proj
            wo = blk.cross_attn.out_proj.weight
            gwo, gbo = _grad(wo), _grad(blk.cross_attn.out_proj.bias)
            dat, dap = new(n, d), new(n, d)
            linear_bwd(s["at"], wo, dx, gwo, gbo, dat)
            linear_bwd(s["ap"], wo, dx, gwo, gbo, dap)
            gin_w, gin_b = _grad(blk.cross_attn.in_proj_weight), _grad(blk.cross_attn.in_proj_bias)
            wq, wkv = blk.cross_attn.in_proj_weight[:d], blk.cross_attn.in_proj_weight[d:]
            for q_name, h_name, datt, kv, cond, dcond, norm in (("qt", "h2", dat, kv_t[l], cond_t, dcond_t, blk.norm2),
                                                               ("qp", "h22", dap, kv_p[l], cond_p, dcond_p, blk.norm22)):
                S = kv.shape[1]
                dq, dkv = new(1, n, d), new(1, S, 2 * d)
                attention_bwd(s[q_name].view(1, n, d), kv[..., :d], kv[..., d:], datt.view(1, n, d), dq, dkv[..., :d], dkv[..., d:], H, scale)
                dq2, dkv2 = dq.view(n, d), dkv.view(S, 2 * d)
                dh = new(n, d)
                linear_bwd(s[h_name], wq, dq2, gin_w[:d], gin_b[:d], dh)
                linear_bwd(cond, wkv, dkv2, gin_w[d:], gin_b[d:], dcond, dx_beta=1.0)
                layernorm_bwd(s["x1"], dh, norm.weight, norm.bias, dx, _grad(norm.weight), _grad(norm.bias))
            # ---- x1 = xin + attn.out_proj(self-attention(LN1))
            datt = new(n, d)
            linear_bwd(s["att"], blk.attn.out_proj.weight, dx, _grad(blk.attn.out_proj.weight), _grad(blk.attn.out_proj.bias), datt)
            qkv, dqkv = s["qkv"].view(1, n, 3 * d), new(1, n, 3 * d)
            attention_bwd(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], datt.view(1, n, d), dqkv[..., :d], dqkv[..., d:2 * d],
                          dqkv[..., 2 * d:], H, scale)
            dh1 = new(n, d)
            linear_bwd(s["h1"], blk.attn.in_proj_weight, dqkv.view(n, 3 * d), _grad(blk.attn.in_proj_weight), _grad(blk.attn.in_proj_bias), dh1)
            layernorm_bwd(s["xin"], dh1, blk.norm1.weight, blk.norm1.bias, dx, _grad(blk.norm1.weight), _grad(blk.norm1.bias))
            mask_rows(dx, fm)                                                     # x * mask at the start of the block
        embed_bwd(x_t.reshape(-1), fm, dx, _grad(m.resps_emb.weight), padding_idx=0)

    @torch.no_grad()
    def forward_backward(self, text_list: Sequence[torch.Tensor], proms_list: Sequence[torch.Tensor], resps_list: Sequence[torch.Tensor],
                         *, seed: int = 0, timesteps: Optional[int] = None):
        """The loss of AR.forward (mean over the utterances) and its gradient, accumulated into `param.grad`; eval-mode
        arithmetic (no dropout in the condition encoders: module docstring).
        Returns (loss fp32 scalar tensor, [(dcond_text, dcond_prompt)] per utterance)."""
        m, cfg = self.model, self.model.cfg
        smp = m.sampler()
        T = m.timesteps if timesteps is None else int(timesteps)
        B = len(text_list)
        losses, dconds = [], []
        with torch.cuda.device(m.device):
            for b, (text, prom, resps) in enumerate(zip(text_list, proms_list, resps_list)):
                r = resps.reshape(-1).to(m.device).long()[: cfg.canvas]
                x0 = F.pad(r, (0, cfg.canvas - r.shape[0])).to(torch.int32)[None].contiguous()
                fm = (x0[0] != 0).to(torch.uint8)
                n_live = int(fm.sum().item())
                targets = (x0[0] * fm.to(torch.int32)).contiguous()
                text_p, prom_p = m._padded_inputs([text], [prom])
                prom_p = prom_p[0].to(torch.int32)
                if prom_p.shape[-1] < cfg.n_levels:
                    prom_p = F.pad(prom_p, (0, cfg.n_levels - prom_p.shape[-1]), value=-1)
                cond_t2, st_t = self._encode(0, text_p[0].to(torch.int32).contiguous())
                cond_p2, st_p = self._encode(1, prom_p.contiguous())
                kv_t, kv_p = smp.cond_kv(cond_t2[None], cond_p2[None])
                dcond_t, dcond_p = torch.zeros_like(cond_t2), torch.zeros_like(cond_p2)
                gscale = 1.0 / (cfg.canvas * max(n_live, 1) * B)
                total = torch.zeros((), dtype=torch.float32, device=m.device)
                for t in range(1, T):
                    x_t = smp.q_sample(x0, fm, t, seed, b)
                    x_last, logits, stash = self._forward(x_t, fm, t, kv_t, kv_p, smp.film)
                    total += smp.ce_loss_rows(logits.view(1, cfg.canvas, cfg.n_classes), targets.view(1, -1), fm).mean()
                    self._backward(x_t, fm, t, kv_t, kv_p, cond_t2, cond_p2, dcond_t, dcond_p, x_last, logits, stash, targets, gscale)
                self._encode_backward(st_t, dcond_t)
                self._encode_backward(st_p, dcond_p)
                losses.append(total / max(n_live, 1))
                dconds.append((dcond_t, dcond_p))
        loss = torch.stack(losses).mean()
        m.loss = loss
        return loss, dconds


def all_reduce_gradients(model, *, bucket_bytes: int = 64 << 20, average: bool = True):
    """Data-parallel gradient reduction (the reference: DeepSpeed inside engine.backward / step, utils/engines.py:144-147):
    every parameter's .grad summed over the ranks of the default process group in buckets of <= `bucket_bytes` -- one flat
    buffer per bucket, one all-reduce each (backend "nccl" = RCCL over xGMI; ring all-reduce is per-link bound, so a few
    large messages instead of one per tensor) -- then divided by the world size.  A parameter that has a gradient on some
    rank but not on this one contributes zeros, so all ranks issue identical collectives; a parameter without a gradient
    on EVERY rank (upstream's dead `cross_attn2` / `token_emb` tensors) is left out and keeps `.grad = None`, exactly as at
    world size 1 -- otherwise weight decay would touch it in data-parallel runs only (one small MAX all-reduce of the
    has-gradient bitmask decides that)."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    world = dist.get_world_size()
    params = [p for p in model.parameters() if p.requires_grad]
    if not params:
        return 0
    has = torch.tensor([1 if p.grad is not None else 0 for p in params], dtype=torch.int32, device=params[0].device)
    dist.all_reduce(has, op=dist.ReduceOp.MAX)
    params = [p for p, h in zip(params, has.tolist()) if h]
    n_coll, i = 1, 0
    while i < len(params):
        bucket, size = [], 0
        while i < len(params) and (not bucket or size + params[i].numel() * 4 <= bucket_bytes):
            bucket.append(params[i])
            size += params[i].numel() * 4
            i += 1
        flat = torch.cat([(_grad(p) if p.grad is not None else torch.zeros_like(p, dtype=torch.float32)).reshape(-1).float()
                          for p in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if average:
            flat /= world
        off = 0
        for p in bucket:
            k = p.numel()
            if p.grad is None:
                p.grad = torch.zeros_like(p, dtype=torch.float32)
            p.grad.copy_(flat[off:off + k].view_as(p))
            off += k
        n_coll += 1
    return n_coll
