"""Drop-in `AR` model of the discrete-diffusion (D3PM, absorbing state) codec-token sampler.

Same public surface as the reference's /root/reference/vall_e/vall_e/ar_discrete.py `class AR`
(ctor signature :205, `generate_audio` :696, `p_sample` :401, `q_sample` :467, `timesteps`,
state-dict key layout :210-240) but the reverse process runs in hand-written HIP kernels for
gfx950 behind the C ABI of include/d3pm_hip.h.  PyTorch only stores the weights, runs the two
small once-per-utterance condition encoders (:216-230, 0.3 % of the reference's time) and provides
the HIP stream.  There is no CPU or eager fallback for the diffusion loop.

Differences from upstream, all opt-in or strictly more general:
  * the ctor honours its arguments (upstream overrides them with d=32,H=16,L=8,steps=100, :207-238);
    `AR.reference_native()` builds exactly the upstream shape;
  * batches: upstream only works for one utterance (:699); here B utterances are B independent
    runs (per-utterance Philox noise stream), returned as [B, canvas];
  * noise comes from a counter-based Philox stream (`seed=`) instead of the CPU Mersenne generator;
  * the 630 MB of dense transition tables are replaced by their 4 fp16 scalars per step.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from . import _hip
from .synth import MASK_ID, N_CLASSES, D3PMConfig


class _Mlp(nn.Module):
    """timm-style Mlp (fc1 -> act -> fc2); state-dict keys fc1.*, fc2.* as upstream's timm import."""

    def __init__(self, d_in, d_hidden, d_out, act):
        super().__init__()
        self.fc1 = nn.Linear(d_in, d_hidden)
        self.act = act
        self.fc2 = nn.Linear(d_hidden, d_out)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class _LevelSumEmbedding(nn.Module):
    """Prompt embedding: sum over quantizer levels of per-level tables (base.py:244-274)."""

    def __init__(self, n_levels, n_tokens, d):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(n_levels, n_tokens, d))

    def forward(self, codes: Tensor) -> Tensor:
        """codes int64 [..., S, l<=n_levels] -> [..., S, d]; fp32 sum, one rounding (the one-hot
        contraction upstream accumulates the <= 8 non-zero terms in fp32)."""
        w = self.weight
        out = torch.zeros(codes.shape[:-1] + (w.shape[-1],), dtype=torch.float32, device=w.device)
        for lvl in range(codes.shape[-1]):
            out += F.embedding(codes[..., lvl], w[lvl]).float()
        return out.to(w.dtype)


class _DiTBlockParams(nn.Module):
    """Parameter container with upstream's names (ar_discrete.py:103-124).  Never called: the
    block forward is the HIP path.  cross_attn2 is kept so upstream state dicts load strictly."""

    def __init__(self, d, heads):
        super().__init__()
        self.norm1 = nn.LayerNorm(d, eps=1e-6)
        self.attn = nn.MultiheadAttention(d, heads)
        self.norm2 = nn.LayerNorm(d, eps=1e-6)
        self.cross_attn = nn.MultiheadAttention(d, heads)
        self.norm22 = nn.LayerNorm(d, eps=1e-6)
        self.cross_attn2 = nn.MultiheadAttention(d, heads)
        self.norm3 = nn.LayerNorm(d, eps=1e-6)
        self.mlp = _Mlp(d, 4 * d, d, nn.GELU())
        self.timestep_fc = nn.Linear(d, 2 * d)

    def forward(self, *a, **k):
        raise RuntimeError("DiT blocks execute inside libd3pm_hip.so; call AR.generate_audio")


def _sinusoid_table(n: int, d_model: int, dtype: torch.dtype) -> Tensor:
    """[n, d] host table [sin | cos] as upstream's SinusodialEmbedding yields it (ar_discrete.py:41-72):
    omega is *computed* in fp16, then follows the module dtype (`.half()` keeps it, `.float()` widens
    the fp16 values) and the angles / sin / cos are evaluated in that dtype."""
    half = d_model // 2
    omega = torch.exp(-math.log(1e4) * (torch.arange(half, dtype=torch.float16) / half)).to(dtype)
    ang = omega[None, :] * torch.arange(n)[:, None]
    return torch.cat([ang.sin(), ang.cos()], dim=-1)


class SymmapState:
    """`phone_symmap` / `spkr_symmap` as the reference's export attaches them to the trained module
    (/root/reference/vall_e/export.py:18-19; read back as `ar.phone_symmap` at /root/reference/vall_e/__main__.py:56).
    Upstream ships them inside a whole-module pickle; here they ride in the state dict under one extra key,
    `_symmaps`, present only when a map is set -- so a state dict exported upstream (no such key) loads strictly,
    and a state dict saved here loads upstream after `sd.pop("_symmaps", None)`."""
    SYMMAP_KEY = "_symmaps"

    def _init_symmaps(self):
        self.phone_symmap: dict = {}      # {phone symbol: id >= 1}, data.py:125-127
        self.spkr_symmap: dict = {}       # {speaker name: id >= 0}, data.py:133-134

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        sd = super().state_dict(*args, destination=destination, prefix=prefix, keep_vars=keep_vars)
        if self.phone_symmap or self.spkr_symmap:
            sd[prefix + self.SYMMAP_KEY] = {"phone_symmap": dict(self.phone_symmap), "spkr_symmap": dict(self.spkr_symmap)}
        return sd

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        state_dict = dict(state_dict)
        maps = state_dict.pop(self.SYMMAP_KEY, None)
        out = super().load_state_dict(state_dict, strict=strict, assign=assign)
        if maps is not None:
            self.phone_symmap = dict(maps.get("phone_symmap", {}))
            self.spkr_symmap = dict(maps.get("spkr_symmap", {}))
        return out

    @classmethod
    def load_exported(cls, path, **ctor):
        """A checkpoint written by tools/convert_upstream_pickle.py (run once in the upstream environment on the
        whole-module pickle of /root/reference/vall_e/export.py:20): {"state_dict", "phone_symmap", "spkr_symmap"}."""
        blob = torch.load(path, map_location="cpu")
        model = cls(**ctor) if ctor else cls.reference_native() if hasattr(cls, "reference_native") else cls()
        model.load_state_dict(blob["state_dict"])
        model.phone_symmap = dict(blob.get("phone_symmap") or {})
        model.spkr_symmap = dict(blob.get("spkr_symmap") or {})
        return model


class AR(SymmapState, nn.Module):
    n_resp_levels = 1
    num_classes = N_CLASSES

    def __init__(self, d_model=512, n_steps=100, n_tokens=1024, max_n_levels=8, n_heads=8, num_layers=6, *,
                 canvas: int = 448, n_frames: int = 350, s_text: int = 50, s_prompt: int = 398, n_q: int = 1):
        """Positional arguments as upstream (ar_discrete.py:205).  `n_q` > 1 is this build's extension (SURVEY.md section 8d
        config 2, BASELINE.json configs[1] "x 8 quantizers"; upstream generates level 0 only and leaves levels 1..7 to the NAR
        model): the D3PM then denoises all n_q quantizer levels of a frame jointly -- token grids [B, canvas, n_q], a frame's
        input = the sum of its level embeddings (`resps_emb.weight` [n_q, K, d]), `final` has n_q * K outputs, and every
        (frame, level) is sampled like a level-0 token.  n_q = 1 is the upstream model, bit for bit."""
        super().__init__()
        if n_tokens + 1 != N_CLASSES:
            raise ValueError("the absorbing-state tables assume 1024 codec ids + 1 mask id")
        if not 1 <= n_q <= 16:
            raise ValueError("n_q must be in 1..16")
        self.cfg = D3PMConfig(d_model=d_model, n_heads=n_heads, n_layers=num_layers, canvas=canvas, n_frames=n_frames,
                              s_text=s_text, s_prompt=s_prompt, timesteps=n_steps, n_levels=max_n_levels, n_q=n_q)
        self.n_resp_levels = n_q
        cfg, d = self.cfg, d_model
        self.timesteps = n_steps                       # read at call time, like upstream (:750)
        self.text_emb = nn.Embedding(N_CLASSES, d, padding_idx=0)
        self.proms_emb = _LevelSumEmbedding(max_n_levels, N_CLASSES, d)
        self.resps_emb = nn.Embedding(N_CLASSES, d, padding_idx=0) if n_q == 1 else _LevelSumEmbedding(n_q, N_CLASSES, d)
        self.time_emb = nn.Embedding(n_steps + 1, d)
        self.token_emb = nn.Embedding(N_CLASSES, d)    # unused upstream too; kept for state-dict parity

        def encoder(mult):
            layer = nn.TransformerEncoderLayer(d_model=d, nhead=cfg.cond_heads, dim_feedforward=cfg.cond_ff, dropout=0.0)
            return nn.Sequential(nn.TransformerEncoder(layer, num_layers=cfg.cond_layers, enable_nested_tensor=False),
                                 _Mlp(d, d * mult, d, nn.SiLU()))

        self.encodertext = encoder(2)
        self.encoder2 = encoder(3)
        self.blocks = nn.ModuleList([_DiTBlockParams(d, n_heads) for _ in range(num_layers)])
        self.final = nn.Linear(d, n_q * N_CLASSES)
        self._pe_cache = {}
        self.eps = 1.0e-6
        self._sampler = None
        self._sampler_key = None
        self.loop_streams = 1     # >1: the batch is cut into that many independent chunks on separate HIP streams
        self._streams = []
        self._init_symmaps()

    # ------------------------------------------------------------------ construction helpers
    @classmethod
    def reference_native(cls) -> "AR":
        """The only shape upstream's class can build: d=32, 16 heads, 8 blocks, 100 steps."""
        return cls(d_model=32, n_steps=100, n_tokens=1024, max_n_levels=8, n_heads=16, num_layers=8)

    @classmethod
    def from_config(cls, cfg: D3PMConfig) -> "AR":
        return cls(cfg.d_model, cfg.timesteps, cfg.n_classes - 1, cfg.n_levels, cfg.n_heads, cfg.n_layers,
                   canvas=cfg.canvas, n_frames=cfg.n_frames, s_text=cfg.s_text, s_prompt=cfg.s_prompt, n_q=cfg.n_q)

    @property
    def dtype(self) -> torch.dtype:
        return self.final.weight.dtype

    @property
    def device(self) -> torch.device:
        return self.final.weight.device

    def _pe(self):
        """(pe_text0 [1,d], pe_prompt [S_p,d]) in the model dtype on the model device, built on the host."""
        key = (self.dtype, self.device)
        if key not in self._pe_cache:
            self._pe_cache[key] = (_sinusoid_table(1, self.cfg.d_model, self.dtype).to(self.device),
                                   _sinusoid_table(self.cfg.s_prompt, self.cfg.d_model, self.dtype).to(self.device))
        return self._pe_cache[key]

    # ------------------------------------------------------------------ HIP sampler plumbing
    def sampler(self) -> _hip.Sampler:
        """(Re)binds the C-ABI pointer tables when weights moved or changed dtype."""
        if self.device.type != "cuda":
            raise RuntimeError("the D3PM sampler runs on MI355X only: move the model to a HIP device "
                               "(model.to('cuda')); there is no CPU path")
        sd = {k: v for k, v in self.named_parameters()}
        key = (self.dtype, self.device, tuple((k, v.data_ptr(), v._version) for k, v in sd.items()))
        if self._sampler is None or self._sampler_key != key:
            with torch.cuda.device(self.device):
                pe_text0, pe_prompt = self._pe()
                self._sampler = _hip.Sampler(self.cfg, {k: v.detach() for k, v in sd.items()}, self.dtype, self.device,
                                             pe_text0.contiguous(), pe_prompt.contiguous())
            self._sampler_key = key
        return self._sampler

    # ------------------------------------------------------------------ conditioning (torch-ROCm)
    @staticmethod
    def _pad_rows(x: Tensor, n: int) -> Tensor:
        if x.shape[0] >= n:
            return x[:n]
        return F.pad(x, [0, 0] * (x.dim() - 1) + [0, n - x.shape[0]])

    def _padded_inputs(self, text_list, proms_list):
        cfg, dev = self.cfg, self.device
        text = torch.stack([self._pad_rows(t.to(dev).long(), cfg.s_text) for t in text_list])           # [B,S_t]
        prom = torch.stack([self._pad_rows(p.to(dev).long(), cfg.s_prompt) for p in proms_list])        # [B,S_p,l]
        return text, prom

    def encode_conditions(self, text_list: Sequence[Tensor], proms_list: Sequence[Tensor]):
        """-> (cond_text [B,S_t,d], cond_prompt [B,S_p,d]) through the HIP condition encoders
        (d3pm_encode_conditions).  Same statements as upstream (:711-746): zero pad / truncate, embed, text gets
        PE(position 0) on every phoneme (the x.shape[0] quirk at :89), the prompt true positions; two post-norm
        encoder layers + Mlp.  Prompts with fewer than n_levels quantizer levels: the missing levels add nothing."""
        text, prom = self._padded_inputs(text_list, proms_list)
        if prom.shape[-1] < self.cfg.n_levels:
            prom = F.pad(prom, (0, self.cfg.n_levels - prom.shape[-1]), value=-1)
        with torch.cuda.device(self.device):
            return self.sampler().encode_conditions(text, prom)

    def encode_conditions_torch(self, text_list: Sequence[Tensor], proms_list: Sequence[Tensor]):
        """The same encoders on PyTorch-ROCm modules (tests cross-check the HIP path against it)."""
        text, prom = self._padded_inputs(text_list, proms_list)
        pe_text0, pe_prompt = self._pe()
        ct = self.text_emb(text) + pe_text0
        cp = self.proms_emb(prom) + pe_prompt
        # sequence-first batches: per-utterance arithmetic is what upstream's unbatched call does
        ct = self.encodertext(ct.transpose(0, 1)).transpose(0, 1)
        cp = self.encoder2(cp.transpose(0, 1)).transpose(0, 1)
        return ct.contiguous(), cp.contiguous()

    def canvas_init(self, batch: int, n_frames: Optional[int] = None):
        """x_T: `n_frames` mask ids then zeros; the frame mask is fixed for the whole loop (:699-709)."""
        cfg = self.cfg
        n_frames = cfg.n_frames if n_frames is None else n_frames
        if not 0 < n_frames <= cfg.canvas:
            raise ValueError(f"n_frames must be in 1..{cfg.canvas}")
        shape = (batch, cfg.canvas) if cfg.n_q == 1 else (batch, cfg.canvas, cfg.n_q)
        x = torch.zeros(shape, dtype=torch.int32, device=self.device)
        x[:, :n_frames] = MASK_ID
        frame_mask = (x[0].reshape(cfg.canvas, -1)[:, 0] != 0).to(torch.uint8)
        return x, frame_mask

    # ------------------------------------------------------------------ the hot path
    @torch.no_grad()
    def generate_audio(self, text_list, proms_list, resps_list=None, *, steps: Optional[int] = None,
                       n_frames: Optional[int] = None, seed: Optional[int] = None, greedy: bool = False,
                       utt0: int = 0, return_trace: bool = False, flags: int = 0, streams: Optional[int] = None,
                       graph: Optional[bool] = None, fp8: bool = False, global_batch: Optional[int] = None):
        """Reverse diffusion for len(text_list) utterances.  Positional behaviour as upstream:
        one utterance -> int64 [canvas] (squeezed, untrimmed; rows >= n_frames are sampled from
        final.bias and meaningless); with n_q > 1 (constructor) [canvas, n_q] / [B, canvas, n_q].  `resps_list` is ignored, as
        upstream ignores it (:699).
        `fp8=True` is the fast configuration of BASELINE.json configs[4]: the QKV, cross-attention query, fc1 and fc2
        projections run on the block-scaled fp8 matrix instruction (e4m3 codes, one power-of-two scale per 32 elements;
        d_model = 512, 16-bit model, batch * canvas a multiple of 192); the reference has no such mode.
        `global_batch`: the size of the logical batch these utterances are a shard of (vall_e/vall_e/dp.py passes it; the stream
        chunks below do too).  The attention kernels come in two instruction shapes that are picked by batch size and accumulate
        in different orders; with the global batch given, a shard takes the kernels of the unsplit batch, so the ids of an
        utterance do not depend on how the batch was split (d3pm_tuning.regime_batch).
        `graph=True` replays the loop from a captured HIP graph (seed read from HBM, identical results).  Off by
        default: measured on MI355X one utterance takes 66.6 ms replayed and 66.3 ms launched eagerly -- the ~5000
        kernels of a reverse process are bound by their own ~10 us latency at M = 768 rows, not by launch overhead."""
        if len(text_list) != len(proms_list) or len(text_list) == 0:
            raise ValueError("text_list and proms_list must be non-empty and of equal length")
        B = len(text_list)
        smp = self.sampler()
        t_start = (self.timesteps - 1) if steps is None else steps
        if not 0 < t_start < smp.schedule.timesteps:
            raise ValueError(f"steps must be in 1..{smp.schedule.timesteps - 1}")
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())       # follows torch.manual_seed
        n_streams = max(1, min(B, self.loop_streams if streams is None else streams))
        regime = max(int(global_batch), B) if global_batch else (B if n_streams > 1 else 0)
        with torch.cuda.device(self.device), _hip.tuning(regime_batch=regime):
            cond_text, cond_prompt = self.encode_conditions(text_list, proms_list)
            x, frame_mask = self.canvas_init(B, n_frames)
            fl = flags | (_hip.FLAG_GREEDY if greedy else 0)
            use_graph = bool(graph)
            if fp8 and (use_graph or (n_streams > 1 and not return_trace)):
                raise ValueError("fp8=True runs on the single-stream eager loop only (no graph replay, no stream chunks)")
            use_graph = use_graph and not return_trace and not _hip.profiling()
            if use_graph:
                kv_t, kv_p = smp.cond_kv(cond_text, cond_prompt)
                trace = None
                smp.sample_loop_graphed(x, frame_mask, t_start, 0, kv_t, kv_p, seed, utt0, fl)
            elif n_streams == 1 or return_trace:
                kv_t, kv_p = smp.cond_kv(cond_text, cond_prompt)
                trace = smp.sample_loop(x, frame_mask, t_start, 0, kv_t, kv_p, seed, utt0, fl, trace=return_trace, fp8=fp8)
            else:
                # utterances are independent: chunks of the batch run the whole loop on their own stream so that
                # the short kernels of one chunk fill the ramp-up / epilogue bubbles of the others
                trace = None
                while len(self._streams) < n_streams:
                    self._streams.append(torch.cuda.Stream(device=self.device))
                cur = torch.cuda.current_stream()
                bounds = [(B * i) // n_streams for i in range(n_streams + 1)]
                for i in range(n_streams):
                    lo, hi = bounds[i], bounds[i + 1]
                    st = self._streams[i]
                    st.wait_stream(cur)
                    with torch.cuda.stream(st):
                        kv_t, kv_p = smp.cond_kv(cond_text[lo:hi], cond_prompt[lo:hi])
                        smp.sample_loop(x[lo:hi], frame_mask, t_start, 0, kv_t, kv_p, seed, utt0 + lo, fl, slot=i)
                        for t_ in (kv_t, kv_p, cond_text, cond_prompt, x):
                            t_.record_stream(st)
                for i in range(n_streams):
                    cur.wait_stream(self._streams[i])
        out = x.long()
        out = out[0] if B == 1 else out
        return (out, trace) if return_trace else out

    # ------------------------------------------------------------------ upstream method names
    @torch.no_grad()
    def p_sample(self, model_logits: Tensor, t: Tensor, x: Tensor, *, seed: int = 0, utt0: int = 0):
        """One reverse transition from x0-logits [B,T,K] at step t[0] (ar_discrete.py:401-420).
        Returns (sample int64 [B,T], softmax(logits)) like upstream."""
        smp = self.sampler()
        x_next, _ = smp.posterior_sample(model_logits, x.to(torch.int32).contiguous(), int(t.reshape(-1)[0]), seed, utt0)
        return x_next.long(), F.softmax(model_logits, dim=-1)

    @torch.no_grad()
    def q_sample(self, x_start: Tensor, t: Tensor, mask: Tensor, *, seed: int = 0, utt0: int = 0):
        """Forward noising q(x_t | x_0) (ar_discrete.py:467-487)."""
        smp = self.sampler()
        fm = mask.to(torch.uint8).contiguous()
        return smp.q_sample(x_start.to(torch.int32).contiguous(), fm, int(t.reshape(-1)[0]), seed, utt0).long()

    def forward_backward(self, text_list, proms_list, resps_list, *, seed: Optional[int] = None, timesteps: Optional[int] = None):
        """The training step's compute (reference: `engine.backward(engine(...))`, utils/engines.py:144-147 over
        ar_discrete.py:588-694): the loss of `forward` AND its gradient for every parameter the forward reads, accumulated
        into `param.grad` by the HIP backward kernels (vall_e/vall_e/train.py; fp32 model).  Follow it with
        `train.all_reduce_gradients(self)` under torch.distributed and any torch.optim step.  Returns the loss.
        Eval-mode arithmetic: the dropout the reference applies inside its condition encoders in train mode (p = 0.1 /
        0.01, ar_discrete.py:216-230) is omitted (vall_e/vall_e/train.py)."""
        from .train import D3PMTrainer
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        loss, _ = D3PMTrainer(self).forward_backward(text_list, proms_list, resps_list, seed=seed, timesteps=timesteps)
        return loss

    @torch.no_grad()
    def forward(self, text_list, proms_list, resps_list=None, spkr_name=None, *, seed: Optional[int] = None):
        """Training-side forward, evaluation only (forward half; ar_discrete.py:588-694): for every
        utterance, x_0 = the target codes zero-padded / truncated to the canvas, mask = (x_0 != 0), and
            loss = sum_{t=1}^{timesteps-1} mean_canvas CE(final(blocks(q_sample(x_0, t))) * mask, x_0 * mask) / mask.sum().
        Sets `self.loss` (mean over the utterances; upstream indexes `[0]` throughout, so for one utterance this is its
        value) and returns the masked logits of the last step of the last utterance, `[canvas, n_classes]`, as upstream.
        q_sample draws Philox stream 1 keyed by `seed` instead of torch.rand.  No autograd graph is built: gradients come from
        `forward_backward` (hand-written backward kernels), not from torch.autograd."""
        if resps_list is None or not (len(text_list) == len(proms_list) == len(resps_list)) or len(text_list) == 0:
            raise ValueError("text_list, proms_list and resps_list must be non-empty and of equal length")
        smp = self.sampler()
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        losses, last = [], None
        with torch.cuda.device(self.device):
            for b, (text, prom, resps) in enumerate(zip(text_list, proms_list, resps_list)):
                r = resps.reshape(-1).to(self.device).long()[: self.cfg.canvas]
                x0 = F.pad(r, (0, self.cfg.canvas - r.shape[0])).to(torch.int32)[None].contiguous()
                frame_mask = (x0[0] != 0).to(torch.uint8)
                cond_text, cond_prompt = self.encode_conditions([text], [prom])
                kv_t, kv_p = smp.cond_kv(cond_text, cond_prompt)
                loss, logits = smp.training_forward(x0, frame_mask, kv_t, kv_p, seed, utt0=b, timesteps=self.timesteps)
                losses.append(loss[0])
                last = logits[0]
        self.loss = torch.stack(losses).mean()
        return last
