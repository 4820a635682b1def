"""Drop-in `NAR`: the stock non-autoregressive VALL-E model that fills quantizer levels 1..7 once the D3PM sampler
has produced level 0 (the step right after the hot path: /root/reference/vall_e/__main__.py:36-38,
SURVEY.md §8f row 1).

Same surface as the reference's /root/reference/vall_e/vall_e/nar.py `class NAR(Base)`: ctor
`NAR(n_tokens, d_model=512, n_heads=8, n_layers=12, p_dropout=0.1)` (base.py:316-323), state-dict key layout
(base.py:336-360), `forward(text_list, proms_list, resps_list, sampling_temperature=0.2) -> [LongTensor[t, 8]]`.
Every level runs in HIP behind `d3pm_nar_level`; PyTorch stores the weights.  The training branch
(nar.py:53-74) is out of scope.  Sampling uses a Philox Gumbel-max instead of torch's multinomial stream (`seed=`).
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import torch
from torch import Tensor, nn

from . import _hip
from .ar_discrete import SymmapState
from .synth import NARConfig


def _sinusoid_table(n: int, d_model: int, dtype: torch.dtype) -> Tensor:
    """base.py:38-89: omega computed in fp32, then it follows the module dtype (.half() rounds it)."""
    half = d_model // 2
    omega = torch.exp(-math.log(1e4) * (torch.arange(half, dtype=torch.float32) / half)).to(dtype)
    ang = omega[None, :] * torch.arange(n)[:, None]
    return torch.cat([ang.sin(), ang.cos()], dim=-1)


class _Table(nn.Module):
    def __init__(self, *shape):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(*shape))


class _AdaLNParams(nn.Module):
    def __init__(self, d, n_levels):
        super().__init__()
        self.emb = nn.Embedding(n_levels, 2 * d)
        nn.init.zeros_(self.emb.weight)


class _AttnParams(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.to_qkv = nn.Linear(d, 3 * d, bias=False)
        self.to_out = nn.Linear(d, d)


class _Prenorm(nn.Module):
    def __init__(self, block, d, n_levels):
        super().__init__()
        self.block = block
        self.norm = _AdaLNParams(d, n_levels)


class _BlockParams(nn.Module):
    """Parameter container with upstream's names (base.py:197-222); never called."""

    def __init__(self, d, n_levels):
        super().__init__()
        self.attn = _Prenorm(_AttnParams(d), d, n_levels)
        self.ffn = _Prenorm(nn.Sequential(nn.Linear(d, 4 * d), nn.GELU(), nn.Dropout(0.0), nn.Linear(4 * d, d)), d, n_levels)


class NAR(SymmapState, nn.Module):
    n_resp_levels = 7
    n_prom_levels = 8
    pad_rows_to_tiles = True      # _pack: round the padded grid up so that batch * t_max is a multiple of 192 (big-tile GEMMs)

    def __init__(self, n_tokens: int = 1024, d_model: int = 512, n_heads: int = 8, n_layers: int = 12, p_dropout: float = 0.1):
        super().__init__()
        self.cfg = NARConfig(d_model=d_model, n_heads=n_heads, n_layers=n_layers, n_tokens=n_tokens)
        self.n_tokens = n_tokens
        self.text_emb = _Table(n_tokens, d_model)
        self.proms_emb = _Table(self.n_prom_levels, n_tokens, d_model)
        self.resps_emb = _Table(self.n_resp_levels, n_tokens, d_model)
        self.sep = nn.Parameter(torch.randn(d_model))
        self.blocks = nn.ModuleList([_BlockParams(d_model, self.n_resp_levels) for _ in range(n_layers)])
        self.classifier = nn.Linear(d_model, n_tokens)
        self._runner = None
        self._runner_key = None
        self._init_symmaps()

    @property
    def dtype(self):
        return self.classifier.weight.dtype

    @property
    def device(self):
        return self.classifier.weight.device

    def runner(self, t_max: int) -> _hip.NarRunner:
        if self.device.type != "cuda":
            raise RuntimeError("the NAR levels run on MI355X only: move the model to a HIP device; there is no CPU path")
        sd = dict(self.named_parameters())
        key = (self.dtype, self.device, tuple((k, v.data_ptr(), v._version) for k, v in sd.items()))
        if self._runner is None or self._runner_key != key or self._runner.weights.pe_rows < t_max:
            rows = max(2048, t_max)
            pe = _sinusoid_table(rows, self.cfg.d_model, self.dtype).to(self.device).contiguous()
            with torch.cuda.device(self.device):
                self._runner = _hip.NarRunner(self.cfg, {k: v.detach() for k, v in sd.items()}, self.dtype, self.device, pe)
            self._runner_key = key
        return self._runner

    def _pack(self, text_list, proms_list, resps_list):
        """Ragged lists -> padded int32 grids on the model's device.  The lengths come from the tensors' shapes (host
        integers): nothing here reads device memory back, so a batch that is already on the GPU (the D3PM stage's output)
        is packed without a single synchronisation."""
        import torch.nn.functional as F
        from torch.nn.utils.rnn import pad_sequence
        dev = self.device
        lens_host = [(len(t), len(p), len(r)) for t, p, r in zip(text_list, proms_list, resps_list)]
        i32 = lambda x: x.to(device=dev, dtype=torch.int32, non_blocking=True)
        text = pad_sequence([i32(t) for t in text_list], batch_first=True)
        # absent prompt levels are -1 (they contribute nothing), padded prompt rows are 0 like upstream's zero padding
        prom = pad_sequence([F.pad(i32(p), (0, self.n_prom_levels - p.shape[-1]), value=-1) for p in proms_list], batch_first=True)
        resp = pad_sequence([F.pad(i32(r), (0, self.n_resp_levels + 1 - r.shape[-1])) for r in resps_list], batch_first=True)
        lens = torch.tensor(lens_host, dtype=torch.int32).to(dev, non_blocking=True)
        t_max = max(a + b_ + c for a, b_, c in lens_host) + 2
        if self.pad_rows_to_tiles:
            # t_max is the padded grid's row count per utterance (rows past an utterance's own length are masked padding): a few more
            # of them make batch * t_max a multiple of 192, so the projections run as big-tile GEMMs (192 x 128) instead of
            # one 128 x 128 tile per workgroup -- same bits (the schedules accumulate in the same order)
            step = 192 // math.gcd(len(lens_host), 192)
            padded = -(-t_max // step) * step
            if padded * 100 <= t_max * 103:
                t_max = padded
        return lens, text.contiguous(), prom.contiguous(), resp.contiguous(), t_max, lens_host

    @torch.no_grad()
    def forward(self, text_list: Sequence[Tensor], proms_list: Sequence[Tensor], resps_list: Sequence[Tensor],
                sampling_temperature: float = 0.2, *, seed: Optional[int] = None, greedy: bool = False, utt0: int = 0,
                return_logits_level: Optional[int] = None):
        """resps_list: [t, l] with the levels already known (l = 1 after the D3PM stage) -> [LongTensor[t, 8]]."""
        levels = {r.shape[-1] for r in resps_list}
        if len(levels) > 1:
            raise ValueError(f"Please give only one level, got {levels}.")
        n_given = next(iter(levels))
        if n_given == self.n_resp_levels + 1:
            raise NotImplementedError("the training branch of NAR.forward (nar.py:53-74) is outside this build's scope")
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        lens, text, prom, resp, t_max, lens_host = self._pack(text_list, proms_list, resps_list)
        run = self.runner(t_max)
        flags = _hip.FLAG_GREEDY if greedy else 0
        logits = None
        with torch.cuda.device(self.device):
            for level in range(n_given - 1, self.n_resp_levels):
                lg = run.level(lens, text, prom, resp, t_max, level, sampling_temperature, seed, utt0, flags,
                               want_logits=(return_logits_level == level))
                if lg is not None:
                    logits = lg
        out = [resp[b, : lens_host[b][2]].long() for b in range(len(text_list))]
        return (out, logits, lens, t_max) if return_logits_level is not None else out
