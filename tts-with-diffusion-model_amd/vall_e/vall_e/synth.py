"""Shapes of the D3PM model and seed-deterministic synthetic weights / inputs.

The state-dict key layout is the reference's (probed from
/root/reference/vall_e/vall_e/ar_discrete.py:210-240): loading the dict produced here into the
reference model with `load_state_dict` works 1:1, so parity fixtures need no committed weight blobs.
Weights are drawn with numpy PCG64 in sorted-key order with PyTorch-default-like scales
(there is no network for checkpoints; BASELINE.md §3 prescribes random-init weights).
"""
from __future__ import annotations

import dataclasses
import math

import numpy as np
import torch

N_CLASSES = 1025            # 1024 EnCodec ids + the absorbing id (ar_discrete.py:255)
MASK_ID = N_CLASSES // 2    # absorbing state 512 (ar_discrete.py:332,699)


@dataclasses.dataclass(frozen=True)
class D3PMConfig:
    """Dimensions of one denoiser. `native()` is the only shape the reference itself can build."""
    d_model: int = 32
    n_heads: int = 16
    n_layers: int = 8
    canvas: int = 448          # T: frames the denoiser sees (ar_discrete.py:704-707)
    n_frames: int = 350        # live frames initialised to MASK_ID (ar_discrete.py:699)
    s_text: int = 50           # phoneme keys (ar_discrete.py:714)
    s_prompt: int = 398        # acoustic-prompt keys (ar_discrete.py:726)
    timesteps: int = 100       # loop runs t = timesteps-1 .. 1 (ar_discrete.py:207,750)
    n_levels: int = 8          # prompt quantizer levels summed by MultiEmbedding (base.py:244)
    cond_heads: int = 16       # TransformerEncoderLayer(nhead=16) (ar_discrete.py:218,226)
    cond_ff: int = 2048        # TransformerEncoderLayer default dim_feedforward
    cond_layers: int = 2
    n_classes: int = N_CLASSES
    mask_id: int = MASK_ID
    n_q: int = 1               # quantizer levels the D3PM generates jointly; 1 = upstream (level 0 only).  > 1: this build's
                               # extension: [n_q] level embeddings summed in, n_q x K logits out

    @property
    def head_dim(self) -> int:
        return self.d_model // self.n_heads

    @staticmethod
    def native() -> "D3PMConfig":
        """What AR.__init__ really builds whatever it is passed (ar_discrete.py:207-240)."""
        return D3PMConfig()

    @staticmethod
    def libritts() -> "D3PMConfig":
        """BASELINE.json configs[1] / SURVEY.md §8d config 2: the widths get_model asks for
        (vall_e/vall_e/__init__.py:24-29), 750 live frames on a 768 canvas, 3 s prompt."""
        return D3PMConfig(d_model=512, n_heads=8, n_layers=6, canvas=768, n_frames=750,
                          s_text=50, s_prompt=225)

    @staticmethod
    def libritts_8q() -> "D3PMConfig":
        """The n_q = 8 extension of the LibriTTS config (BASELINE.json configs[1] "750 codec frames x 8 quantizers")."""
        return dataclasses.replace(D3PMConfig.libritts(), n_q=8)

    @staticmethod
    def vctk_long_prompt() -> "D3PMConfig":
        """10 s prompt, 5 s target, 200-step schedule."""
        return D3PMConfig(d_model=512, n_heads=8, n_layers=6, canvas=384, n_frames=375,
                          s_text=50, s_prompt=750, timesteps=200)


def state_dict_spec(cfg: D3PMConfig) -> dict[str, tuple[int, ...]]:
    """key -> shape, in the reference's layout (271 tensors at native size)."""
    d, K = cfg.d_model, cfg.n_classes
    spec: dict[str, tuple[int, ...]] = {
        "text_emb.weight": (K, d),
        "proms_emb.weight": (cfg.n_levels, K, d),
        "resps_emb.weight": (K, d) if cfg.n_q == 1 else (cfg.n_q, K, d),
        "time_emb.weight": (cfg.timesteps + 1, d),
        "token_emb.weight": (K, d),            # present in the reference state dict, never used
        "final.weight": (cfg.n_q * K, d),
        "final.bias": (cfg.n_q * K,),
    }

    def mha(prefix):
        spec[prefix + ".in_proj_weight"] = (3 * d, d)
        spec[prefix + ".in_proj_bias"] = (3 * d,)
        spec[prefix + ".out_proj.weight"] = (d, d)
        spec[prefix + ".out_proj.bias"] = (d,)

    def ln(prefix):
        spec[prefix + ".weight"] = (d,)
        spec[prefix + ".bias"] = (d,)

    def linear(prefix, n_out, n_in):
        spec[prefix + ".weight"] = (n_out, n_in)
        spec[prefix + ".bias"] = (n_out,)

    for enc, mult in (("encodertext", 2), ("encoder2", 3)):
        for j in range(cfg.cond_layers):
            p = f"{enc}.0.layers.{j}"
            mha(p + ".self_attn")
            linear(p + ".linear1", cfg.cond_ff, d)
            linear(p + ".linear2", d, cfg.cond_ff)
            ln(p + ".norm1")
            ln(p + ".norm2")
        linear(f"{enc}.1.fc1", mult * d, d)
        linear(f"{enc}.1.fc2", d, mult * d)
    for i in range(cfg.n_layers):
        p = f"blocks.{i}"
        for n in ("norm1", "norm2", "norm22", "norm3"):
            ln(f"{p}.{n}")
        for a in ("attn", "cross_attn", "cross_attn2"):   # cross_attn2 is dead weight upstream
            mha(f"{p}.{a}")
        linear(f"{p}.mlp.fc1", 4 * d, d)
        linear(f"{p}.mlp.fc2", d, 4 * d)
        linear(f"{p}.timestep_fc", 2 * d, d)
    return spec


def make_state_dict(cfg: D3PMConfig, seed: int = 0, logit_gain: float = 1.0) -> dict[str, torch.Tensor]:
    """fp32 synthetic weights. `logit_gain` scales final.weight (SURVEY §8c uses x30 to make the
    greedy mode unmask at all)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out: dict[str, torch.Tensor] = {}
    spec = state_dict_spec(cfg)
    for key in sorted(spec):
        shape = spec[key]
        leaf = key.rsplit(".", 1)[-1]
        if key.endswith("_emb.weight"):
            w = rng.standard_normal(shape)
            if key == "text_emb.weight" or (key == "resps_emb.weight" and cfg.n_q == 1):
                w[0] = 0.0                      # nn.Embedding(padding_idx=0) (ar_discrete.py:210,212)
        elif ".norm" in key:
            w = (1.0 if leaf == "weight" else 0.0) + 0.1 * rng.standard_normal(shape)
        elif key.endswith("in_proj_weight"):
            a = math.sqrt(6.0 / (shape[0] + shape[1]))
            w = rng.uniform(-a, a, shape)
        elif key.endswith("in_proj_bias") or key.endswith("out_proj.bias"):
            w = rng.uniform(-0.05, 0.05, shape)
        else:                                   # nn.Linear-style
            fan_in = shape[-1] if leaf == "weight" else spec[key[:-4] + "weight"][-1]
            a = 1.0 / math.sqrt(fan_in)
            w = rng.uniform(-a, a, shape)
        out[key] = torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32))
    if logit_gain != 1.0:
        out["final.weight"] = out["final.weight"] * logit_gain
    return out


def make_inputs(cfg: D3PMConfig, batch: int, seed: int = 1):
    """Synthetic (phonemes, prompt) lists: phonemes randint[1,70) of length U{10..50},
    prompt randint[0,1024) [s_prompt_raw, n_levels]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    texts, proms = [], []
    for _ in range(batch):
        n = int(rng.integers(10, cfg.s_text + 1))
        texts.append(torch.from_numpy(rng.integers(1, 70, size=n).astype(np.int64)))
        m = int(rng.integers(cfg.s_prompt // 2, cfg.s_prompt + 1))
        proms.append(torch.from_numpy(rng.integers(0, 1024, size=(m, cfg.n_levels)).astype(np.int64)))
    return texts, proms


# ---- stock NAR model (quantizer levels 1..7 after the D3PM sampler) ----------------------
@dataclasses.dataclass(frozen=True)
class NARConfig:
    """Shapes of /root/reference/vall_e/vall_e/nar.py `NAR(Base)`; the registry builds d=1024, 16 heads, 12 layers
    (`-half`: 512/8/12, `-quarter`: 256/4/12; vall_e/vall_e/__init__.py:34-57): head_dim is 64 in all of them."""
    d_model: int = 1024
    n_heads: int = 16
    n_layers: int = 12
    n_tokens: int = 1024
    n_prom_levels: int = 8
    n_resp_levels: int = 7

    @property
    def head_dim(self) -> int:
        return self.d_model // self.n_heads


def nar_state_dict_spec(cfg: NARConfig) -> dict[str, tuple[int, ...]]:
    d, K = cfg.d_model, cfg.n_tokens
    spec = {"sep": (d,), "text_emb.weight": (K, d), "proms_emb.weight": (cfg.n_prom_levels, K, d),
            "resps_emb.weight": (cfg.n_resp_levels, K, d), "classifier.weight": (K, d), "classifier.bias": (K,)}
    for i in range(cfg.n_layers):
        p = f"blocks.{i}"
        spec[f"{p}.attn.block.to_qkv.weight"] = (3 * d, d)
        spec[f"{p}.attn.block.to_out.weight"] = (d, d)
        spec[f"{p}.attn.block.to_out.bias"] = (d,)
        spec[f"{p}.attn.norm.emb.weight"] = (cfg.n_resp_levels, 2 * d)
        spec[f"{p}.ffn.block.0.weight"] = (4 * d, d)
        spec[f"{p}.ffn.block.0.bias"] = (4 * d,)
        spec[f"{p}.ffn.block.3.weight"] = (d, 4 * d)
        spec[f"{p}.ffn.block.3.bias"] = (d,)
        spec[f"{p}.ffn.norm.emb.weight"] = (cfg.n_resp_levels, 2 * d)
    return spec


def make_nar_state_dict(cfg: NARConfig, seed: int = 0) -> dict[str, torch.Tensor]:
    rng = np.random.Generator(np.random.PCG64(seed + 7919))
    out = {}
    spec = nar_state_dict_spec(cfg)
    for key in sorted(spec):
        shape = spec[key]
        if key.endswith("_emb.weight") or key == "sep":
            w = rng.standard_normal(shape)
        elif key.endswith("norm.emb.weight"):
            w = 0.1 * rng.standard_normal(shape)         # upstream initialises AdaLN to zeros; non-zero exercises it
        else:
            fan_in = shape[-1] if key.endswith("weight") else spec[key[:-4] + "weight"][-1]
            a = 1.0 / math.sqrt(fan_in)
            w = rng.uniform(-a, a, shape)
        out[key] = torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32))
    return out


def make_nar_inputs(batch: int, seed: int = 1, t_text=(8, 16), t_prom=(20, 32), t_resp=(30, 44), n_levels: int = 1):
    """Ragged synthetic (phonemes, prompt codes [t,8], response codes [t,n_levels]) lists."""
    rng = np.random.Generator(np.random.PCG64(seed + 104729))
    texts, proms, resps = [], [], []
    for _ in range(batch):
        texts.append(torch.from_numpy(rng.integers(1, 70, size=int(rng.integers(*t_text))).astype(np.int64)))
        proms.append(torch.from_numpy(rng.integers(0, 1024, size=(int(rng.integers(*t_prom)), 8)).astype(np.int64)))
        resps.append(torch.from_numpy(rng.integers(0, 1024, size=(int(rng.integers(*t_resp)), n_levels)).astype(np.int64)))
    return texts, proms, resps
