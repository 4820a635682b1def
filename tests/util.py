"""Helpers shared by the parity tests."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REPORT = {}      # measured parity numbers of a GPU run; conftest.py writes them to gpurun_out/parity_report.json


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def f16(bits: np.ndarray) -> torch.Tensor:
    """uint16 bit patterns -> torch.float16 tensor."""
    return torch.from_numpy(bits.view(np.int16).copy()).view(torch.float16)


def bits(x: torch.Tensor) -> np.ndarray:
    return x.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def ulp16_diff(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Distance in fp16 representable steps between two fp16 tensors (monotone integer mapping)."""
    def key(x):
        i = x.contiguous().view(torch.int16).to(torch.int32)
        return torch.where(i < 0, -(i & 0x7FFF), i)
    return (key(a.to(torch.float16)) - key(b.to(torch.float16))).abs()


def native_setup(dtype=torch.float16, batch=1):
    from oracle import d3pm_oracle as O
    from vall_e.vall_e import synth
    cfg = synth.D3PMConfig.native()
    sd32 = synth.make_state_dict(cfg, 0)
    texts, proms = synth.make_inputs(cfg, max(batch, 2), 1)
    orc = O.Oracle({k: v.to(dtype) for k, v in sd32.items()}, O.Shape.of(cfg))
    return cfg, sd32, texts, proms, orc
