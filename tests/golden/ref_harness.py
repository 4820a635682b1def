"""Container-only loader for the upstream reference's D3PM model (test infrastructure).

Loads /root/reference/vall_e/vall_e/{base,ar_discrete}.py *by path* (bypassing the package
__init__ that needs omegaconf/diskcache/deepspeed) with three shims, exactly as SURVEY.md §8c
prescribes:

  * stub modules for `diffusers` (imported, never used) and `timm.models.vision_transformer`
    (only `Mlp` is executed -- restated here from timm's published definition
    fc1 -> act -> drop1 -> norm(Identity) -> fc2 -> drop2);
  * "cuda*" device strings rewritten to "cpu" (the reference hard-codes them,
    ar_discrete.py:269,275,277,700,751);
  * nothing else: every arithmetic op executed is the reference's own.

This file never travels to the GPU box's tests (nothing under -m gpu imports it) and is used
only by make_golden.py and by the CPU-side pinning tests, which skip when /root/reference is absent.
"""
from __future__ import annotations

import contextlib
import importlib.util
import os
import sys
import types

import torch
from torch import nn

REF_ROOT = os.environ.get("D3PM_REFERENCE_ROOT", "/root/reference")
_PKG = "_upstream_valle"


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "vall_e", "vall_e", "ar_discrete.py"))


class _TimmMlp(nn.Module):
    """timm.layers.Mlp as published: Linear -> act -> Dropout -> Identity -> Linear -> Dropout."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU,
                 norm_layer=None, bias=True, drop=0.0, use_conv=False):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features, bias=bias)
        self.act = act_layer()
        self.drop1 = nn.Dropout(drop)
        self.norm = norm_layer(hidden_features) if norm_layer is not None else nn.Identity()
        self.fc2 = nn.Linear(hidden_features, out_features, bias=bias)
        self.drop2 = nn.Dropout(drop)

    def forward(self, x):
        return self.drop2(self.fc2(self.norm(self.drop1(self.act(self.fc1(x))))))


def _install_stubs():
    if "diffusers" not in sys.modules:
        m = types.ModuleType("diffusers")
        for name in ("UNet3DConditionModel", "UNet2DConditionModel", "DDPMScheduler",
                     "CosineDPMSolverMultistepScheduler", "DDIMScheduler"):
            setattr(m, name, type(name, (), {}))
        sys.modules["diffusers"] = m
    if "timm" not in sys.modules:
        timm = types.ModuleType("timm")
        models = types.ModuleType("timm.models")
        vt = types.ModuleType("timm.models.vision_transformer")
        vt.Mlp = _TimmMlp
        vt.PatchEmbed = type("PatchEmbed", (), {})
        vt.Attention = type("Attention", (), {})
        timm.models = models
        models.vision_transformer = vt
        sys.modules.update({"timm": timm, "timm.models": models,
                            "timm.models.vision_transformer": vt})


def _cpuify(dev):
    if isinstance(dev, str) and dev.startswith("cuda"):
        return "cpu"
    if isinstance(dev, torch.device) and dev.type == "cuda":
        return torch.device("cpu")
    return dev


@contextlib.contextmanager
def cuda_strings_as_cpu():
    """Rewrite the reference's hard-coded "cuda"/"cuda:0" literals to "cpu" while active."""
    orig_to, orig_tensor, orig_full = torch.Tensor.to, torch.tensor, torch.full

    def to(self, *a, **k):
        a = tuple(_cpuify(x) for x in a)
        if "device" in k:
            k["device"] = _cpuify(k["device"])
        return orig_to(self, *a, **k)

    def tensor(*a, **k):
        if "device" in k:
            k["device"] = _cpuify(k["device"])
        return orig_tensor(*a, **k)

    torch.Tensor.to, torch.tensor = to, tensor
    try:
        yield
    finally:
        torch.Tensor.to, torch.tensor, torch.full = orig_to, orig_tensor, orig_full


def load_reference_modules():
    """Returns (base_module, ar_discrete_module) executed from the reference sources."""
    if _PKG + ".ar_discrete" in sys.modules:
        return sys.modules[_PKG + ".base"], sys.modules[_PKG + ".ar_discrete"]
    _install_stubs()
    pkg_dir = os.path.join(REF_ROOT, "vall_e", "vall_e")
    pkg = types.ModuleType(_PKG)
    pkg.__path__ = [pkg_dir]
    sys.modules[_PKG] = pkg
    mods = []
    for name in ("base", "ar_discrete"):
        spec = importlib.util.spec_from_file_location(f"{_PKG}.{name}", os.path.join(pkg_dir, name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"{_PKG}.{name}"] = mod
        spec.loader.exec_module(mod)
        mods.append(mod)
    return tuple(mods)


def build_reference_native():
    """The reference's D3PM model exactly as its registry builds it (vall_e/vall_e/__init__.py:22-31):
    AR(512, 100, 1024, 8, 8, 6) -- the ctor ignores all but max_n_levels and forces d=32,H=16,L=8."""
    _, ard = load_reference_modules()
    with cuda_strings_as_cpu():
        model = ard.AR(512, 100, 1024, 8, 8, 6)
    return model.eval()


def load_reference_data_module():
    """/root/reference/vall_e/data.py executed from source (its `_load_quants`, `_get_phones`, `VALLEDatset` symmaps are the
    reference's statement of the on-disk formats).  Its package imports need omegaconf / diskcache (`.config`), absent
    here: `.config` is replaced by a stub carrying the three fields data.py reads, with the reference's own defaults
    (config.py:41-44); `.sampler` is the reference's file."""
    name = _PKG + "_top"
    if name + ".data" in sys.modules:
        return sys.modules[name + ".data"]
    top_dir = os.path.join(REF_ROOT, "vall_e")
    pkg = types.ModuleType(name)
    pkg.__path__ = [top_dir]
    sys.modules[name] = pkg
    cfg_mod = types.ModuleType(name + ".config")

    class _Cfg:
        min_phones, max_phones = 10, 50                  # config.py:43-44
        max_prompts, p_additional_prompt = 3, 0.8
        diskcache = staticmethod(lambda: lambda fn: fn)  # config.py:90-93 with cache_dataloader off: the identity decorator
        get_spkr = staticmethod(lambda p: p.parts[-2])   # config.py:41 is `p.parts[-1]` (the file name itself); the LibriTTS /
                                                         # VCTK yml files of the reference override it with parts[-2]
    cfg_mod.cfg = _Cfg()
    sys.modules[name + ".config"] = cfg_mod
    for mod_name in ("sampler", "data"):
        spec = importlib.util.spec_from_file_location(f"{name}.{mod_name}", os.path.join(top_dir, mod_name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"{name}.{mod_name}"] = mod
        spec.loader.exec_module(mod)
    return sys.modules[name + ".data"]
