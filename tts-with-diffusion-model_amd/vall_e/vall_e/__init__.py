"""Model registry with the reference's surface (/root/reference/vall_e/vall_e/__init__.py:7-59).

This package implements the discrete-diffusion sampler ("diffusion*") and the stock NAR model that completes its
output ("nar*"); the stock causal AR model is out of scope and raises NotImplementedError.  Importing this package has no argv side effect (the reference's
`from ..config import cfg` parses sys.argv at import, config.py:96).
"""
from .ar_discrete import AR
from .nar import NAR
from .synth import D3PMConfig, NARConfig


def get_model(name: str):
    """`name.lower().startswith("diffusion")` -> AR(512, 100, 1024, 8, 8, 6) on the GPU, positional
    and in that order exactly as the reference registry passes them (__init__.py:22-31)."""
    name = name.lower()
    if name.startswith("diffusion"):
        max_n_levels, n_tokens, d_model, n_steps, n_heads, num_layers = 8, 1024, 512, 100, 8, 6
        return AR(d_model, n_steps, n_tokens, max_n_levels, n_heads, num_layers).to("cuda")
    if name.startswith("nar"):
        # the reference sizes (__init__.py:34-57): -quarter 256/4/12, -half 512/8/12, default 1024/16/12; 1024 tokens
        if "-quarter" in name:
            return NAR(1024, d_model=256, n_heads=4, n_layers=12).to("cuda")
        if "-half" in name:
            return NAR(1024, d_model=512, n_heads=8, n_layers=12).to("cuda")
        if name != "nar":
            raise NotImplementedError(name)
        return NAR(1024, d_model=1024, n_heads=16, n_layers=12).to("cuda")
    if name.startswith("ar"):
        raise NotImplementedError(f"{name}: the stock causal VALL-E AR model is not part of this build")
    raise ValueError("Model name should start with AR or NAR.")


__all__ = ["AR", "NAR", "D3PMConfig", "NARConfig", "get_model"]
