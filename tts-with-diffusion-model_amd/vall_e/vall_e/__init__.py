"""Model registry with the reference's surface (/root/reference/vall_e/vall_e/__init__.py:7-59).

Only the discrete-diffusion sampler is implemented by this package (the stock AR / NAR VALL-E
models are out of its scope); their names raise NotImplementedError instead of
silently building something else.  Importing this package has no argv side effect (the reference's
`from ..config import cfg` parses sys.argv at import, config.py:96).
"""
from .ar_discrete import AR
from .synth import D3PMConfig


def get_model(name: str):
    """`name.lower().startswith("diffusion")` -> AR(512, 100, 1024, 8, 8, 6) on the GPU, positional
    and in that order exactly as the reference registry passes them (__init__.py:22-31)."""
    name = name.lower()
    if name.startswith("diffusion"):
        max_n_levels, n_tokens, d_model, n_steps, n_heads, num_layers = 8, 1024, 512, 100, 8, 6
        return AR(d_model, n_steps, n_tokens, max_n_levels, n_heads, num_layers).to("cuda")
    if name.startswith("ar") or name.startswith("nar"):
        raise NotImplementedError(f"{name}: the stock VALL-E AR/NAR models are not part of the D3PM sampler build")
    raise ValueError("Model name should start with AR or NAR.")


__all__ = ["AR", "D3PMConfig", "get_model"]
