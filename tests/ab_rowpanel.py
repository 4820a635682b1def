"""Timing of the row-panel launches against the launches they replace (not a test): python tests/ab_rowpanel.py"""
import math, statistics, sys, torch
sys.path[:0] = ["tts-with-diffusion-model_amd", "."]
import __graft_entry__ as g
g.build_ab()                       # libd3pm_hip_ab.so: the no-LayerNorm timing build is not in the product
from vall_e.vall_e import _hip
_hip.use_ab_library()
DEV, dtype = "cuda", torch.bfloat16
M, K = 24576, 512
g = torch.Generator(device="cpu").manual_seed(0)
mk = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dtype).to(DEV)
x, x2, r1 = mk(M, K), mk(M, K), mk(M, 512)
w, b = mk(512, K, sc=1 / math.sqrt(K)), mk(512)
lw, lb, lw2, lb2, film = mk(512) + 1, mk(512), mk(512) + 1, mk(512), mk(1024, sc=0.3)


def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def sep_self():
    y = _hip.op_linear(x, w, b, r1=r1, family=_hip.FAMILY_MFMA)
    _hip.op_layernorm(y, lw, lb); _hip.op_layernorm(y, lw2, lb2)


arms = {"self_out fused": lambda: _hip.op_linear_rowpanel(x, w, b, r1, lw, lb, ln2_w=lw2, ln2_b=lb2),
        "self_out GEMM only (192x256)": lambda: _hip.op_linear(x, w, b, r1=r1, family=_hip.FAMILY_MFMA),
        "cross_out fused": lambda: _hip.op_linear_rowpanel(x, w, b, r1, lw, lb, x2=x2, film=film)}
modes = [(1, "")] + ([(1025, " [no LN arithmetic]")] if "--abl" in sys.argv else [])
for mode, tag in modes:
    _hip.set_gemm_big_mode(mode)
    for name, f in arms.items():
        t = statistics.median(timeit(f) for _ in range(5))
        print(f"{name + tag:48s} {t:6.1f} us", flush=True)
_hip.set_gemm_big_mode(1)
