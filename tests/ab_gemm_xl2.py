"""Probe (not a test): how much of a big-tile GEMM launch is the X operand coming from HBM / Infinity Cache?
Arm "x0": every row of X is the SAME 2-KiB row (row stride 0), so X is served by L2 / L1 while W, the MFMAs and the output
stores stay exactly as they are.  python tests/ab_gemm_xl2.py"""
import math, statistics, sys, torch
sys.path.insert(0, "tts-with-diffusion-model_amd")
from vall_e.vall_e import _hip
DEV, dtype = "cuda", torch.bfloat16


def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, M, N, K, act, res in [("qkv", 24576, 1536, 512, 0, False), ("merged-q", 49152, 512, 512, 0, False),
                                ("fc1+gelu", 24576, 2048, 512, 1, False), ("fc2+res", 24576, 512, 2048, 0, True)]:
    x = torch.randn(M, K, device=DEV).to(dtype); w = (torch.randn(N, K, device=DEV) / math.sqrt(K)).to(dtype)
    x0 = x[:1].expand(M, K)
    b = torch.randn(N, device=DEV).to(dtype); y = torch.empty(M, N, device=DEV, dtype=dtype)
    r = torch.randn(M, N, device=DEV).to(dtype) if res else None
    arms = {"x": lambda: _hip.op_linear(x, w, b, act=act, r1=r, family=_hip.FAMILY_MFMA, out=y, ldy=N),
            "x0": lambda: _hip.op_linear(x0, w, b, act=act, r1=r, family=_hip.FAMILY_MFMA, out=y, ldy=N)}
    t = {k: [] for k in arms}
    for rep in range(7):
        for k, f in arms.items():
            t[k].append(timeit(f))
    print(f"{name:9s} X from memory {statistics.median(t['x']):6.1f} us | X from L2 (row stride 0) {statistics.median(t['x0']):6.1f} us", flush=True)
