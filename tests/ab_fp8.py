"""Block-scaled fp8 (MX) vs bf16 GEMM on the projections of a DiT block at the bench batch, interleaved in one process
(not a test): python tests/ab_fp8.py"""
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


SHAPES = (("qkv", 24576, 1536, 512, 0, False, False), ("merged-q", 49152, 512, 512, 0, False, False),
          ("fc1+gelu", 24576, 2048, 512, 1, False, False), ("fc1+gelu->mx", 24576, 2048, 512, 1, False, True),
          ("fc2+res", 24576, 512, 2048, 0, True, False))
for name, M, N, K, act, res, mx_out in SHAPES:
    x = torch.randn(M, K, device=DEV).to(dtype); w = (torch.randn(N, K, device=DEV) / math.sqrt(K)).to(dtype)
    b = torch.randn(N, device=DEV).to(dtype); y = torch.empty(M, N, device=DEV, dtype=dtype)
    r = torch.randn(M, N, device=DEV).to(dtype) if res else None
    x8, sx = _hip.op_quantize_mx(x); w8, sw = _hip.quantize_mx(w)
    f16 = lambda: _hip.op_linear(x, w, b, act=act, r1=r, family=_hip.FAMILY_MFMA, out=y, ldy=N)
    f8 = lambda: _hip.op_linear_mx(x8, sx, w8, sw, b, dtype, act=act, r1=r, mx_out=mx_out)
    t16, t8, t8a, t8b = [], [], [], []
    for rep in range(7):
        t16.append(timeit(f16)); t8.append(timeit(f8))
        _hip.set_gemm_variant(6); t8a.append(timeit(f8)); _hip.set_gemm_variant(8); t8b.append(timeit(f8)); _hip.set_gemm_variant(0)
    a, c = statistics.median(t16), statistics.median(t8)
    print(f"{name:13s} bf16 {a:7.1f} us {2 * M * N * K / a / 1e6:6.0f} TF/s | mx-fp8 {c:7.1f} us {2 * M * N * K / c / 1e6:6.0f} TF/s | x{a / c:4.2f}"
          f" | 192x256: {statistics.median(t8a):6.1f} us, 192x128 x 2: {statistics.median(t8b):6.1f} us", flush=True)
x = torch.randn(24576, 512, device=DEV).to(dtype); w = torch.ones(512, device=DEV, dtype=dtype); b = torch.zeros(512, device=DEV, dtype=dtype)
print(f"layernorm bf16 {timeit(lambda: _hip.op_layernorm(x, w, b)):6.1f} us | -> mx rows {timeit(lambda: _hip.op_layernorm_mx(x, w, b)):6.1f} us")
