"""fp8 vs bf16 GEMM on the LayerNorm-fed shapes, interleaved in one process."""
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


for name, M, N, K, act in (("qkv", 24576, 1536, 512, 0), ("merged-q", 49152, 512, 512, 0), ("fc1+gelu", 24576, 2048, 512, 1)):
    x = torch.randn(M, K, device=DEV).to(dtype); w = (torch.randn(N, K, device=DEV) / math.sqrt(K)).to(dtype)
    b = torch.randn(N, device=DEV).to(dtype); y = torch.empty(M, N, device=DEV, dtype=dtype)
    x8, sx = _hip.quantize_rows_e4m3(x); w8, sw = _hip.quantize_rows_e4m3(w)
    f16 = lambda: _hip.op_linear(x, w, b, act=act, family=_hip.FAMILY_MFMA, out=y, ldy=N)
    f8 = lambda: _hip.op_linear_fp8(x8, sx, w8, sw, b, dtype, act=act)
    t16, t8 = [], []
    for rep in range(7):
        t16.append(timeit(f16)); t8.append(timeit(f8))
    print(f"{name:9s} bf16 {statistics.median(t16):7.1f} us | fp8 {statistics.median(t8):7.1f} us", flush=True)
g = torch.Generator(device="cpu").manual_seed(0)
x = torch.randn(24576, 512, device=DEV).to(dtype); w = torch.ones(512, device=DEV, dtype=dtype); b = torch.zeros(512, device=DEV, dtype=dtype)
print(f"layernorm bf16 {timeit(lambda: _hip.op_layernorm(x, w, b)):6.1f} us | fp8 rows {timeit(lambda: _hip.op_layernorm_fp8(x, w, b)):6.1f} us")
