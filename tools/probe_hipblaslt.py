"""Probe (not part of the product): what the vendor library reaches on the block's projection shapes, for calibration of
profiles/*ab_gemm*.txt.  torch.nn.functional.linear on bf16 operands runs hipBLASLt / rocBLAS here; the product never calls it
(its GEMMs carry the reference's bias / GELU / residual / rounding epilogues and row-panel LayerNorms).   python tools/probe_hipblaslt.py"""
import math, statistics, torch
DEV, dtype = "cuda", torch.bfloat16


def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, M, N, K, bias in (("qkv", 24576, 1536, 512, True), ("merged-q", 49152, 512, 512, True), ("proj", 24576, 512, 512, True),
                            ("fc1", 24576, 2048, 512, True), ("fc2", 24576, 512, 2048, True), ("final", 24576, 1025, 512, True)):
    x = torch.randn(M, K, device=DEV).to(dtype); w = (torch.randn(N, K, device=DEV) / math.sqrt(K)).to(dtype)
    b = torch.randn(N, device=DEV).to(dtype)
    ts = [timeit(lambda: torch.nn.functional.linear(x, w, b if bias else None)) for _ in range(5)]
    t = statistics.median(ts)
    print(f"{name:9s} {M} x {N} x {K}: vendor library {t:6.1f} us  {2 * M * N * K / t / 1e6:6.0f} TFLOP/s (bias add included, no activation / residual)", flush=True)
