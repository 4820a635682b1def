"""Kernel micro-benchmarks on the GPU box (not a test): TFLOP/s of the GEMM / attention kernels at the
bench shapes, per GEMM variant.  python tests/bench_kernels.py > gpurun_out/kernels.txt"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tts-with-diffusion-model_amd"), ROOT]
import torch  # noqa: E402

import __graft_entry__ as g  # noqa: E402

g.build()
from vall_e.vall_e import _hip  # noqa: E402

DEV = "cuda:0"


def timeit(fn, iters=20):
    if "--quick" in sys.argv:
        iters = 3
    for _ in range(1 if "--quick" in sys.argv else 3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    quick = "--quick" in sys.argv      # PMC passes: default GEMM variant only, 3 timed launches each
    dtype = torch.bfloat16
    M = 32 * 768
    print(f"# GEMM  M={M}  dtype={dtype}")
    shapes = [("qkv", 1536, 512, 0, False), ("proj", 512, 512, 0, True), ("fc1+gelu", 2048, 512, 1, False),
              ("fc2+res", 512, 2048, 0, True), ("final", 1025, 512, 0, False)]
    for name, N, K, act, res in shapes:
        x = torch.randn(M, K, device=DEV).to(dtype)
        w = (torch.randn(N, K, device=DEV) / math.sqrt(K)).to(dtype)
        b = torch.randn(N, device=DEV).to(dtype)
        ldy = (N + 7) // 8 * 8
        y = torch.empty(M, ldy, device=DEV, dtype=dtype)
        r = torch.randn(M, ldy, device=DEV).to(dtype) if res else None
        line = f"{name:10s} N={N:5d} K={K:5d}"
        for variant in ((0,) if quick else (0, 2, 5)):
            _hip.set_gemm_variant(variant)
            t = timeit(lambda: _hip.op_linear(x, w, b, act=act, r1=r, family=_hip.FAMILY_MFMA, out=y, ldy=ldy))
            line += f" | v{variant}: {t * 1e6:7.1f} us {2 * M * N * K / t / 1e12:7.1f} TF/s"
        if "--slots" in sys.argv:
            _hip.set_gemm_variant(2)
            for slots in (512, 768, 1280):
                _hip.set_gemm_persist_slots(slots)
                t = timeit(lambda: _hip.op_linear(x, w, b, act=act, r1=r, family=_hip.FAMILY_MFMA, out=y, ldy=ldy))
                line += f" | s{slots}: {t * 1e6:6.1f}"
            _hip.set_gemm_persist_slots(1024)
        print(line, flush=True)
    _hip.set_gemm_variant(0)
    if "--small" in sys.argv or not quick:
        Ms = 768
        print(f"# GEMM  M={Ms} (one utterance: latency regime)")
        for name, N, K, act, res in shapes:
            x = torch.randn(Ms, K, device=DEV).to(dtype)
            w = (torch.randn(N, K, device=DEV) / math.sqrt(K)).to(dtype)
            b = torch.randn(N, device=DEV).to(dtype)
            ldy = (N + 7) // 8 * 8
            y = torch.empty(Ms, ldy, device=DEV, dtype=dtype)
            r = torch.randn(Ms, ldy, device=DEV).to(dtype) if res else None
            line = f"{name:10s} N={N:5d} K={K:5d}"
            for variant in (2, 3, 4):
                _hip.set_gemm_variant(variant)
                t = timeit(lambda: _hip.op_linear(x, w, b, act=act, r1=r, family=_hip.FAMILY_MFMA, out=y, ldy=ldy), 50)
                line += f" | v{variant}: {t * 1e6:6.1f} us"
            print(line, flush=True)
        _hip.set_gemm_variant(0)
    print("# attention  B=32 H=8 hd=64")
    for name, Tq, S in (("self", 768, 768), ("text", 768, 50), ("prompt", 768, 225)):
        q = torch.randn(32, Tq, 512, device=DEV).to(dtype)
        kv = torch.randn(32, S, 1024, device=DEV).to(dtype)
        line = f"{name:8s} Tq={Tq} S={S:4d}:"
        for qg in ((0,) if quick else (1, 2, 32)):
            _hip.set_attn_query_groups(qg)
            t = timeit(lambda: _hip.op_attention(q, kv[..., :512], kv[..., 512:], 8, 0.125, family=_hip.FAMILY_MFMA))
            line += f" | qg{qg}: {t * 1e6:7.1f} us {4 * 32 * 8 * Tq * S * 64 / t / 1e12:7.1f} TF/s"
        print(line, flush=True)
    _hip.set_attn_query_groups(0)
    print("# layernorm  N=24576 d=512")
    x = torch.randn(M, 512, device=DEV).to(dtype)
    w = torch.randn(512, device=DEV).to(dtype)
    t = timeit(lambda: _hip.op_layernorm(x, w, w))
    print(f"layernorm: {t * 1e6:7.1f} us {2 * M * 512 * 2 / t / 1e9:7.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
