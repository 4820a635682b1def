"""A/B timing of the folded-LayerNorm GEMM epilogues against the plain launches they replace (MI355X, M = 32 x 768 rows).
    python tests/ab_fold.py            # prints us per launch, hot (back to back) and cold (a 600 MB write between launches)
Not a test: a measurement script (profiles/round4_*_ab_fold.txt)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tts-with-diffusion-model_amd"), ROOT]
from vall_e.vall_e import _hip  # noqa: E402

DEV = "cuda:0"


def timeit(fn, reps=30, flush=None):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(reps):
        if flush is not None:
            flush.add_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / reps * 1e3


def main():
    torch.manual_seed(0)
    M, d = 32 * 768, 512
    dt = torch.bfloat16
    flush = torch.zeros(300 * 1024 * 1024 // 2, dtype=dt, device=DEV)
    x = torch.randn(M, d, device=DEV).to(dt)
    gamma = (1 + 0.1 * torch.randn(d, device=DEV)).to(dt)
    beta = (0.1 * torch.randn(d, device=DEV)).to(dt)
    h = _hip.op_layernorm(x, gamma, beta)
    st = _hip.op_row_stats(x)
    for name, N, act in (("qkv", 1536, 0), ("q2", 1024, 0), ("fc1+gelu", 2048, 1)):
        w = (torch.randn(N, d, device=DEV) / d ** 0.5).to(dt)
        b = (0.1 * torch.randn(N, device=DEV)).to(dt)
        wf, fs, fb = _hip.op_fold_weights(w, b, gamma, beta)
        y = torch.empty(M, N, dtype=dt, device=DEV)
        for mode, fl in (("hot", None), ("cold", flush)):
            t_plain = timeit(lambda: _hip.op_linear(h, w, b, act=act, out=y), flush=fl)
            t_plain2 = timeit(lambda: _hip.op_linear(x, wf, b, act=act, out=y), flush=fl)
            t_fold = timeit(lambda: _hip.op_linear_fold(x, wf, fs, fb, st, act=act), flush=fl)
            print(f"{name:10s} {mode:5s} plain {t_plain:7.1f} us   plain on (x, wf) {t_plain2:7.1f} us   folded {t_fold:7.1f} us", flush=True)
    r1 = torch.randn(M, d, device=DEV).to(dt)
    for name, K in (("out+r1", 512), ("fc2+r1", 2048)):
        xx = torch.randn(M, K, device=DEV).to(dt)
        w = (torch.randn(d, K, device=DEV) / K ** 0.5).to(dt)
        b = (0.1 * torch.randn(d, device=DEV)).to(dt)
        y = torch.empty(M, d, dtype=dt, device=DEV)
        for mode, fl in (("hot", None), ("cold", flush)):
            t_plain = timeit(lambda: _hip.op_linear(xx, w, b, r1=r1, out=y), flush=fl)
            t_st = timeit(lambda: _hip.op_linear_stats(xx, w, b, r1), flush=fl)      # allocates and zeroes its outputs: ~ +4 us of fill kernels
            print(f"{name:10s} {mode:5s} plain {t_plain:7.1f} us   +stats {t_st:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
