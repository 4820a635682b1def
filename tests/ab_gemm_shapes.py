"""Which GEMM schedule wins on the projections of a DiT block at a given row count (not a test):
python tests/ab_gemm_shapes.py [rows ...]   default: 12288 (VCTK config, 32 x 384) 6144 (16 x 384) 24576 (libritts)
Interleaved repetitions in one process, medians; product library (shipped schedules only: d3pm_tuning.gemm_variant)."""
import math, statistics, sys, torch
sys.path[:0] = ["tts-with-diffusion-model_amd", "."]
import __graft_entry__ as g
g.build()
from vall_e.vall_e import _hip
DEV, dtype = "cuda", torch.bfloat16
ROWS = [int(a) for a in sys.argv[1:]] or [12288, 6144, 24576]
VARIANTS = [0, 2, 5, 6, 7, 8]


def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M in ROWS:
    for name, mul, N, K, act, res in (("qkv", 1, 1536, 512, 0, False), ("merged-q", 2, 512, 512, 0, False), ("out-proj", 1, 512, 512, 0, True),
                                      ("fc1+gelu", 1, 2048, 512, 1, False), ("fc2+res", 1, 512, 2048, 0, True)):
        m = M * mul
        x = torch.randn(m, K, device=DEV).to(dtype); w = (torch.randn(N, K, device=DEV) / math.sqrt(K)).to(dtype)
        b = torch.randn(N, device=DEV).to(dtype); y = torch.empty(m, N, device=DEV, dtype=dtype)
        r = torch.randn(m, N, device=DEV).to(dtype) if res else None
        f = lambda: _hip.op_linear(x, w, b, act=act, r1=r, family=_hip.FAMILY_MFMA, out=y, ldy=N)
        res_t = {v: [] for v in VARIANTS}
        for rep in range(5):
            for v in VARIANTS:
                _hip.set_gemm_variant(v)
                res_t[v].append(timeit(f))
        _hip.set_gemm_variant(0)
        line = f"M={m:6d} {name:9s}"
        for v in VARIANTS:
            t = statistics.median(res_t[v])
            line += f" | v{v}: {t:6.1f} us {2 * m * N * K / t / 1e6:5.0f} TF"
        print(line, flush=True)
