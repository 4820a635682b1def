"""Interleaved A/B of GEMM schedules inside one process: medians over alternating repetitions.
arguments: <variant>[m<big mode>] ...   variants: 2 = 128x128 persistent (round-1 default), 6 = 192x256 big tile,
7 = 96x512 big tile, 9 = the five-slab ring; big modes: 0 compiler-placed reads, 1 hand-placed reads, 32769 = 1 + epilogue operands prefetched at the top of the tile,
16 / 17 / 32 timing-only ablations."""
import math, statistics, sys, torch
sys.path[:0] = ["tts-with-diffusion-model_amd", "."]
import __graft_entry__ as g
g.build_ab()                       # libd3pm_hip_ab.so (include/d3pm_hip_ab.h): big modes other than 1 and the ring are not in the product
from vall_e.vall_e import _hip
_hip.use_ab_library()
DEV, dtype = "cuda", torch.bfloat16
ARMS = [(int(a.split("m")[0]), int(a.split("m")[1]) if "m" in a else 1) for a in (sys.argv[1:] or ["2", "6", "7", "8", "0"])]


def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


shapes = [("qkv", 24576, 1536, 512, 0, False), ("merged-q", 49152, 512, 512, 0, False), ("proj", 24576, 512, 512, 0, True),
          ("fc1+gelu", 24576, 2048, 512, 1, False), ("fc2+res", 24576, 512, 2048, 0, True)]
for name, M, N, K, act, res in shapes:
    x = torch.randn(M, K, device=DEV).to(dtype); w = (torch.randn(N, K, device=DEV) / math.sqrt(K)).to(dtype)
    b = torch.randn(N, device=DEV).to(dtype); y = torch.empty(M, N, device=DEV, dtype=dtype)
    r = torch.randn(M, N, device=DEV).to(dtype) if res else None
    f = lambda: _hip.op_linear(x, w, b, act=act, r1=r, family=_hip.FAMILY_MFMA, out=y, ldy=N)
    res_t = {a: [] for a in ARMS}
    outs, clocks = {}, {}
    for rep in range(7):
        for arm in ARMS:
            _hip.set_gemm_ring(arm[0] == 9); _hip.set_gemm_variant(0 if arm[0] == 9 else arm[0]); _hip.set_gemm_big_mode(arm[1])
            res_t[arm].append(timeit(f))
            if arm[1] & 256 and rep == 6:
                clocks[arm] = _hip.gemm_clock_ghz()
            if rep == 0 and (arm[1] < 16 or arm[1] == 32769):      # result-preserving modes
                outs[arm] = y.clone()
    ref = next(iter(outs.values()))
    same = all(torch.equal(ref, o) for o in outs.values())
    line = f"{name:9s}"
    for arm in ARMS:
        t = statistics.median(res_t[arm])
        line += f" | v{arm[0]}m{arm[1]}: {t:6.1f} us {2 * M * N * K / t / 1e6:6.0f} TF/s" + (f" @{clocks[arm]:.2f} GHz" if arm in clocks else "")
    print(line + f" | bit-identical: {same}", flush=True)
_hip.set_gemm_variant(0); _hip.set_gemm_big_mode(1); _hip.set_gemm_ring(False)
