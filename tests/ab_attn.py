"""Timing-only ablations of the MFMA self-attention kernel (not a test): python tests/ab_attn.py
Runs on libd3pm_hip_ab.so (include/d3pm_hip_ab.h).  Arms: the shipped kernel and builds with parts removed (results wrong by construction) -- what each part costs in place."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tts-with-diffusion-model_amd"), ROOT]
import __graft_entry__ as g  # noqa: E402

g.build_ab()                       # libd3pm_hip_ab.so: the ablation arms are not in the product
from vall_e.vall_e import _hip  # noqa: E402
_hip.use_ab_library()

DEV = "cuda:0"
B, T, H, d = 32, 768, 8, 512
torch.manual_seed(0)
qkv = torch.randn(B, T, 3 * d, device=DEV).to(torch.bfloat16)
q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


ARMS = [(2, "shipped (QG 2)"), (228, "K/V tiles by direct-to-LDS DMA (same results)"), (164, "all fragment reads of a tile issued at its top (same results)"), (3, "QG 3 (48 queries per wave)"), (201, "shipped, two workgroups per CU"), (202, "shipped, one workgroup per CU"), (1, "QG 1"), (101, "no v_exp"), (102, "no max / rescale"), (103, "no exp, no max"), (104, "no staging"),
        (112, "no staging, no barriers"), (116, "no P.V"), (132, "no Q.K"), (148, "no MFMA products"), (115, "only the MFMA products"),
        (160, "only softmax VALU")]
arms = [int(a) for a in sys.argv[1:]] or [a for a, _ in ARMS]
names = dict(ARMS)
flops = 4.0 * B * H * T * T * 64
ref = None
for arm in arms:
    if arm in (1, 2):
        _hip.set_attn_arm(0); _hip.set_attn_query_groups(arm)
    else:
        _hip.set_attn_query_groups(0); _hip.set_attn_arm(arm)
    if arm in (1, 2, 3, 164, 228, 201, 202):                  # the arms that must give the shipped kernel's bits
        o = _hip.op_attention(q, k, v, H, 0.125, family=_hip.FAMILY_MFMA)
        ref = o.clone() if ref is None else ref
        assert torch.equal(o, ref), f"arm {arm}: output differs from the first arm"
    t = timeit(lambda: _hip.op_attention(q, k, v, H, 0.125, family=_hip.FAMILY_MFMA))
    print("%-28s %7.1f us  %6.0f TFLOP/s-equivalent" % (names.get(arm, str(arm)), t, flops / t / 1e6), flush=True)
_hip.set_attn_query_groups(0)
_hip.set_attn_arm(0)
