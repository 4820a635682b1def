"""A/B library, GPU box: start / end stamps (shader clocks and 100 MHz ticks) of every 32nd workgroup of the 32 x 32 x 16
self-attention kernels -> the clock held, the time one workgroup takes, how the workgroups of a CU follow each other."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tts-with-diffusion-model_amd"))
from vall_e.vall_e import _hip  # noqa: E402
import ctypes as C  # noqa: E402


def main():
    _hip.use_ab_library()
    B, T, H, hd = 32, 768, 8, 64
    d = H * hd
    qkv = torch.randn(B, T, 3 * d, device="cuda:0").to(torch.bfloat16)
    q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
    _hip.set_attn_query_groups(32)
    for arm, name in ((321, "plain"), (322, "pipelined"), (323, "plain, one workgroup per CU"), (324, "pipelined, one workgroup per CU")):
        _hip.set_attn_arm(arm)
        for _ in range(20):
            _hip.op_attention(q, k, v, H, math.sqrt(1.0 / hd))
        buf = (C.c_uint64 * 192)()
        _hip.check(_hip.lib().d3pm_debug_attn32_stamps(buf, 192), "stamps")
        rows = [[int(buf[i * 4 + j]) for j in range(4)] for i in range(48)]
        r0 = min(r[2] for r in rows)
        print(f"{name}: workgroup 32 i: start us, end us, duration us, cycles, GHz")
        for i, (c0, c1, t0, t1) in enumerate(rows):
            dur = (t1 - t0) / 100.0
            if i % 4 == 0:
                print(f"  wg {32 * i:5d}: {(t0 - r0) / 100.0:7.2f} {(t1 - r0) / 100.0:7.2f} {dur:6.2f} {c1 - c0:7d} {(c1 - c0) / max(dur, 1e-9) / 1e3:5.2f}")
        print(f"  kernel span {(max(r[3] for r in rows) - r0) / 100.0:.2f} us")
    _hip.set_attn_arm(0)


if __name__ == "__main__":
    main()
