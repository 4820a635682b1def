"""A/B library, GPU box: in-kernel timeline of the 32 x 32 x 16 self-attention kernel (shader-clock stamps of wave 0 of two
workgroups at eight points of every key tile).    python tests/ab_attn32_stamps.py"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tts-with-diffusion-model_amd"))
from vall_e.vall_e import _hip  # noqa: E402

NAMES = ["loads issued", "scores (K reads, QK, max)", "softmax (exp, sum, pack)", "V reads + PV issued", "wait global loads",
         "LDS stores issued", "barrier"]


def main():
    _hip.use_ab_library()
    B, T, H, hd = 32, 768, 8, 64
    d = H * hd
    qkv = torch.randn(B, T, 3 * d, device="cuda:0").to(torch.bfloat16)
    q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
    _hip.set_attn_query_groups(32)
    _hip.set_attn_arm(320)
    for _ in range(3):
        _hip.op_attention(q, k, v, H, math.sqrt(1.0 / hd))
    st = _hip.attn32_stamps()
    _hip.set_attn_arm(0)
    for slot in range(2):
        print(f"workgroup slot {slot}: cycles per phase, tiles 0..11 (wave 0)")
        tot = [0] * 7
        for t in range(12):
            row = st[slot][t]
            ph = [row[i + 1] - row[i] for i in range(7)]
            nxt = st[slot][t + 1][0] - row[7] if t + 1 < 12 else 0
            print(f"  tile {t:2d}: " + " ".join(f"{x:6d}" for x in ph) + f" | tile total {row[7] - row[0]:6d}  gap to next {nxt}")
            if 2 <= t <= 10:
                tot = [a + b for a, b in zip(tot, ph)]
        print("  mean of tiles 2..10: " + ", ".join(f"{n} {x / 9:.0f}" for n, x in zip(NAMES, tot)))
        print(f"  whole walk: {st[slot][11][7] - st[slot][0][0]} cycles")


if __name__ == "__main__":
    main()
