"""A/B on the GPU box: the 32 x 32 x 16 self-attention kernel (attn_query_groups = 32) against the 16 x 16 x 32 one (1 / 2):
error against torch fp32 on the same inputs, and time per launch at the bench shape and the VCTK shape.
    python tests/ab_attn32.py
"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tts-with-diffusion-model_amd"))
from vall_e.vall_e import _hip  # noqa: E402

_hip.use_ab_library()      # arm 35 (192-query workgroups) lives in libd3pm_hip_ab.so only: __graft_entry__.build_ab()

DEV = "cuda:0"


def torch_attention(q, k, v, H, scale):
    B, Tq, d = q.shape
    S = k.shape[1]
    hd = d // H
    qf = q.float().view(B, Tq, H, hd).transpose(1, 2) * scale
    kf = k.float().view(B, S, H, hd).transpose(1, 2)
    vf = v.float().view(B, S, H, hd).transpose(1, 2)
    p = torch.softmax(qf @ kf.transpose(-1, -2), dim=-1)
    return (p @ vf).transpose(1, 2).reshape(B, Tq, d)


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
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    H, hd = 8, 64
    d = H * hd
    scale = math.sqrt(1.0 / hd)
    for dtype in (torch.bfloat16, torch.float16):
        for (B, T, sigma) in ((2, 768, 1.0), (3, 384, 3.0), (2, 128, 1.0), (1, 64 * 5, 2.0)):
            g = torch.Generator(device="cpu").manual_seed(T)
            qkv = (sigma * torch.randn(B, T, 3 * d, generator=g)).to(dtype).to(DEV)
            q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
            ref = torch_attention(q, k, v, H, scale)
            errs = {}
            for qg in (2, 33, 32):
                _hip.set_attn_query_groups(qg)
                o = _hip.op_attention(q, k, v, H, scale).float()
                errs[qg] = (o - ref).abs().max().item()
            print(f"{dtype} B={B} T={T} sigma={sigma}: max abs err vs torch fp32: qg2 {errs[2]:.3e}  plain32 {errs[33]:.3e}  pipelined32 {errs[32]:.3e}", flush=True)
    for (B, T) in ((32, 768), (32, 384), (16, 768)):
        qkv = torch.randn(B, T, 3 * d, device=DEV).to(torch.bfloat16)
        q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
        fl = 4.0 * B * H * T * T * hd
        line = f"self-attention B={B} T={T}:"
        outs = {}
        res = {qg: [] for qg in (2, 33, 32, 35)}
        for rep in range(3):                       # interleaved repetitions; 32 / 35 = the pipelined kernel with 128- / 192-query workgroups
            for qg in res:
                _hip.set_attn_query_groups(qg)
                res[qg].append(timeit(lambda: _hip.op_attention(q, k, v, H, scale)))
                if rep == 0:
                    outs[qg] = _hip.op_attention(q, k, v, H, scale)
        for qg in res:
            us = sorted(res[qg])[1]
            line += f"  qg{qg} {us:7.1f} us {fl / us / 1e6:7.1f} TF/s |"
        print(line + f"  192- and 128-query workgroups bit-identical: {torch.equal(outs[32], outs[35])}", flush=True)
    _hip.set_attn_query_groups(0)


if __name__ == "__main__":
    main()
