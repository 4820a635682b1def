"""A/B (not a test): the cross-attention pair of a block, resident on the 16 x 16 x 32 instruction (attn_cross_resident 4) against
resident on the 32 x 32 x 16 instruction with the software-pipelined block (5), and tile by tile (0): microbenchmark at the
bench shape and the attention class inside the sampler's loop (HIP-event hooks).   python tests/ab_cross32.py"""
import math, statistics, sys, torch
sys.path[:0] = ["tts-with-diffusion-model_amd", "."]
from vall_e.vall_e import _hip, synth, AR
DEV = "cuda:0"


def timeit(f, n=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for dtype in (torch.bfloat16, torch.float16):
    for B, Tq, S1, S2 in ((32, 768, 50, 225), (32, 384, 50, 256), (16, 768, 50, 225)):
        H, hd = 8, 64
        d = H * hd
        g = torch.Generator(device="cpu").manual_seed(1)
        q = torch.randn(2, B, Tq, d, generator=g).to(dtype).to(DEV)
        kv1 = torch.randn(B, S1, 2 * d, generator=g).to(dtype).to(DEV)
        kv2 = torch.randn(B, S2, 2 * d, generator=g).to(dtype).to(DEV)
        scale = math.sqrt(1.0 / hd)
        f = lambda: _hip.op_attention_pair(q[0], kv1[..., :d], kv1[..., d:], q[1], kv2[..., :d], kv2[..., d:], H, scale)
        res, outs = {a: [] for a in (0, 4, 5)}, {}
        for rep in range(5):
            for arm in (0, 4, 5):
                _hip.set_attn_cross_resident(arm)
                res[arm].append(timeit(f))
                if rep == 0: outs[arm] = [o.float() for o in f()]
        _hip.set_attn_cross_resident(1)
        fl = 4.0 * B * H * Tq * (S1 + S2) * hd
        line = f"{str(dtype)[6:]:9s} B={B} Tq={Tq} S={S1}+{S2}:"
        for arm in (0, 4, 5):
            t = statistics.median(res[arm])
            line += f"  arm {arm}: {t:6.1f} us {fl / t / 1e6:5.0f} TF/s"
        diff = max((a - b).abs().max().item() for a, b in zip(outs[4], outs[5]))
        print(line + f"  | max |resident32 - resident16| = {diff:.2e}", flush=True)

cfg = synth.D3PMConfig.libritts()
m = AR.from_config(cfg)
m.load_state_dict(synth.make_state_dict(cfg, 0))
m = m.to(torch.bfloat16).to(DEV)
texts, proms = synth.make_inputs(cfg, 32, 1)
res, ids = {a: [] for a in (4, 5)}, {}
for rep in range(4):
    for arm in (4, 5):
        _hip.set_attn_cross_resident(arm)
        _hip.prof_enable(_hip.K_ATTN, 4096)
        out = m.generate_audio(texts, proms, steps=32, seed=5)
        n, ms, fl, by = _hip.prof_read_class(_hip.K_ATTN)
        _hip.prof_disable()
        res[arm].append(ms * 1e3 / max(n, 1))
        ids.setdefault(arm, out.clone())
_hip.set_attn_cross_resident(1)
for arm in (4, 5):
    print(f"in the loop, arm {arm}: attention launches avg {statistics.median(res[arm]):6.1f} us (self + cross pair)/2; "
          f"ids equal to arm 4 on {(ids[arm] == ids[4]).float().mean().item():.4f} of the frames", flush=True)
