"""In-loop attention time per A/B arm (not a test): the sampler's own launches at the bench shape, timed by the library's
HIP-event hooks (class D3PM_K_ATTN: self-attention + the cross-attention pair of every block), interleaved repetitions.
python tests/ab_attn_loop.py [arm ...]     arms: 0 shipped, 300 software-pipelined cross-attention walk, 301 + self-attention"""
import statistics, sys, torch
sys.path[:0] = ["tts-with-diffusion-model_amd", "."]
import __graft_entry__ as g
g.build_ab()
from vall_e.vall_e import _hip, synth, AR
_hip.use_ab_library()
DEV = "cuda:0"
arms = [int(a) for a in sys.argv[1:]] or [0, 300]
cfg = synth.D3PMConfig.libritts()
m = AR.from_config(cfg)
m.load_state_dict(synth.make_state_dict(cfg, 0))
m = m.to(torch.bfloat16).to(DEV)
texts, proms = synth.make_inputs(cfg, 32, 1)
res, ids = {a: [] for a in arms}, {}
for rep in range(4):
    for arm in arms:
        _hip.set_attn_arm(arm)
        _hip.prof_enable(_hip.K_ATTN, 4096)
        out = m.generate_audio(texts, proms, steps=32, seed=5)
        n, ms, fl, by = _hip.prof_read_class(_hip.K_ATTN)
        _hip.prof_disable()
        res[arm].append(ms * 1e3 / max(n, 1))
        ids.setdefault(arm, out.clone())
_hip.set_attn_arm(0)
ref = ids[arms[0]]
for arm in arms:
    print(f"arm {arm:4d}: attention launches avg {statistics.median(res[arm]):6.1f} us (self + cross pair)/2   ids identical to arm {arms[0]}: {torch.equal(ids[arm], ref)}", flush=True)
