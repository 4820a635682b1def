"""A/B of the paired cross-attention modes inside one process (same device, same clocks)."""
import sys, time, torch
sys.path.insert(0, "tts-with-diffusion-model_amd")
from vall_e.vall_e import _hip, synth, AR
cfg = synth.D3PMConfig.libritts()
m = AR.from_config(cfg); m.load_state_dict(synth.make_state_dict(cfg, 0)); m = m.to(torch.bfloat16).to("cuda")
texts, proms = synth.make_inputs(cfg, 32, 1)
m.generate_audio(texts, proms, steps=5, seed=1)
for rep in range(2):
    for mode in (1, 0):
        _hip.set_attn_pair_sequential(2 if mode else 0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = m.generate_audio(texts, proms, seed=3)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"sequential={mode}: {dt*1e3:7.1f} ms  {32*750/dt:9.0f} tokens/s  checksum {int(out.sum())}", flush=True)
