"""A/B (not a test): the 32-utterance reverse process launched eagerly vs replayed from a HIP graph.  python tests/ab_graph.py"""
import sys, time, torch
sys.path.insert(0, "tts-with-diffusion-model_amd")
from vall_e.vall_e import synth, AR
cfg = synth.D3PMConfig.libritts()
m = AR.from_config(cfg); m.load_state_dict(synth.make_state_dict(cfg, 0)); m = m.to(torch.bfloat16).to("cuda")
texts, proms = synth.make_inputs(cfg, 32, 1)
for g in (False, True):
    for i in range(3):
        m.generate_audio(texts, proms, seed=i, graph=g)
for rep in range(3):
    for g in (False, True):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = m.generate_audio(texts, proms, seed=3, graph=g)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"graph={g!s:5s}: {dt*1e3:7.1f} ms  {32*750/dt:9.0f} tokens/s  checksum {int(out.sum())}", flush=True)
