"""A/B (not a test): p50 latency of one (and two) utterances under the latency-regime knobs, interleaved arms in one process.
python tests/ab_latency.py            (needs libd3pm_hip_ab.so for the LayerNorm-prologue arm: __graft_entry__.build_ab())
Arms: shipped defaults | no dual out-projection (row_panel 3) | key-split attention (attn_query_groups 4) | LayerNorm as the
prologue of the projection it feeds (A/B library knob)."""
import statistics, sys, time, torch
sys.path.insert(0, "tts-with-diffusion-model_amd")
from vall_e.vall_e import synth, AR, _hip
ab = True
try:
    _hip.use_ab_library()
except Exception as e:      # product library only: the experiment arm is skipped
    print("A/B library not loaded:", e)
    ab = False
cfg = synth.D3PMConfig.libritts()
m = AR.from_config(cfg); m.load_state_dict(synth.make_state_dict(cfg, 0)); m = m.to(torch.bfloat16).to("cuda")
arms = [("shipped", {}), ("row_panel=3 (two out-projection launches)", {"row_panel": 3}),
        ("attn_query_groups=4 (key-split attention, opt-in)", {"attn_query_groups": 4})]
if ab:
    arms.append(("LayerNorm prologue (A/B library)", {"ln_prologue": 1}))
for batch in (1, 2):
    texts, proms = synth.make_inputs(cfg, batch, 1)
    times = {name: [] for name, _ in arms}
    sums = {}
    for rep in range(9):
        for name, knobs in arms:
            _hip.reset_tuning()
            if ab: _hip.set_ln_prologue(False)
            for k, v in knobs.items():
                if k == "ln_prologue": _hip.set_ln_prologue(True)
                else: _hip.set_tuning_field(k, v)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = m.generate_audio(texts, proms, seed=3)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            if rep >= 2: times[name].append(dt * 1e3)
            sums[name] = int(out.sum())
    _hip.reset_tuning()
    if ab: _hip.set_ln_prologue(False)
    for name, _ in arms:
        t = times[name]
        print(f"batch {batch}  {name:48s} p50 {statistics.median(t):7.2f} ms  min {min(t):7.2f}  checksum {sums[name]}", flush=True)
