"""Writes tests/golden/wide_f16_trajectory.npz: every x_t (t = 98 .. 0) of the ORACLE's fp16 99-step loop at the reference's own
classes with d_model = 512 (canvas 448, Philox seed 123) -- the loop whose end point the reference itself produced
(wide_step.npz:loop_seed123, written by make_golden.py from the reference's code).  The script asserts that the oracle still ends
on the reference's ids before it writes anything.  tests/test_gpu_bench_path.py compares every id the HIP path samples along this
trajectory (teacher-forced) without re-running the 99 CPU iterations on the GPU box (~200 s of a ~430 s suite).
~4 minutes of CPU time:  python tests/golden/make_wide_trajectory.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "tts-with-diffusion-model_amd"), ROOT]
from oracle import d3pm_oracle as O  # noqa: E402
from vall_e.vall_e import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    cfg = synth.D3PMConfig(d_model=512, n_heads=8, n_layers=6)
    sd32 = synth.make_state_dict(cfg, 0)
    texts, proms = synth.make_inputs(cfg, 1, 1)
    ref_final = np.load(os.path.join(HERE, "wide_step.npz"))["loop_seed123"].astype(np.int64)
    orc = O.Oracle({k: v.half() for k, v in sd32.items()}, O.Shape.of(cfg))
    trace = []
    with torch.no_grad():
        end = orc.generate(texts[0], proms[0], O.philox_noise(123, cfg.canvas), trace=trace)
    assert np.array_equal(end.numpy(), ref_final), "the oracle no longer reproduces the reference's d = 512 loop"
    rows = np.stack([x.numpy().astype(np.int16) for x in trace])          # ids <= 1024
    out = os.path.join(HERE, "wide_f16_trajectory.npz")
    np.savez_compressed(out, x=rows, x_0=end.numpy().astype(np.int16), seed=np.int64(123))
    print(out, rows.shape)


if __name__ == "__main__":
    main()
