"""Writes tests/golden/libritts_bf16_trajectory.npz: x_t of ONE utterance at t = 75, 50, 25, 1 along the ORACLE's own bf16 99-step
trajectory at the libritts shape (d = 512, canvas 768; oracle/d3pm_oracle.py, pinned to the reference by make_golden.py), Philox
seed 123.  tests/test_gpu_measured_paths.py feeds these rows to the HIP denoiser, so that the teacher-forced check at the bench
shape does not depend on the path under test.  ~1 minute of CPU time:  python tests/golden/make_oracle_trajectory.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "tts-with-diffusion-model_amd"), ROOT]
from oracle import d3pm_oracle as O  # noqa: E402
from vall_e.vall_e import synth  # noqa: E402


def main():
    cfg = synth.D3PMConfig.libritts()
    sd32 = synth.make_state_dict(cfg, 0)
    texts, proms = synth.make_inputs(cfg, 1, 1)
    orc = O.Oracle({k: v.to(torch.bfloat16) for k, v in sd32.items()}, O.Shape.of(cfg))
    trace = []
    with torch.no_grad():
        end = orc.generate(texts[0], proms[0], O.philox_noise(123, cfg.canvas), trace=trace)
    ts = np.array([75, 50, 25, 1], np.int32)
    rows = np.stack([trace[99 - t - 1].numpy().astype(np.int16) for t in ts])          # ids <= 1024
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libritts_bf16_trajectory.npz")
    np.savez_compressed(out, t=ts, x_t=rows, x_0=end.numpy().astype(np.int16), seed=np.int64(123))
    print(out, rows.shape, "masked frames per row:", [(r[: cfg.n_frames] == cfg.mask_id).sum() for r in rows])


if __name__ == "__main__":
    main()
