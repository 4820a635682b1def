"""TEST INFRASTRUCTURE (oracle side) -- numpy restatement of the sampler's counter-based noise stream.

The reference draws its per-step noise with the CPU Mersenne generator
(`torch.rand(size=x.shape+(1025,))`, /root/reference/vall_e/vall_e/ar_discrete.py:402,480), which
cannot be reproduced on a GPU.  The build replaces the *source* of the uniforms by a stateless
Philox4x32-10 stream (Salmon et al., SC'11, the published round function and constants) and
feeds the same numbers to the reference / oracle (SURVEY.md §8c "P2 shared-noise").  Everything
downstream of the uniforms (clamp, -log(-log u), argmax) stays the reference's arithmetic.

Stream definition (must match csrc/philox.h bit for bit):
    key     = (seed & 0xffffffff, seed >> 32)
    counter = (class_group = j >> 2, row = utterance_global * T + frame, t = diffusion step, stream)
    word    = j & 3 of the 4 output words
    u       = (word >> 8) * 2**-24            (fp32, in [0, 1))
stream 0 = reverse process (p_sample), stream 1 = forward noising (q_sample).
"""
from __future__ import annotations

import numpy as np

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)
_S32 = np.uint64(32)

STREAM_P_SAMPLE = 0
STREAM_Q_SAMPLE = 1


def philox4x32_10(c0, c1, c2, c3, k0: int, k1: int):
    """Vectorised Philox4x32-10. c* are broadcastable uint32 arrays; returns 4 uint32 arrays."""
    c0, c1, c2, c3 = np.broadcast_arrays(*(np.asarray(c, dtype=np.uint64) for c in (c0, c1, c2, c3)))
    k0 &= 0xFFFFFFFF
    k1 &= 0xFFFFFFFF
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> _S32, p0 & _MASK
        hi1, lo1 = p1 >> _S32, p1 & _MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0)), lo1, (hi0 ^ c3 ^ np.uint64(k1)), lo0
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def uniform_rows(seed: int, t: int, row0: int, n_rows: int, n_classes: int = 1025,
                 stream: int = STREAM_P_SAMPLE) -> np.ndarray:
    """fp32 uniforms [n_rows, n_classes] for global rows row0 .. row0+n_rows-1 at diffusion step t."""
    groups = (n_classes + 3) // 4
    g = np.arange(groups, dtype=np.uint64)[None, :]
    r = (np.arange(n_rows, dtype=np.uint64) + np.uint64(row0))[:, None]
    words = philox4x32_10(g, r, np.uint64(t), np.uint64(stream), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    w = np.stack(words, axis=-1).reshape(n_rows, groups * 4)[:, :n_classes]
    return ((w >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24))


def uniform_batch(seed: int, t: int, utt0: int, batch: int, frames: int, n_classes: int = 1025,
                  stream: int = STREAM_P_SAMPLE) -> np.ndarray:
    """fp32 uniforms [batch, frames, n_classes]; utterance b uses rows (utt0+b)*frames .. +frames-1."""
    return uniform_rows(seed, t, utt0 * frames, batch * frames, n_classes, stream).reshape(batch, frames, n_classes)
