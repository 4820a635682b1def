"""Glue to the reference's third-party front-ends, imported lazily (both need packages and downloaded
weights that are outside this build and absent from its image; with them installed the upstream command line works as is).

    text  -> phones     g2p_en.G2p, punctuation and blanks -> "_"     (/root/reference/vall_e/emb/g2p.py:24-28)
    wav   -> codes      EnCodec 24 kHz at 6 kbps = 8 codebooks        (/root/reference/vall_e/emb/qnt.py:19-30,57-76)
    codes -> wav        EnCodec decoder, written with soundfile       (/root/reference/vall_e/emb/qnt.py:33-49)

Nothing here is on the D3PM hot path; a missing package is an error that names it, never a silent fallback.
"""
from __future__ import annotations

import importlib
import string
from functools import lru_cache


def _need(module: str, why: str):
    try:
        return importlib.import_module(module)
    except ImportError as e:
        raise RuntimeError(f"`python -m vall_e TEXT REFERENCE OUT` needs the third-party package `{module}` ({why}); "
                           "it is not part of this build.  Use the pre-tokenised form instead "
                           "(--phn-file / --prompt-qnt, see `python -m vall_e --help`).") from e


@lru_cache(maxsize=1)
def _g2p():
    return _need("g2p_en", "grapheme-to-phoneme model of emb/g2p.py").G2p()


def g2p_encode(text: str) -> list[str]:
    ignored = {" ", *string.punctuation}
    return ["_" if p in ignored else p for p in _g2p()(text)]


@lru_cache(maxsize=2)
def _encodec(device: str):
    encodec = _need("encodec", "EnCodec codec of emb/qnt.py")
    model = encodec.EncodecModel.encodec_model_24khz()
    model.set_target_bandwidth(6.0)                       # 8 codebooks
    return model.to(device)


def encodec_encode_file(path, device: str = "cuda"):
    """wav file -> int64 [1, 8, t] (the tensor emb/qnt.py saves as `.qnt.pt`)."""
    import torch
    torchaudio = _need("torchaudio", "audio loading of emb/qnt.py")
    utils = _need("encodec.utils", "EnCodec resampling helper")
    model = _encodec(device)
    wav, sr = torchaudio.load(str(path))
    if wav.shape[0] == 2:
        wav = wav[:1]
    with torch.inference_mode():
        wav = utils.convert_audio(wav.unsqueeze(0), sr, model.sample_rate, model.channels).to(device)
        frames = model.encode(wav)
        return torch.cat([f[0] for f in frames], dim=-1)


def encodec_decode_to_file(codes_t_q, path, device: str = "cuda"):
    """(t, 8) codes -> wav file at 24 kHz."""
    import torch
    soundfile = _need("soundfile", "audio writing of emb/qnt.py")
    model = _encodec(device)
    with torch.inference_mode():
        wav = model.decode([(codes_t_q.t().unsqueeze(0).to(device), None)])
    soundfile.write(str(path), wav[0, 0].cpu().numpy(), model.sample_rate)
